#!/bin/bash
# usage: tools/gpu_variants.sh <tag> <rounds> <bench args ...> -- v1 v2 ...   — the same bench line for several library builds
# (fgumi_amd/variant_<v>.so; "main" = the product library), interleaved <rounds> times so that box drift shows as noise
R=$GRAFT_REPO_ROOT; TAG=$1; ROUNDS=$2; shift; shift
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
    timeout 300 python bench.py "${ARGS[@]}" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${v}_$r.log 2>&1
    grep '^{' $OUT/${v}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v r$r', 'k_family_ms=%.3f k_emit_ms=%.2f ms_step=%.2f'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step']))" || tail -5 $OUT/${v}_$r.log
  done
done
