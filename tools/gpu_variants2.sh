#!/bin/bash
# A/B of library builds on the default bench line, interleaved: usage bash tools/gpu_variants2.sh <tag> <families> v1 v2 ... ("main" = product)
R=$GRAFT_REPO_ROOT; TAG=$1; FAM=$2; shift; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for round in 1 2; do for v in "$@"; do
  if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
  timeout 300 python bench.py --families $FAM --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$v.$round.log 2>&1
  grep '^{' $OUT/$v.$round.log | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('$v', 'k_family_ms=%.2f k_emit_ms=%.2f device_ms=%.2f ms_step=%.2f'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['roofline']['device_ms_per_step'], d['ms_per_step']))"
done; done
