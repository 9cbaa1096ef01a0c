#!/bin/bash
# usage: tools/gpu_ab.sh <tag> <test-k-expr> v1 v2 ...   — quick GPU parity subset + d8 / d3 / lt bench lines for the product library ("main")
# and for fgumi_amd/variant_<v>.so builds, in one gpurun call
R=$GRAFT_REPO_ROOT; TAG=$1; KEXPR=$2; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for v in main "$@"; do
  if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
  timeout 600 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/pytest_$v.log 2>&1; echo "$v pytest rc=$? $(tail -1 $OUT/pytest_$v.log)"
  b() { local name=$1; shift
    timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${name}_$v.log 2>&1
    grep '^{' $OUT/${name}_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.3g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))" || tail -5 $OUT/${name}_$v.log
  }
  b d8 --families 1000000
  b d3 --families 2000000 --depth 3
  b lt --families 1000000 --depth 2 --depth-max 50
done
