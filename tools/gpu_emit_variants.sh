#!/bin/bash
# duplex / CODEC record writers: their GPU test files on the product library, then the duplex and CODEC bench lines for several library builds.
# usage: bash tools/gpu_emit_variants.sh <tag> <rounds> v1 v2 ...
TAG=$1; ROUNDS=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_duplex.py tests/test_gpu_codec.py tests/test_gpu_duplex_canon.py tests/test_gpu_zz_canon_device.py tests/test_gpu_zz_codec_canon.py tests/test_gpu_methylation.py -m gpu -q -p no:cacheprovider -rfE --timeout 500 > $OUT/pytest_emit.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_emit.log | head -20
for c in duplex codec; do
  for r in $(seq 1 $ROUNDS); do
    for v in "$@"; do
      if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
      timeout 300 python bench.py --caller $c --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${c}_${v}_$r.log 2>&1
      grep '^{' $OUT/${c}_${v}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $v r$r', 'value=%.4g ms_step=%.2f'%(d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in d['roofline'].items() if k.endswith('_ms')})" || tail -5 $OUT/${c}_${v}_$r.log
    done
  done
done
