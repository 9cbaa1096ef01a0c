#!/bin/bash
# Variants of the streaming column kernel on the long-tail shape: the deep tests with the product library, then rocprofv3 kernel stats per variant.
# usage: bash tools/gpu_deep_variants.sh <tag> v1 v2 ...   ("main" = the product library)
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_deep_families.py -m gpu -q -p no:cacheprovider -rfEs --timeout 600 > $OUT/pytest_deep.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_deep.log | head -20
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lt_$v -- python $R/bench.py --depth 2 --depth-max 50 --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/lt_$v.log 2>&1
  python - $OUT lt_$v $v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/' + sys.argv[2] + '_kernel_stats.csv', recursive=True):
    rows = {r['Name']: r for r in csv.DictReader(open(f))}
    pick = lambda s: next((float(r['AverageNs']) / 1e6 for n, r in rows.items() if s in n), 0.0)
    print('%-10s k_deep_cols %.3f ms  k_deep_parse %.3f ms' % (sys.argv[3], pick('k_deep_cols'), pick('k_deep_parse')), end='  ')
PY
  grep '^{' $OUT/lt_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3f G  ms/step %.2f stage %.2f'%(d['value']/1e9, d['ms_per_step'], r['kernel_ms']))" || tail -3 $OUT/lt_$v.log
done
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*domain_stats.csv
