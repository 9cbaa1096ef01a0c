#!/bin/bash
# Round 6 development call: a set of GPU test files (or the whole suite), then the headline line for several library builds interleaved
# (fgumi_amd/variant_<v>.so; "main" = the product library), then optionally the kernel stats of the product.
# usage (via gpurun): bash tools/gpu_r6.sh <tag> <tests: "all" | "none" | "file1 file2 ..."> <rounds> <stats: 0|1> v1 v2 ...
TAG=$1; TESTS=$2; ROUNDS=$3; STATS=$4; shift; shift; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
if [ "$TESTS" = "all" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
elif [ "$TESTS" != "none" ]; then
  timeout 1200 python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider -rfE --timeout 600 --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
fi
[ "$TESTS" != "none" ] && grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -15
[ "$ROUNDS" != "0" ] && bash tools/gpu_variants.sh $TAG $ROUNDS --no-strong-block --end-to-end-families 0 -- "$@"
if [ "$STATS" = "1" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 > $OUT/stats.log 2>&1
  rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*/*_agent_info.csv $OUT/*/*kernel_trace.csv
  python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
fi
