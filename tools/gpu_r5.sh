#!/bin/bash
# Round 5 development call: the parity files that cover the split pipeline, then the headline line with environment A/B settings interleaved
# (each "NAME=VALUE[,NAME=VALUE]" argument is one setting; "-" = the defaults), then the kernel trace of the defaults.
# usage (via gpurun): bash tools/gpu_r5.sh <tag> <tests: cols|all|none> <rounds> [setting ...]
TAG=$1; TESTS=$2; ROUNDS=$3; shift; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
if [ "$TESTS" = "cols" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_full_size.py tests/test_gpu_deep_families.py tests/test_gpu_indels.py tests/test_gpu_direct_records.py tests/test_gpu_full_size.py -m gpu -q -x -p no:cacheprovider -rfE --timeout 600 --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
elif [ "$TESTS" = "all" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
fi
[ "$TESTS" != "none" ] && grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -15
for r in $(seq 1 $ROUNDS); do
  for S in "$@"; do
    name=$(echo "$S" | tr ',=' '__'); [ "$S" = "-" ] && name=default
    envs=""; [ "$S" != "-" ] && envs=$(echo "$S" | tr ',' ' ')
    env $envs timeout 300 python bench.py --no-cpu-baseline --no-strong-block --end-to-end-families 0 --steps 10 --warmup 2 > $OUT/line_${name}_$r.json 2> $OUT/err_${name}_$r.txt
    python - $OUT/line_${name}_$r.json "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-40s value %.4g  ms/step %.2f  k_family %.2f  k_emit %.2f  frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["k_emit_ms"], r["frac"]))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
  done
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*/*_agent_info.csv $OUT/*/*kernel_trace.csv
python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
