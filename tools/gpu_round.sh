#!/bin/bash
# One gpurun call of a development round: the whole GPU suite, the default bench line, optionally the kernel trace.
# usage (on the GPU box, via gpurun):  bash tools/gpu_round.sh <tag> [trace]      results land in gpurun_out/<tag>/
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -25
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-700 $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("value %.4g  ms/step %.2f  k_family %.2f  k_emit %.2f  frac %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["k_emit_ms"], r["frac"]))
    print("strong:", json.dumps(d.get("strong_scaling"))[:600])
    print("cpu:", json.dumps(d.get("cpu_baseline"))[:300])
except Exception as e:
    print("no bench line:", e)
PY
if [ "$2" = "trace" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/stats.log 2>&1
  rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
  python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
fi
