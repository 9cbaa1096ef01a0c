#!/bin/bash
# Direct-records check round (on the GPU box, via gpurun): tools/direct_check.py (direct vs scratch vs oracle, field-level diffs), then the
# GPU suite and the bench line + kernel trace.   usage: bash tools/gpu_direct.sh <tag> [quick]
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
FGX_S2_VERBOSE=1 timeout 600 python tools/direct_check.py > $OUT/direct_check.txt 2>&1; echo "direct_check rc=$?"
grep -v "^\[fgx\] split pipeline" $OUT/direct_check.txt | head -120
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -25
fi
timeout 400 python bench.py --no-strong-block --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - $OUT/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("value %.4g  ms/step %.2f  k_family %.2f  k_emit %.2f  frac %.4f  deferred %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["k_emit_ms"], r["frac"], d["config"]["deferred_families"]))
except Exception as e:
    print("no bench line:", e)
PY
FGX_S2_VERBOSE=1 FGX_DIRECT=1 timeout 400 python bench.py --no-strong-block --no-cpu-baseline --steps 5 2> $OUT/bench_direct.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('FGX_DIRECT=1: value %.4g ms/step %.2f k_family %.2f k_emit %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['k_emit_ms']))"
grep "direct records" $OUT/bench_direct.err | tail -2
cd /tmp; export TMPDIR=/tmp
FGX_DIRECT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex_direct -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
