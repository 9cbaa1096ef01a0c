#!/bin/bash
# Quick check of a kernel change on the GPU box: field-level diff against the oracle, (optionally) the GPU parity suite, 1M / 5M bench
# lines and the kernel averages.  usage: bash tools/gpu_quick.sh <tag> [pytest]
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
FGX_S2_VERBOSE=1 timeout 600 python tools/split_debug.py quick > $OUT/split_debug.log 2>&1; echo "split_debug rc=$?" >> $OUT/split_debug.log
grep -v "^\[fgx\]\|amdgpu.ids" $OUT/split_debug.log | tail -30
if [ "$2" = pytest ]; then timeout 900 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log; fi
line() { local name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $OUT/$name.log 2>&1; grep '^{' $OUT/$name.log | tail -1 > $OUT/${name}_bench_line.json
  python -c "import sys,json; d=json.load(open('$OUT/${name}_bench_line.json')); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f device_ms=%.2f ms_step=%.2f reads/s=%.4g deferred=%s full=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['roofline']['device_ms_per_step'], d['ms_per_step'], d['value'], d['config']['deferred_families'], d['config']['columns_needing_call_full_per_step']))" || tail -5 $OUT/$name.log; }
FGX_S2_VERBOSE=1 line split_1M --families 1000000 --steps 4 --warmup 1
line split_5M --steps 4 --warmup 1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o split -- python $R/bench.py --families 1000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
python - $OUT <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:7]: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
