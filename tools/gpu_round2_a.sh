#!/bin/bash
# round-2 first GPU pass: parity suite with the new simplex kernel, on/off bench, kernel stats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02a; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for v in 1 0; do
  FGX_V2=$v timeout 300 python bench.py --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_1m_v2_$v.log 2>&1
  grep '^{' $OUT/bench_1m_v2_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v2=$v', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.3g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))"
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $R/bench.py --families 5000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_5m_line.json
python - <<'PY'
import csv,glob,os
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02a/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
ls $OUT
