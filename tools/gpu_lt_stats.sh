#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02lt; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lt -- python $R/bench.py --families 1000000 --depth 2 --depth-max 50 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/lt.log 2>&1
python - $OUT <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:75], r['Calls'], r['AverageNs'], r['Percentage'])
PY
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
