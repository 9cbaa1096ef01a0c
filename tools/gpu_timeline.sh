#!/bin/bash
# Round 5: the launch timeline of ONE bench step (every kernel with its start / end relative to the step's first kernel, stream by queue id).
# usage (via gpurun): bash tools/gpu_timeline.sh <tag> [bench args / env via ENVS="A=1 B=2"]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
env $ENVS timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 "$@" > $OUT/tl.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: from the last k_col_bound (first kernel of a step) on
idx = [i for i, r in enumerate(rows) if 'k_col_bound' in r['Kernel_Name']]
start = idx[-1] if idx else 0
t0 = int(rows[start]['Start_Timestamp'])
def short(n):
    n = n.replace('fgx::(anonymous namespace)::', '').replace('void ', '')
    return n[:44]
with open(sys.argv[1] + '/timeline.txt', 'w') as o:
    for r in rows[start:]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        line = "%9.3f %9.3f %8.3f q%-3s %s grid %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get('Queue_Id', '?'), short(r['Kernel_Name']), r.get('Grid_Size_X', r.get('Grid_Size', '?')))
        o.write(line + "\n")
print(open(sys.argv[1] + '/timeline.txt').read()[:6000])
PY
rm -rf $OUT/*/*_agent_info.csv $OUT/*/*kernel_trace.csv $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
