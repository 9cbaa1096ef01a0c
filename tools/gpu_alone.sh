#!/bin/bash
# Kernels of the split pipeline ALONE on the chip (FGX_SPLIT_CHUNKS=1 serialises the record kernel, the column kernel and the finish kernel),
# with and without direct records: rocprofv3 kernel trace of a 1 M-family bench step.   usage: bash tools/gpu_alone.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for D in 0 1; do
  FGX_DIRECT=$D FGX_SPLIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o alone_d$D -- python $R/bench.py --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/alone_d$D.log 2>&1
  echo "== FGX_DIRECT=$D, chunks=1, 1 M families"
  python - $OUT alone_d$D <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/' + sys.argv[2] + '_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        if 'sim_generate' in r['Name']: continue
        print('  ', r['Name'][:60], r['Calls'], '%.3f ms' % (float(r['AverageNs']) / 1e6))
PY
  grep '^{' $OUT/alone_d$D.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ms/step %.2f k_family %.2f k_emit %.2f'%(d['ms_per_step'], r['kernel_ms'], r['k_emit_ms']))"
done
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
