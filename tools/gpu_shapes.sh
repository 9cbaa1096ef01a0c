#!/bin/bash
# Kernel traces of the secondary shapes: long tail (configs[3]), duplex (configs[2]), CODEC (configs[4]).  usage: bash tools/gpu_shapes.sh <tag> [shape ...]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SHAPES=${@:-longtail duplex codec}
for S in $SHAPES; do
  case $S in
    longtail) ARGS="--depth 2 --depth-max 50 --families 1000000";;
    duplex) ARGS="--caller duplex";;
    codec) ARGS="--caller codec";;
    depth3) ARGS="--depth 3";;
    *) ARGS="";;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $S -- python $R/bench.py $ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/$S.log 2>&1
  echo "== $S ($ARGS)"
  python - $OUT $S <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/' + sys.argv[2] + '_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        if 'sim_generate' in r['Name']: continue
        print('  ', r['Name'][:70], r['Calls'], 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'total %.2f ms' % (float(r['TotalDurationNs']) / 1e6))
PY
  grep '^{' $OUT/$S.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   value %.3f G  ms/step %.2f k_family %.2f k_emit %.2f frac %.4f deferred %s'%(d['value']/1e9, d['ms_per_step'], r['kernel_ms'], r['k_emit_ms'], r['frac'], d['config'].get('deferred_families')))" || tail -5 $OUT/$S.log
done
for S in $SHAPES; do python $R/tools/launch_times.py $OUT $S k_split_cols 12; done
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
