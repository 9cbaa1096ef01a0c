#!/bin/bash
# usage: bash tools/gpu_duplex_variants.sh <tag> <caller> v1 v2 ...   — rocprofv3 kernel stats of the duplex / CODEC shape per library variant ("main" = product)
TAG=$1; CALLER=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${CALLER}_$v -- python $R/bench.py --caller $CALLER --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/${CALLER}_$v.log 2>&1
  python - $OUT ${CALLER}_$v $v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/' + sys.argv[2] + '_kernel_stats.csv', recursive=True):
    rows = {r['Name']: r for r in csv.DictReader(open(f))}
    pick = lambda s: next((float(r['AverageNs']) / 1e6 for n, r in rows.items() if s in n), 0.0)
    print('%-10s k_family_wave %.3f ms  emit_fast %.3f ms' % (sys.argv[3], pick('k_family_wave'), pick('_fast')), end='  ')
PY
  grep '^{' $OUT/${CALLER}_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3f G  ms/step %.2f stage %.2f deferred %s'%(d['value']/1e9, d['ms_per_step'], r['kernel_ms'], d['config'].get('deferred_families')))" || tail -3 $OUT/${CALLER}_$v.log
done
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*domain_stats.csv
