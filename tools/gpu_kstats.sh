#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the bench workload under knobs. usage: gpu_kstats.sh <tag> <families> "ENV.." ...  ("-" = none)
R=$GRAFT_REPO_ROOT; TAG=$1; FAM=$2; shift; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for kv in "$@"; do i=$((i+1))
  if [ "$kv" = "-" ]; then E=""; else E="$kv"; fi
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$i -o s -- python $R/bench.py --families $FAM --steps 3 --warmup 1 --no-cpu-baseline > $OUT/s$i.log 2>&1
  echo "== $kv"
  python - $OUT/s$i <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f))):
        n=r['Name']
        if any(k in n for k in ('k_split','k_emit','k_call_full','k_simplex','k_col_bound')): print('   %-60s calls %5s avg %10.1f us total %10.1f us'%(n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
  grep '^{' $OUT/s$i.log | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('   k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step']))"
  rm -rf $OUT/s$i
done
