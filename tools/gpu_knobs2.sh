#!/bin/bash
# bench lines under environment knobs, interleaved twice. usage: bash tools/gpu_knobs2.sh <tag> <families> "ENV1=.." "ENV2=.." ...   ("-" = no knob)
R=$GRAFT_REPO_ROOT; TAG=$1; FAM=$2; shift; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for round in 1 2; do i=0; for kv in "$@"; do i=$((i+1))
  if [ "$kv" = "-" ]; then E=""; else E="$kv"; fi
  env $E timeout 300 python bench.py --families $FAM --steps 4 --warmup 1 --no-cpu-baseline > $OUT/k$i.$round.log 2>&1
  grep '^{' $OUT/k$i.$round.log | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('%-28s'%'$kv', 'k_family_ms=%.2f k_emit_ms=%.2f device_ms=%.2f ms_step=%.2f'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['roofline']['device_ms_per_step'], d['ms_per_step']))"
done; done
