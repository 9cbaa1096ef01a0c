#!/bin/bash
# usage: tools/gpu_check_all.sh <tag>   — gpu_check.sh plus the duplex / CODEC bench shapes
R=$GRAFT_REPO_ROOT; bash $R/tools/gpu_check.sh $1
cd $R
for c in duplex codec; do
  timeout 300 python bench.py --caller $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', 'kernel_ms=%.2f ms_step=%.2f reads/s=%.4g def=%s'%(d['roofline']['kernel_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))"
done
