#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02abl2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for k in 5 62 61 6 0; do
  LIBF=$R/fgumi_amd/variant_abl$k.so; [ $k = 0 ] && LIBF=$R/fgumi_amd/libfgumi_amd.so
  FGX_LIB=$LIBF timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$k -o p -- python $R/bench.py --families 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a$k.log 2>&1
  python $R/tools/pmc_parse.py $OUT/a$k > $OUT/abl$k.json; rm -rf $OUT/a$k
  python -c "
import json; d=json.load(open('$OUT/abl$k.json'))['k_simplex_wave2']; print('abl$k', {c.replace('SQ_INSTS_',''): round(d[c]/1e6,1) for c in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM')})"
done
cd $R; timeout 300 python bench.py --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('d8 k_family_ms=%.2f ms_step=%.2f'%(d['roofline']['kernel_ms'], d['ms_per_step']))"
