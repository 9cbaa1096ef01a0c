#!/bin/bash
# Collect PMC counters for the bench in separate passes (gpurun forbids --pmc with trace domains other than kernel-trace).
# usage: tools/pmc_run.sh <outdir> <families> 
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; FAM=${2:-1000000}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc$i -- python $GRAFT_REPO_ROOT/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
ls $OUT
