#!/bin/bash
# Full-size measurement of the bench workload: bench line (+cpu baseline), rocprofv3 kernel stats, PMC passes.
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag> [families]
TAG=$1; FAM=${2:-5000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --families $FAM > $OUT/bench_line.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/bench.py --families $FAM --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc$i -- python $GRAFT_REPO_ROOT/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
ls -la $OUT
