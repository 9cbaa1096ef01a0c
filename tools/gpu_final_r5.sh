#!/bin/bash
# Round 5, last GPU call (the budget had 6 minutes left): the test files no call had run since the last full suite, then — only if they pass — the
# four PMC passes of the bench workload, the default bench line, and the kernel stats.  usage (via gpurun): bash tools/gpu_final_r5.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 200 python -m pytest tests/test_golden.py tests/test_gpu_indels.py tests/test_gpu_direct_records.py -m gpu -q -x -p no:cacheprovider --timeout 120 > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log | cut -c1-300
[ $rc -ne 0 ] && { echo "tests failed (rc=$rc): nothing else is run"; exit 1; }
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc -o pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_parse.py $OUT/pmc 3 > $OUT/pmc_5M_families.json
cp $OUT/pmc_5M_families.json $R/profiles/${TAG}_pmc_5M_families.json
rm -rf $OUT/pmc
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*/*_agent_info.csv $OUT/*/*kernel_trace.csv
cd $R
timeout 200 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
