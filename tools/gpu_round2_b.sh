#!/bin/bash
# phase shares of the new kernel, tuning variants, PMC passes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02b; mkdir -p $OUT
cd $R
FGX_LIB=$R/fgumi_amd/variant_phase.so timeout 300 python bench.py --families 1000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/phase_d8.log 2>&1
grep "phase share" $OUT/phase_d8.log
FGX_LIB=$R/fgumi_amd/variant_phase.so timeout 300 python bench.py --families 2000000 --depth 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/phase_d3.log 2>&1
grep "phase share\|^{" $OUT/phase_d3.log | cut -c1-400
bash tools/variants.sh 1000000 b1 b4 occ4
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc$i -- python $R/bench.py --families 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_parse.py $OUT > $OUT/pmc_1M.json
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02b/pmc_1M.json'))
for k in ('k_simplex_wave2','k_emit'):
    if k in d:
        print(k, {c: round(v/1e6,1) for c,v in d[k].items()})
PY
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*counter_collection.csv
