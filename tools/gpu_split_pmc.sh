#!/bin/bash
# PMC passes + phase shares of the split simplex pipeline. usage: bash tools/gpu_split_pmc.sh <tag> [families]
R=$GRAFT_REPO_ROOT; TAG=$1; FAM=${2:-1000000}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc -o pmc$i -- python $R/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
done
python $R/tools/pmc_parse.py $OUT/pmc 3 > $OUT/pmc.json
rm -rf $OUT/pmc
python - $OUT/pmc.json $FAM <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); fam=float(sys.argv[2])
for k in ('k_split_parse','k_split_cols','k_emit','k_call_full','k_simplex_wave2'):
    if k in d: print(k, {c.replace('SQ_','').replace('SQC_',''): round(v/fam,1) for c,v in sorted(d[k].items())})
PY
cd $R
FGX_LIB=$R/fgumi_amd/variant_phase.so timeout 300 python bench.py --families $FAM --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "phase share" | head -2
