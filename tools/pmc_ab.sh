#!/bin/bash
# PMC comparison of library builds on the bench workload: usage tools/pmc_ab.sh <tag> <families> v1 v2 ...  ("main" = product library)
R=$GRAFT_REPO_ROOT; TAG=$1; FAM=$2; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = main ]; then unset FGX_LIB; else export FGX_LIB=$R/fgumi_amd/variant_$v.so; fi
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_ICACHE_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$v -o pmc$i -- python $R/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${v}_pmc$i.log 2>&1 || tail -3 $OUT/${v}_pmc$i.log
  done
  python $R/tools/pmc_parse.py $OUT/$v > $OUT/pmc_$v.json
  rm -rf $OUT/$v
  python - $OUT/pmc_$v.json $FAM $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])).get('k_simplex_wave2',{}); fam=float(sys.argv[2])
print(sys.argv[3], {k.replace('SQ_','').replace('SQC_',''): round(v/fam,1) for k,v in sorted(d.items())})
PY
done
