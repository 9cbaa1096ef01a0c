#!/bin/bash
# Per-phase instruction counts of k_split_cols: builds that return after phase k (fgumi_amd/variant_s2abl<k>.so, built with
# `python -m fgumi_amd.build --variant s2abl<k> -DFGX_S2_ABLATE=<k>`) run the bench workload under one PMC pass each; consecutive
# differences are the phases.  usage (GPU box): bash tools/ablate_split.sh <outdir> [families]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; FAM=${2:-1000000}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for k in 1 3 4 5 6 0; do
  LIBF=$R/fgumi_amd/variant_s2abl$k.so; [ $k = 0 ] && LIBF=$R/fgumi_amd/libfgumi_amd.so
  FGX_LIB=$LIBF timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$k -o p -- python $R/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a$k.log 2>&1
  python $R/tools/pmc_parse.py $OUT/a$k > $OUT/abl$k.json
  rm -rf $OUT/a$k
done
python - $OUT $FAM <<'PY'
import json,sys
out,fam=sys.argv[1],float(sys.argv[2])
names={1:"prologue+stage",3:"overlap",4:"clip/fill+final_len",5:"gates",6:"columns+call",0:"descriptors+items+stats"}
prev={"SQ_INSTS_VALU":0,"SQ_INSTS_SALU":0,"SQ_INSTS_LDS":0,"SQ_INSTS_VMEM":0,"SQ_WAVE_CYCLES":0}
res={}
for k in (1,3,4,5,6,0):
    d=json.load(open(f"{out}/abl{k}.json")).get("k_split_cols",{})
    cur={c:d.get(c,0)/fam for c in prev}
    res[names[k]]={c.replace("SQ_INSTS_","").lower():round(cur[c]-prev[c],1) for c in prev}
    prev=cur
res["total"]={c.replace("SQ_INSTS_","").lower():round(prev[c],1) for c in prev}
json.dump(res,open(f"{out}/phase_instructions.json","w"),indent=1)
for k,v in res.items(): print("%-26s"%k,v)
PY
