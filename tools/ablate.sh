#!/bin/bash
# Per-phase instruction counts of k_simplex_wave2: builds that return after phase k (fgumi_amd/variant_abl<k>.so, built with
# `python -m fgumi_amd.build --variant abl<k> -DFGX_ABLATE=<k>`) run the bench workload under two PMC passes each; consecutive
# differences are the phases.  usage (GPU box): bash tools/ablate.sh <outdir> [families]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; FAM=${2:-1000000}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for k in ${ABL_SET:-1 2 3 4 5 6 0}; do
  LIBF=$R/fgumi_amd/variant_abl$k.so; [ $k = 0 ] && LIBF=$R/fgumi_amd/libfgumi_amd.so
  FGX_LIB=$LIBF timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$k -o p -- python $R/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a$k.log 2>&1
  python $R/tools/pmc_parse.py $OUT/a$k > $OUT/abl$k.json
  rm -rf $OUT/a$k
done
python - $OUT $FAM <<'PY'
import json,sys
out,fam=sys.argv[1],float(sys.argv[2])
names={1:"stage",2:"parse",3:"pairing+overlap",4:"geometry",5:"gates+umi gating",6:"columns+call",0:"umi+descriptors+stats"}
prev={"SQ_INSTS_VALU":0,"SQ_INSTS_SALU":0,"SQ_INSTS_LDS":0,"SQ_INSTS_VMEM":0}
res={}
for k in (1,2,3,4,5,6,0):
    d=json.load(open(f"{out}/abl{k}.json")).get("k_simplex_wave2",{})
    cur={c:d.get(c,0)/fam for c in prev}
    res[names[k]]={c.replace("SQ_INSTS_","").lower():round(cur[c]-prev[c],1) for c in prev}
    prev=cur
res["total"]={c.replace("SQ_INSTS_","").lower():round(prev[c],1) for c in prev}
json.dump(res,open(f"{out}/phase_instructions.json","w"),indent=1)
for k,v in res.items(): print("%-24s"%k,v)
PY
