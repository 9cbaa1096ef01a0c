#!/bin/bash
# Round 5: where k_split_cols's TIME goes.  The phase-ablation builds (fgumi_amd/variant_s2abl<k>.so: the kernel returns after phase k) and the
# product library (sum-free step on / off) each run 1 M depth-8 families as ONE chunk (the kernel alone on the chip) under a kernel trace with
# one PMC pass; per variant: duration and counters of the LARGEST k_split_cols launch.  Consecutive differences are the phases — in time,
# not only in instructions (profiles/r03b used the counters alone).
# usage (via gpurun): [PMC="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ..."] bash tools/gpu_ablate_time.sh <tag> [families]
TAG=$1; FAM=${2:-1000000}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name lib env...
  local name=$1 lib=$2; shift; shift
  env FGX_LIB=$lib FGX_SPLIT_CHUNKS=1 "$@" timeout 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY} --output-format csv -d $OUT/$name -o p -- python $R/bench.py --families $FAM --steps 2 --warmup 1 --no-cpu-baseline --no-strong-block --end-to-end-families 0 > $OUT/$name.log 2>&1
  python - $OUT/$name $name $FAM <<'PY'
import csv, glob, sys, collections
d, name, fam = sys.argv[1], sys.argv[2], float(sys.argv[3])
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_split_cols" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_split_cols" in r["Kernel_Name"]:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in dur and "Start_Timestamp" in r:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
if not dur:
    print(name, "no k_split_cols launch found"); sys.exit(0)
big = max(dur, key=lambda k: dur[k][1])
same = [k for k in dur if dur[k][1] == dur[big][1]]
ns = sum(dur[k][0] for k in same) / len(same)
c = collections.defaultdict(float)
for k in same:
    for a, b in cnt[k].items(): c[a] += b / len(same)
waves = c.get("SQ_WAVES", 0) or 1
print("%-14s %8.3f ms per %d families | per family: VALU %7.1f SALU %7.1f LDS %6.1f | wave cycles %8.0f busy %.3g active_valu %.3g wait_inst %.3g" % (
    name, ns / 1e6, int(fam), c["SQ_INSTS_VALU"] / fam, c["SQ_INSTS_SALU"] / fam, c["SQ_INSTS_LDS"] / fam, c["SQ_WAVE_CYCLES"] / waves, c["SQ_BUSY_CYCLES"], c["SQ_ACTIVE_INST_VALU"], c["SQ_WAIT_INST_ANY"]))
extra = {k: v for k, v in c.items() if k not in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY")}
if extra: print("%-14s   other counters per family: %s" % (name, {k: round(v / fam, 1) for k, v in extra.items()}))
PY
  rm -rf $OUT/$name
}
for k in 1 3 4 5 6; do [ -f $R/fgumi_amd/variant_s2abl$k.so ] && run abl$k $R/fgumi_amd/variant_s2abl$k.so; done
run product $R/fgumi_amd/libfgumi_amd.so
run product_unpacked $R/fgumi_amd/libfgumi_amd.so FGX_S2_PACKED=0
