#!/bin/bash
# Round 5: secondary shapes with environment A/B settings ("-" = defaults).  usage (via gpurun): bash tools/gpu_r5_shapes.sh <tag> "<bench args>" [setting ...]
TAG=$1; ARGS=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for S in "$@"; do
  name=$(echo "$S" | tr ',=' '__'); [ "$S" = "-" ] && name=default
  envs=""; [ "$S" != "-" ] && envs=$(echo "$S" | tr ',' ' ')
  env $envs timeout 300 python bench.py $ARGS --no-cpu-baseline --no-strong-block --end-to-end-families 0 --steps 5 --warmup 1 > $OUT/line_${name}.json 2> $OUT/err_${name}.txt
  python - $OUT/line_${name}.json "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-40s value %.4g  ms/step %.2f  k_family %.2f  k_emit %.2f  frac %.4f deferred %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["k_emit_ms"], r["frac"], d["config"].get("deferred_families")))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
done
