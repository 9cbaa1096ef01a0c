#!/bin/bash
# The headline line for ONE library under several environment settings, interleaved (each "NAME=VALUE[,NAME=VALUE]" argument is one setting; "-" = none).
# Measurement knobs exist only in the profiling build: LIB=fgumi_amd/variant_knobs.so (python -m fgumi_amd.build --variant knobs -DFGX_KNOBS=1).
# usage (via gpurun): LIB=... bash tools/gpu_env_ab.sh <tag> <rounds> <setting> ...
TAG=$1; ROUNDS=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
[ -n "$LIB" ] && export FGX_LIB=$R/$LIB
for r in $(seq 1 $ROUNDS); do
  for S in "$@"; do
    name=$(echo "$S" | tr ',=' '__'); [ "$S" = "-" ] && name=default
    envs=""; [ "$S" != "-" ] && envs=$(echo "$S" | tr ',' ' ')
    env $envs timeout 300 python bench.py --no-cpu-baseline --no-strong-block --end-to-end-families 0 --steps 6 --warmup 2 $BENCH_ARGS > $OUT/line_${name}_$r.json 2> $OUT/err_${name}_$r.txt
    python - $OUT/line_${name}_$r.json "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-40s value %.4g  ms/step %.2f  k_family %.2f  k_emit %.2f  frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["k_emit_ms"], r["frac"]))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
  done
done
