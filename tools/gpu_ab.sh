#!/bin/bash
# GPU parity suite of the product build, then an interleaved A/B of library builds / knobs on the bench line.
# usage: bash tools/gpu_ab.sh <tag> <pytest|nopytest> <families> "ENV.." ...   ("-" = product build, no knob)
R=$GRAFT_REPO_ROOT; TAG=$1; PYT=$2; FAM=$3; shift; shift; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python tools/split_debug.py quick > $OUT/split_debug.log 2>&1; echo "split_debug rc=$?" >> $OUT/split_debug.log; tail -2 $OUT/split_debug.log
if [ "$PYT" = pytest ]; then timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log; fi
bash tools/gpu_knobs2.sh $TAG/k $FAM "$@"
