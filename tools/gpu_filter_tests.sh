#!/bin/bash
# `fgumi filter` on the device: its GPU tests, then its rate (tools/bench_filter.py).  usage: bash tools/gpu_filter_tests.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_filter.py -m gpu -q -p no:cacheprovider -rfEs --timeout 600 > $OUT/pytest_filter.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_filter.log | head -40
timeout 600 python tools/bench_filter.py > $OUT/bench_filter.log 2>&1; tail -8 $OUT/bench_filter.log
