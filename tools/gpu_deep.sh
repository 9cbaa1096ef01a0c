#!/bin/bash
# Deep simplex families on the split pipeline: the new GPU tests, then the long-tail shape's kernel trace.  usage: bash tools/gpu_deep.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_deep_families.py -m gpu -q -p no:cacheprovider -rfEs --timeout 600 > $OUT/pytest_deep.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_deep.log | head -40
bash tools/gpu_shapes.sh $TAG longtail
