#!/bin/bash
# Methylation-aware mode on the streaming kernels: its GPU tests.  usage: bash tools/gpu_meth.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_methylation.py tests/test_gpu_deep_families.py -m gpu -q -p no:cacheprovider -rfEs --timeout 900 -x > $OUT/pytest_meth.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_meth.log | head -40
