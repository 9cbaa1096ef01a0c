#!/bin/bash
# the whole GPU suite on the product library, then the headline bench line for several library builds, interleaved.
# usage: bash tools/gpu_suite_variants.sh <tag> <rounds> v1 v2 ...
TAG=$1; ROUNDS=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
bash tools/gpu_variants.sh $TAG $ROUNDS -- "$@"
