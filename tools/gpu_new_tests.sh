#!/bin/bash
# The GPU tests added or changed in round 4, then the inflate microbench (regression check).  usage: bash tools/gpu_new_tests.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_gpu_direct_records.py tests/test_gpu_pipeline.py tests/test_gpu_methylation.py tests/test_gpu_zz_rejects_device.py -m gpu -q -p no:cacheprovider -rfEs --timeout 600 > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_new.log | head -40
python tools/bench_inflate.py --families 150000 --reps 3
