#!/bin/bash
# k_split_cols variants: the parity files that cover the split pipeline on the product library (incl. the full-size oracle comparison), then the
# headline bench line for several library builds, interleaved.  usage: bash tools/gpu_cols_variants.sh <tag> <rounds> v1 v2 ...
TAG=$1; ROUNDS=$2; shift; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_full_size.py tests/test_gpu_deep_families.py tests/test_gpu_indels.py -m gpu -q -x -p no:cacheprovider -rfE --timeout 500 > $OUT/pytest_cols.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_cols.log | head -20
bash tools/gpu_variants.sh $TAG $ROUNDS -- "$@"
