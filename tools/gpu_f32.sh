#!/bin/bash
# f32 hot sums in k_split_cols: parity tests that cover the split pipeline (incl. the full-size oracle comparison), then the headline bench line
# for the product library and the f64 variant.  usage: bash tools/gpu_f32.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle_full_size.py tests/test_gpu_deep_families.py tests/test_gpu_direct_records.py tests/test_gpu_full_size.py tests/test_gpu_indels.py -m gpu -q -p no:cacheprovider -rfE --timeout 900 > $OUT/pytest_f32.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_f32.log | head -20
bash tools/gpu_variants.sh $TAG 2 -- main f64
