#!/bin/bash
# Duplex / CODEC: their GPU tests (incl. the full-size oracle comparison), then the kernel traces of both shapes.  usage: bash tools/gpu_duplex.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_duplex.py tests/test_gpu_codec.py tests/test_gpu_duplex_canon.py tests/test_gpu_zz_codec_canon.py tests/test_gpu_oracle_full_size.py -m gpu -q -p no:cacheprovider -rfEs --timeout 900 > $OUT/pytest_duplex.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_duplex.log | head -30
bash tools/gpu_shapes.sh $TAG duplex codec
