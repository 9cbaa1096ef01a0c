#!/bin/bash
# The first GPU call after round 3: the hardware verdict on the opt-in paths that were written and proved on the CPU after
# the round's GPU budget was spent.  One gpurun call, ≈ 10 minutes:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/gpu_opt_in_first_call.sh r04a'
# Writes gpurun_out/<tag>_*: copy what is to be judged into profiles/.
set -u
TAG="${1:-r04a}"
OUT="gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. the opt-in paths' own GPU tests (xfail non-strict: -rxX lists what XPASSed and what did not, with the child's output)
timeout 900 python -m pytest tests/test_gpu_zz_*.py tests/test_gpu_duplex_canon.py -m gpu -q -rxX -p no:cacheprovider > "$OUT/${TAG}_opt_in_tests.txt" 2>&1
echo "opt-in tests: rc=$?" | tee -a "$OUT/${TAG}_summary.txt"
tail -3 "$OUT/${TAG}_opt_in_tests.txt" | tee -a "$OUT/${TAG}_summary.txt"
# 2. the whole GPU suite with every opt-in path switched on
FGX_OPT_IN_ALL=1 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/${TAG}_suite_all_on.txt" 2>&1
echo "whole suite, FGX_OPT_IN_ALL=1: rc=$?" | tee -a "$OUT/${TAG}_summary.txt"
tail -3 "$OUT/${TAG}_suite_all_on.txt" | tee -a "$OUT/${TAG}_summary.txt"
# 3. every opt-in path beside the path it replaces (bytes compared), then the lane-per-item kernels under a kernel trace
timeout 600 python tools/bench_opt_in_paths.py --molecules 20000 --steps 3 > "$OUT/${TAG}_opt_in_paths.jsonl" 2> "$OUT/${TAG}_opt_in_paths.err"
echo "bench_opt_in_paths: rc=$?" | tee -a "$OUT/${TAG}_summary.txt"
( cd /tmp && FGX_OPT_IN_ALL=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/${TAG}_trace" -- python "$OLDPWD/tools/bench_opt_in_paths.py" --molecules 20000 --steps 2 > /dev/null 2>&1 )
find "$OUT/${TAG}_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_opt_in_kernel_stats.csv" \; 2>/dev/null
echo done | tee -a "$OUT/${TAG}_summary.txt"
