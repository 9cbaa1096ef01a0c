#!/bin/bash
# Round 5: the two-phase device inflate (k_bgzf_tokenize + k_bgzf_resolve).  The pipeline / BGZF test files first, then the inflate alone on
# 150 000 depth-8 families (766 MB in 11 738 blocks, the sample of profiles/r04_experiments.md) with the one-phase kernel and the two-phase
# form at several lane counts per tokenizer workgroup, then (optional) the kernel trace of one two-phase pass.
# usage (via gpurun): bash tools/gpu_inflate.sh <tag> [tests|notests] [trace]
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
if [ "$2" != "notests" ]; then
  timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_bgzf.py -m gpu -q -x -p no:cacheprovider -rfE --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -8
fi
run() { local name=$1; shift; env "$@" timeout 300 python tools/bench_inflate.py --families 150000 --reps 5 > $OUT/$name.json 2> $OUT/$name.err; echo "$name: $(cut -c1-300 $OUT/$name.json) $(tail -1 $OUT/$name.err | cut -c1-200)"; }
run one_phase FGX_INFL_TWO_PHASE=0
for l in 8 16 32; do run two_phase_$l FGX_INFL_LANES=$l; done
run two_phase_lds_resolve FGX_INFL_RESOLVE_LDS=1
if [ "$3" = "trace" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o infl -- python $R/tools/bench_inflate.py --families 150000 --reps 3 > $OUT/trace.log 2>&1
  rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*/*_agent_info.csv $OUT/*/*kernel_trace.csv
  python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
fi
