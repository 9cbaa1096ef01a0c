#!/bin/bash
# End-of-round measurement refresh (on the GPU box, via gpurun): bench lines of the three callers + long tail with rocprofv3 kernel
# stats, filter and grouping tool lines.  usage: bash tools/refresh_round.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${name}_stats -- python $R/bench.py "$@" --no-cpu-baseline > $OUT/${name}_bench.log 2>&1
  grep '^{' $OUT/${name}_bench.log | tail -1 > $OUT/${name}_bench_line.json
}
run simplex --families 5000000 --steps 5 --warmup 1
run duplex --caller duplex --steps 3 --warmup 1
run codec --caller codec --steps 3 --warmup 1
run longtail --families 1000000 --depth 2 --depth-max 50 --steps 3 --warmup 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o filter_stats -- python $R/tools/bench_filter.py > $OUT/filter.log 2>&1
grep '^{' $OUT/filter.log | tail -1 > $OUT/filter_line.json
python $R/tools/bench_filter.py --caller duplex --families 500000 2>/dev/null | grep '^{' >> $OUT/filter_line.json
python $R/tools/bench_filter.py --caller codec --families 500000 2>/dev/null | grep '^{' >> $OUT/filter_line.json
python $R/tools/bench_grouping.py 2>/dev/null | grep '^{' > $OUT/grouping_line.json
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
ls -la $OUT
