#!/bin/bash
# usage: bash tools/gpu_meth_bench.sh <tag> — methylation / deep tests, then the device methylation bench under rocprofv3 (kernel stats)
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_methylation.py tests/test_gpu_deep_families.py -m gpu -q -p no:cacheprovider -rfEs --timeout 900 > $OUT/pytest_meth.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_meth.log | head -40
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o meth -- python $R/tools/bench_methylation_device.py --steps 3 > $OUT/meth_bench.log 2>&1
grep '^{' $OUT/meth_bench.log | cut -c1-900
python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/meth_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        if 'sim_generate' in r['Name']: continue
        print('  ', r['Name'][:80], r['Calls'], 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'total %.1f ms' % (float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv $OUT/*domain_stats.csv
