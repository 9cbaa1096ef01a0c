#!/bin/bash
# The short measurement round at the end of a development round (when the whole GPU suite has run on the same kernels already): the duplex /
# CODEC test files, the four PMC passes of the bench workload (-> profiles/<tag>_pmc_5M_families.json, BEFORE the bench line is taken: the line
# cites the counters of this build), the default bench line, rocprofv3 kernel stats, the other BASELINE shapes.
# usage: bash tools/final_round.sh <tag>        results land in gpurun_out/<tag>/ ; copy the summaries to profiles/
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_duplex.py tests/test_gpu_codec.py tests/test_gpu_duplex_canon.py tests/test_gpu_zz_canon_device.py tests/test_gpu_zz_codec_canon.py -m gpu -q -p no:cacheprovider -rfE > $OUT/pytest_emit.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_emit.log
tail -2 $OUT/pytest_emit.log
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc -o pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_parse.py $OUT/pmc 3 > $OUT/pmc_5M_families.json && cp $OUT/pmc_5M_families.json $R/profiles/${TAG}_pmc_5M_families.json
rm -rf $OUT/pmc
cd $R
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
cd $R
line() { local name=$1; shift; timeout 200 python bench.py "$@" --no-cpu-baseline > $OUT/$name.log 2>&1; grep '^{' $OUT/$name.log | tail -1 > $OUT/${name}_bench_line.json
  python -c "import sys,json; d=json.load(open('$OUT/${name}_bench_line.json')); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.4g'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value']))"; }
line duplex --caller duplex --steps 3 --warmup 1
line codec --caller codec --steps 3 --warmup 1
line longtail --families 1000000 --depth 2 --depth-max 50 --steps 5 --warmup 1
line depth3 --families 5000000 --depth 3 --steps 5 --warmup 1
python - $OUT <<'PY'
import csv,glob,sys,json
for f in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
d=json.load(open(sys.argv[1]+'/pmc_5M_families.json'))
for k,v in d.items():
    if k.startswith('k_split') or k=='k_emit' or k=='k_call_full': print(k, {c: round(x*v.get('_launches_per_step',1)/5e6,1) for c,x in v.items() if not c.startswith('_')})
PY
