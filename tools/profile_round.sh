#!/bin/bash
# Full-size measurement round (on the GPU box, via gpurun): parity suite, the four PMC passes of the bench workload (parsed into
# profiles/<tag>_pmc_5M_families.json BEFORE the bench line is taken, so the line cites the counters of this build), the default
# bench line (+ CPU baseline), the other BASELINE shapes, phase shares at depth 8 / 3 / 1, rocprofv3 kernel stats.
# usage: bash tools/profile_round.sh <tag>        results land in gpurun_out/<tag>/ ; copy the summaries to profiles/
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -rfE --timeout 600 > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
# (round 5: a box whose first GPU test died with a device memory fault went on to burn 25 GPU-minutes in timeouts — nothing below is worth running then)
[ $rc -ne 0 ] && { echo "GPU suite failed (rc=$rc): the measurement round stops here"; exit 1; }
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc -o pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-block > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_parse.py $OUT/pmc 3 > $OUT/pmc_5M_families.json
cp $OUT/pmc_5M_families.json $R/profiles/${TAG}_pmc_5M_families.json
rm -rf $OUT/pmc
cd $R
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
line() { local name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $OUT/$name.log 2>&1; grep '^{' $OUT/$name.log | tail -1 > $OUT/${name}_bench_line.json
  python -c "import sys,json; d=json.load(open('$OUT/${name}_bench_line.json')); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.4g'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value']))"; }
line depth3 --families 5000000 --depth 3 --steps 5 --warmup 1
line depth1 --families 5000000 --depth 1 --steps 5 --warmup 1
line longtail --families 1000000 --depth 2 --depth-max 50 --steps 5 --warmup 1
line longtail_5M --families 5000000 --depth 2 --depth-max 50 --steps 3 --warmup 1
line duplex --caller duplex --steps 3 --warmup 1
line codec --caller codec --steps 3 --warmup 1
line strong_n1 --scaling strong --steps 3 --warmup 1
for d in 8 3 1; do
  FGX_LIB=$R/fgumi_amd/variant_phase.so timeout 300 python bench.py --families 1000000 --depth $d --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "phase share" | sed "s/^/depth $d: /" >> $OUT/phase_share.txt
done
cat $OUT/phase_share.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o simplex -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rm -rf $OUT/*_agent_info.csv $OUT/*kernel_trace.csv
python - $OUT <<'PY'
import csv,glob,sys,json
for f in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:7]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
d=json.load(open(sys.argv[1]+'/pmc_5M_families.json'))
for k,v in d.items():
    if k.startswith('k_split') or k=='k_emit' or k=='k_call_full': print(k, {c: round(x*v.get('_launches_per_step',1)/5e6,1) for c,x in v.items() if not c.startswith('_')})
PY
ls $OUT
