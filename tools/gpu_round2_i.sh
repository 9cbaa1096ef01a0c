#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02i; mkdir -p $OUT
cd $R
FGX_PIPE=1 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
b() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  grep '^{' $OUT/$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.3g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))" || tail -5 $OUT/$name.log
}
b d8_pipe FGX_PIPE=1 --families 1000000
b d8_v2 FGX_PIPE=0 --families 1000000
b d5_pipe FGX_PIPE=1 --families 1000000 --depth 5
b d5_v2 FGX_PIPE=0 --families 1000000 --depth 5
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  FGX_PIPE=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p -o pmc$i -- python $R/bench.py --families 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_parse.py $OUT/p > $OUT/pmc_d8.json
python - $OUT/pmc_d8.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if k.startswith('k_simplex'): print(k, {c: round(x/1e6,1) for c,x in v.items()})
PY
rm -rf $OUT/p
