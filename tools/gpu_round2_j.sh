#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02j; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
b() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  grep '^{' $OUT/$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.3g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))" || tail -5 $OUT/$name.log
}
b d8 A=1 --families 1000000
b d3 A=1 --families 2000000 --depth 3
b d1 A=1 --families 2000000 --depth 1
b lt A=1 --families 1000000 --depth 2 --depth-max 50
