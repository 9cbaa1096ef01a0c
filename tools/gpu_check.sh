#!/bin/bash
# usage: tools/gpu_check.sh <tag> [quick]   — GPU parity suite + the four simplex bench shapes (1M / 2M families)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
if [ "$2" = quick ]; then
  timeout 600 python -m pytest tests -m gpu -x -q -k "simplex or fast or vanilla or caller or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
else
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
fi
tail -6 $OUT/pytest.log
b() { local name=$1; shift
  timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  grep '^{' $OUT/$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'k_family_ms=%.2f k_emit_ms=%.2f ms_step=%.2f reads/s=%.3g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))" || tail -5 $OUT/$name.log
}
b d8 --families 1000000
b d3 --families 2000000 --depth 3
b d1 --families 2000000 --depth 1
b lt --families 1000000 --depth 2 --depth-max 50
