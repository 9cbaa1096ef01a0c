#!/bin/bash
# k_emit check: the simplex parity subset (every record goes through k_emit) + bench lines; usage: tools/gpu_emit_check.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests -m gpu -x -q -k "simplex or fast or vanilla or caller or golden or schedule or device_resident or crafted" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
run() { local tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | tail -1 > $OUT/${tag}_bench_line.json
  python -c "import sys,json; d=json.load(open('$OUT/${tag}_bench_line.json')); print('$tag', 'k_family_ms=%.3f k_emit_ms=%.2f ms_step=%.2f reads/s=%.4g def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['value'], d['config']['deferred_families']))" || tail -3 $OUT/$tag.log
}
run default
run depth3 --families 5000000 --depth 3 --steps 5 --warmup 1
run depth1 --families 5000000 --depth 1 --steps 5 --warmup 1
run longtail --families 1000000 --depth 2 --depth-max 50 --steps 5 --warmup 1
