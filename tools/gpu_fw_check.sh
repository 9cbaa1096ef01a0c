#!/bin/bash
# k_family_wave check: duplex / CODEC / indel / crafted parity tests + the bench lines its three modes dominate; usage: tools/gpu_fw_check.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests -m gpu -x -q -k "duplex or codec or indel or crafted or option_matrix or large_families" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
run() { local tag=$1; shift
  timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'k_family_ms=%.3f k_emit_ms=%.2f ms_step=%.2f def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['config']['deferred_families']))" || tail -3 $OUT/$tag.log
}
run duplex --caller duplex
run codec --caller codec
run lt --families 1000000 --depth 2 --depth-max 50
run d8 --families 1000000
