#!/bin/bash
# k_simplex_wave2 workgroup size A/B (FGX_W2_WPB) on the depth-8 and long-tail shapes; usage: tools/gpu_wpb.sh <tag> [variants...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R; shift
run() { local tag=$1; shift
  for cfg in "d8 --families 1000000" "lt --families 1000000 --depth 2 --depth-max 50"; do
    set -- $cfg; name=$1; shift
    timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${name}_$tag.log 2>&1
    grep '^{' $OUT/${name}_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag $name', 'k_family_ms=%.3f ms_step=%.2f def=%s'%(d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['deferred_families']))" || tail -5 $OUT/${name}_$tag.log
  done
}
timeout 300 python -m pytest tests -m gpu -x -q -k "simplex or fast or vanilla or caller or golden or schedule or device_resident or crafted" 2>&1 | tail -2
for r in 1 2; do
  for w in 4 3 2; do FGX_W2_WPB=$w run wpb$w; done
  for v in "$@"; do FGX_LIB=$R/fgumi_amd/variant_$v.so run $v; done
done
