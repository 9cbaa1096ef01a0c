#!/bin/bash
# occupancy A/B for k_simplex_wave2: builds with a VGPR cap for 6 wavefronts per SIMD need smaller LDS slices (FGX_WAVE_BYTES); usage: tools/gpu_occ.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
run() { local tag=$1; shift
  timeout 300 python bench.py --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/d8_$tag.log 2>&1
  grep '^{' $OUT/d8_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag d8', 'k_family_ms=%.3f ms_step=%.2f def=%s'%(d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['deferred_families']))" || tail -5 $OUT/d8_$tag.log
}
for r in 1 2; do
  run main
  FGX_WAVE_BYTES=5600 run main_5600
  for v in o6 o6b1; do
    for w in 3 4; do FGX_LIB=$R/fgumi_amd/variant_$v.so FGX_WAVE_BYTES=5600 FGX_W2_WPB=$w run ${v}_w$w; done
  done
done
