#!/bin/bash
# one-off A/B of launch-shape knobs (workgroup sizes of k_simplex_seg / k_family_wave, slice size of k_simplex_wave2); usage: tools/gpu_knobs.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
run() { local tag=$1; shift
  timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'k_family_ms=%.3f k_emit_ms=%.2f ms_step=%.2f def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['config']['deferred_families']))" || tail -3 $OUT/$tag.log
}
for r in 1 2; do
  run d8_default --families 1000000
  FGX_WAVE_BYTES=5632 run d8_5632 --families 1000000
  for w in 4 3 2; do FGX_SEG_WPB=$w run d3_segwpb$w --families 2000000 --depth 3; done
  for w in 4 3 2; do FGX_FW_WPB=$w run duplex_fw$w --caller duplex; FGX_FW_WPB=$w run codec_fw$w --caller codec; done
  for w in 4 3; do FGX_FW_WPB=$w run lt_fw$w --families 1000000 --depth 2 --depth-max 50; done
done
