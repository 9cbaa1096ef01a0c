cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02z; mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q -k "simplex or fast or vanilla or caller or golden or schedule or device_resident or crafted" 2>&1 | tail -3
for r in 1 2; do
for mode in 1 0; do
  for cfg in "d8 --families 1000000" "d3 --families 2000000 --depth 3" "lt --families 1000000 --depth 2 --depth-max 50"; do
    set -- $cfg; name=$1; shift
    FGX_PERSISTENT=$mode timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${name}_p$mode.log 2>&1
    grep '^{' $OUT/${name}_p$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent=$mode $name', 'k_family_ms=%.3f k_emit_ms=%.2f ms_step=%.2f def=%s'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['ms_per_step'], d['config']['deferred_families']))" || tail -5 $OUT/${name}_p$mode.log
  done
done
done
FGX_LIB=$PWD/fgumi_amd/variant_base.so timeout 300 python bench.py --families 1000000 --steps 5 --warmup 1 --no-cpu-baseline | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base d8', 'k_family_ms=%.3f'%(d['roofline']['kernel_ms']))"
