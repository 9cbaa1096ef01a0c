#!/bin/bash
# per-phase cycle shares on the long-tail workload (profiling build variant_phase.so)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02lt; mkdir -p $OUT; cd $R
FGX_LIB=$R/fgumi_amd/variant_phase.so timeout 600 python bench.py --families 1000000 --depth 2 --depth-max 50 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v '^{' | tee $OUT/phase.txt
