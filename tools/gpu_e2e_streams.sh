#!/bin/bash
# the bench line's file -> file leg (ring form of fgx_run_bam, FGX_PIPE_RING=1) in the bench process itself (after the 5 M-family steps: the caller's second stream exists, its buffers are
# large), with the compute streams at normal / highest priority and with eight hardware queues.  usage: bash tools/gpu_e2e_streams.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { local name=$1; shift; env FGX_PIPE_RING=1 "$@" timeout 300 python bench.py --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); e=d['end_to_end']
print(sys.argv[2], 'value=%.4g ms_step=%.2f | e2e %.1f M/s total %.3f s chunks %d busy'%(d['value'], d['ms_per_step'], e['value']/1e6, e['total_s'], e['chunks']), {k: round(v,3) for k,v in e['stage_busy_s'].items()}, {k: round(v,3) for k,v in e['device_stage_s'].items()})
PY
}
run normal FGX_STREAM_PRIORITY=0
run high FGX_STREAM_PRIORITY=1
run high_q8 FGX_STREAM_PRIORITY=1 GPU_MAX_HW_QUEUES=8
run normal_q8 FGX_STREAM_PRIORITY=0 GPU_MAX_HW_QUEUES=8
