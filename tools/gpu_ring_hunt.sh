#!/bin/bash
# The hunt for the ring pipeline's failure (~ 2 GPU-minutes): tests/test_gpu_pipeline.py's multi-chunk cases with
# FGX_PIPE_RING=1 and eight hardware queues, the ring brought back to pipeline.cpp's behaviour one knob at a time.  The first combination that
# turns red names the difference that matters.   usage: bash tools/gpu_ring_hunt.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
K="bam_file_to_consensus_bam_file_simplex or leftover_larger or several_chunks"
run() { local name=$1; shift; env FGX_PIPE_RING=1 GPU_MAX_HW_QUEUES=8 "$@" timeout 120 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider --runxfail -k "$K" > $OUT/$name.log 2>&1
  echo "$name: $(tail -1 $OUT/$name.log)"; }
run like_round3        FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=0 FGX_PIPE_ONE_STREAM=1     # one fill at a time, one fill stream: pipeline.cpp with five buffers
run own_streams        FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=0 FGX_PIPE_ONE_STREAM=0     # + a stream per buffer
run poll_ahead         FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=1 FGX_PIPE_ONE_STREAM=1     # + the next fill may start while the current one runs (same stream: still one after the other on the device)
run two_fills_at_once  FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=1 FGX_PIPE_ONE_STREAM=0     # + on its own stream: two inflate kernels side by side
run own_streams_wait  FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=0 FGX_PIPE_ONE_STREAM=0 FGX_PIPE_WAIT_EVENT=1   # the failing step + the compute stream waiting for the fill's event on the device
run own_streams_pinned FGX_PIPE_AHEAD=1 FGX_PIPE_POLL_AHEAD=0 FGX_PIPE_ONE_STREAM=0 FGX_PIPE_PINNED_TABLE=1  # the failing step + the block table through pinned memory
run four_ahead_wait    FGX_PIPE_AHEAD=4 FGX_PIPE_POLL_AHEAD=1 FGX_PIPE_ONE_STREAM=0 FGX_PIPE_WAIT_EVENT=1
run four_ahead_1stream FGX_PIPE_AHEAD=4 FGX_PIPE_POLL_AHEAD=1 FGX_PIPE_ONE_STREAM=1
run four_ahead         FGX_PIPE_AHEAD=4 FGX_PIPE_POLL_AHEAD=1 FGX_PIPE_ONE_STREAM=0
env GPU_MAX_HW_QUEUES=8 timeout 120 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -k "bam_file_to_consensus_bam_file_simplex or leftover_larger" > $OUT/default_form.log 2>&1; echo "default form (pipeline.cpp), eight queues: $(tail -1 $OUT/default_form.log)"
