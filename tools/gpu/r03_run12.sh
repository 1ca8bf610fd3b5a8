#!/bin/bash
# round 3, GPU call 12: streaming-rate probe (copy / torch layer_norm / engine layernorm at in-job and 8x sizes)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gpu/stream_probe.py > gpurun_out/stream_probe.log 2>&1; tail -6 gpurun_out/stream_probe.log | cut -c1-400
