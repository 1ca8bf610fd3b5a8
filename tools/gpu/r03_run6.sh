#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=1 timeout 300 python tools/gpu/debug_frontend.py > gpurun_out/debug_frontend.log 2>&1; echo "rc=$?" >> gpurun_out/debug_frontend.log; tail -30 gpurun_out/debug_frontend.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "split or layernorm or groupnorm" > gpurun_out/pytest_gpu6.log 2>&1; tail -4 gpurun_out/pytest_gpu6.log
