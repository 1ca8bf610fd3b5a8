#!/bin/bash
# round 2: per-phase section timers of the ping-pong GEMM, then the whole GPU suite on the code as it stands
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gemm_sections.py > gpurun_out/sections2.log 2>&1; echo "sections rc=$?"; grep -v amdgpu.ids gpurun_out/sections2.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; grep -E "^\[c1|^\[c3|passed|failed|rc=|Error|assert" gpurun_out/pytest_gpu.log | tail -20
