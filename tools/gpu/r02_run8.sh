#!/bin/bash
# round 2: VALU issue rates + attention component timings (library built with -DSDMI_ATTN_PARTS)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/micro/valu_rates > gpurun_out/valu_rates.log 2>&1; echo "valu rc=$?"; cat gpurun_out/valu_rates.log
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short -k "experiment_variants" > gpurun_out/pytest_attn8.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_attn8.log
timeout 300 python tools/gpu/attn_parts.py 5 10 11 12 13 14 0 8 9 > gpurun_out/attn_parts.log 2>&1; echo "parts rc=$?"; grep -v amdgpu.ids gpurun_out/attn_parts.log
