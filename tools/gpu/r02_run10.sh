#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=line -k "experiment_variants" > gpurun_out/pytest_attn10.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_attn10.log
timeout 300 python tools/gpu/attn_parts.py sections > gpurun_out/attn_sections.log 2>&1; echo "sections rc=$?"; grep -v amdgpu.ids gpurun_out/attn_sections.log
timeout 300 python tools/gpu/attn_parts.py 5 15 18 > gpurun_out/attn_parts3.log 2>&1; grep -v amdgpu.ids gpurun_out/attn_parts3.log
