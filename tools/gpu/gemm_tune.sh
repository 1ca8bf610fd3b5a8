#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "pingpong" -p no:cacheprovider --tb=short > gpurun_out/pytest_phase.log 2>&1
rc=$?; echo "pytest phase rc=$rc" >> gpurun_out/pytest_phase.log; tail -12 gpurun_out/pytest_phase.log
timeout 300 python tools/gemm_sections.py > gpurun_out/sections.log 2>&1; cat gpurun_out/sections.log
timeout 300 python tools/gemm_ab.py 30 > gpurun_out/gemm_ab.log 2>&1; cat gpurun_out/gemm_ab.log
