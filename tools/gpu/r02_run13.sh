#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gemm_sections.py > gpurun_out/sections3.log 2>&1; echo "sections rc=$?"; grep -v amdgpu.ids gpurun_out/sections3.log
