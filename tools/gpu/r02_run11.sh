#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=line -k "experiment_variants or geglu or pingpong" > gpurun_out/pytest_r11.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r11.log
timeout 900 python tools/gpu/knob_sweep.py base gemm_dbgflags=16384 attn_occ=15 attn_occ=19 attn_occ=15,gemm_dbgflags=16384 --profile --out gpurun_out/knob_sweep_r11.json > gpurun_out/knob_sweep_r11.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^attn_occ|^gemm_dbg|attention_mfma_self|conv3x3|    1x1 |geglu" gpurun_out/knob_sweep_r11.log | head -30
