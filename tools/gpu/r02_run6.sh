#!/bin/bash
# round 2: attention variant 7 A/B, then the whole GPU suite on the code as it stands
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu/knob_sweep.py base attn_occ=7 --profile --out gpurun_out/knob_sweep_attn7.json > gpurun_out/knob_sweep_attn7.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^attn_occ|attention_mfma_self" gpurun_out/knob_sweep_attn7.log | head -10
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; grep -E "^\[c1|passed|failed|rc=|Error|assert" gpurun_out/pytest_gpu.log | tail -20
