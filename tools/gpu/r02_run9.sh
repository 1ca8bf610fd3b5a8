#!/bin/bash
# round 2: attention fragment-prefetch / pipelined variants: bit-identity test, standalone timing, in-job A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short -k "experiment_variants" > gpurun_out/pytest_attn9.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_attn9.log
timeout 300 python tools/gpu/attn_parts.py 5 0 15 16 17 8 > gpurun_out/attn_parts2.log 2>&1; echo "parts rc=$?"; grep -v amdgpu.ids gpurun_out/attn_parts2.log
timeout 600 python tools/gpu/knob_sweep.py base attn_occ=0 attn_occ=15 attn_occ=16 attn_occ=17 --profile --out gpurun_out/knob_sweep_attn15.json > gpurun_out/knob_sweep_attn15.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^attn_occ|attention_mfma_self" gpurun_out/knob_sweep_attn15.log | head -12
