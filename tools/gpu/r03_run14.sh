#!/bin/bash
# round 3, GPU call 14: the 77-token text context as one 96-key tile (knob attn_kvt = 96): test, then same-box A/B on the C1 job
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "96_key or attention" > gpurun_out/pytest_gpu14.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu14.log; tail -5 gpurun_out/pytest_gpu14.log | cut -c1-300
timeout 600 python tools/gpu/knob_sweep.py base attn_kvt=96 --profile --out gpurun_out/knob_sweep_r03_14.json > gpurun_out/knob_sweep_r03_14.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_14.log | tail -6 | cut -c1-200; grep "attention_mfma_cross" gpurun_out/knob_sweep_r03_14.log | head -4
