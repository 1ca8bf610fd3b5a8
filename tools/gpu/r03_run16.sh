#!/bin/bash
# round 3, GPU call 16: engine option arena_reuse (block temporaries released on return): bit-identity test, same-box A/B on the C1 job
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "arena_reuse" > gpurun_out/pytest_gpu16.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu16.log; tail -5 gpurun_out/pytest_gpu16.log | cut -c1-300
timeout 600 python tools/gpu/knob_sweep.py base arena_reuse=1 --profile --out gpurun_out/knob_sweep_r03_16.json > gpurun_out/knob_sweep_r03_16.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_16.log | tail -6 | cut -c1-200
grep -E "^    (conv3x3|1x1 |1x1_geglu|groupnorm|layernorm|attention_mfma_self|1x1_vt)" gpurun_out/knob_sweep_r03_16.log | cut -c1-120
