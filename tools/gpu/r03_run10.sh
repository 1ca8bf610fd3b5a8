#!/bin/bash
# round 3, GPU call 10: variant 17 of the level-0 attention kernel (shift folded into the S^T MFMA on the production 4-wave kernel)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "role_offset or pil_front_end" > gpurun_out/pytest_gpu10.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu10.log; tail -5 gpurun_out/pytest_gpu10.log
timeout 600 python tools/gpu/attn_time.py --variants 15,17 --reps 5 --out gpurun_out/attn_time10.json > gpurun_out/attn_time10.log 2>&1; tail -5 gpurun_out/attn_time10.log | cut -c1-400
timeout 600 python tools/gpu/knob_sweep.py base attn_occ=17 --profile --out gpurun_out/knob_sweep_r03_10.json > gpurun_out/knob_sweep_r03_10.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_10.log | tail -6 | cut -c1-200
