#!/bin/bash
# round 3, GPU call 4: attention with the tile maximum as a v_max3 tree (variant 16 = 15 + tree; the role-offset kernels always), PIL front-end test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "attention or img2img or inpaint or layernorm_folded" > gpurun_out/pytest_gpu4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu4.log; tail -6 gpurun_out/pytest_gpu4.log
timeout 300 python tools/gpu/attn_time.py --variants 15,16,30,31 --out gpurun_out/attn_time4.json > gpurun_out/attn_time4.log 2>&1; tail -6 gpurun_out/attn_time4.log | cut -c1-420
timeout 600 python tools/gpu/knob_sweep.py base attn_occ=16 --out gpurun_out/knob_sweep_r03_4.json > gpurun_out/knob_sweep_r03_4.log 2>&1
tail -3 gpurun_out/knob_sweep_r03_4.log | cut -c1-200
