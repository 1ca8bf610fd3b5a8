#!/bin/bash
# round 3, GPU call 5: in-launch ordered split-K reduction (bit-identity test + same-box A/B on the C1 job), PIL front-end test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "split or img2img or inpaint or conv_gemm or pingpong or phase" > gpurun_out/pytest_gpu5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu5.log; tail -8 gpurun_out/pytest_gpu5.log
timeout 600 python tools/gpu/knob_sweep.py base splitk_inkernel=1 --profile --out gpurun_out/knob_sweep_r03_5.json > gpurun_out/knob_sweep_r03_5.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_5.log | tail -6 | cut -c1-200
grep "splitk\|1x1  \|conv3x3" gpurun_out/knob_sweep_r03_5.log | head -12
