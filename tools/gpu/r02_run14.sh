#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_c1_parity.py -m gpu -q -p no:cacheprovider --tb=short -k "pingpong or conv or circular or split or c1_unet or c1_vae or groupnorm_statistics" > gpurun_out/pytest_r14.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r14.log
timeout 900 python tools/gpu/knob_sweep.py base conv_korder=0 --profile --out gpurun_out/knob_sweep_r14.json > gpurun_out/knob_sweep_r14.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^conv_korder|conv3x3|    1x1 " gpurun_out/knob_sweep_r14.log | head -12
