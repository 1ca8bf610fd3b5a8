#!/bin/bash
# round 2: GroupNorm-statistics fusion — parity at the bench shapes, then the same-process A/B with per-kernel HIP-event groups
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_c1_parity.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short -s > gpurun_out/pytest_gn.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gn.log; grep -E "^\[c1|passed|failed|rc=|Error|assert" gpurun_out/pytest_gn.log | tail -20
timeout 900 python tools/gpu/knob_sweep.py base gn_fuse=0 --profile --out gpurun_out/knob_sweep_gn.json > gpurun_out/knob_sweep_gn.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^gn_fuse|^---|^    " gpurun_out/knob_sweep_gn.log | head -40
