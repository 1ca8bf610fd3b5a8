#!/bin/bash
# round 2: 16-byte epilogue A/B on the C1 job, then the tests that changed
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short -x -k "wide or generic_and_mfma or dpm_adaptive or pingpong or transposed or epilogues or geglu" -s > gpurun_out/pytest_wide.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|rc=|Error|assert|e2e dpm" gpurun_out/pytest_wide.log | tail -12
timeout 600 python tools/gpu/knob_sweep.py base ep_wide=0 --profile --out gpurun_out/knob_sweep_epwide.json > gpurun_out/knob_sweep_epwide.log 2>&1
echo "sweep rc=$?"; grep -E "^base|^ep_wide|    (1x1|conv3x3|1x1_geglu|1x1_vt) " gpurun_out/knob_sweep_epwide.log | head -12
