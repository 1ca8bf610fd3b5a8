#!/bin/bash
# round-4 check pass: standalone kernel checks, the whole GPU suite (incl. the new full-size composition fixtures), then the same-box
# A/B of the 3x3 K walks on the C1 job (row-shared, the default, against tap-major)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/micro/conv_check 20 > gpurun_out/r04_conv_check.log 2>&1; echo "rc=$?" >> gpurun_out/r04_conv_check.log; tail -2 gpurun_out/r04_conv_check.log
timeout 1200 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 4 -p no:cacheprovider --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu/knob_sweep.py base conv_korder=0 --profile --out gpurun_out/r04_knob_dx.json > gpurun_out/r04_knob_dx.log 2>&1
grep -v "^    " gpurun_out/r04_knob_dx.log | head -12
