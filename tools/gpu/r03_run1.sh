#!/bin/bash
# round 3, GPU call 1: the whole GPU suite on the merged tree (index-walk norm / split-K kernels, LayerNorm fold test, role-offset attention,
# full-size parity fixtures), standalone attention timing, then the same-box knob sweep on the C1 job
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu/attn_time.py > gpurun_out/attn_time.log 2>&1; tail -8 gpurun_out/attn_time.log
timeout 600 python tools/gpu/knob_sweep.py base ln_fold=1 streams=2 attn_occ=20 attn_occ=21 ln_fold=1,attn_occ=21 ln_fold=1,attn_occ=21,streams=2 --profile --out gpurun_out/knob_sweep_r03_1.json > gpurun_out/knob_sweep_r03_1.log 2>&1
tail -120 gpurun_out/knob_sweep_r03_1.log | cut -c1-200
