#!/bin/bash
# round 3, GPU call 3: GPU suite (LayerNorm-fold epilogue with hoisted per-column vectors, churn / UniPC vary_coeff / network bias tests, sharded job),
# section timers of the role-offset attention kernel, knob sweep base vs ln_fold
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu/attn_pp_sections.py > gpurun_out/attn_pp_sections.log 2>&1; tail -12 gpurun_out/attn_pp_sections.log | cut -c1-300
timeout 600 python tools/gpu/knob_sweep.py base ln_fold=1 --profile --out gpurun_out/knob_sweep_r03_3.json > gpurun_out/knob_sweep_r03_3.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_3.log | tail -8 | cut -c1-200
