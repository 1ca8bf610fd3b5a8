#!/bin/bash
# round 2, GPU call: whole gpu suite (incl. the new boundary / C1 / transpose / NaN tests) + knob sweep on the C1 job
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 900 python tools/gpu/knob_sweep.py base vt_mode=0 attn_occ=4 attn_occ=5 attn_occ=6 gemm_shortk_cfg=9 \
  gemm_shortk_cfg=9,gemm_shortk_maxk=700 gemm_shortk_cfg=9,gemm_shortk_maxk=1300 gemm_shortk_cfg=7 gemm_shortk_cfg=7,gemm_shortk_maxk=1300 \
  gemm_geglu_cfg=0 gemm_geglu_cfg=3 --reps 3 --jobs 2 --profile > gpurun_out/knob_sweep.log 2>&1
echo "sweep rc=$?" >> gpurun_out/knob_sweep.log; grep -E "min|rc=" gpurun_out/knob_sweep.log | head -30
