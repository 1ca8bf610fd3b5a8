#!/bin/bash
# round 3, GPU call 9: occupancy probe of the level-0 attention kernel (extra LDS per workgroup), second tuner pass on C1 over the new
# table, tuner on the SDXL 1024x1024 and the hires-fix shapes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gpu/attn_time.py --variants 15 --lds-pads 0,8000,24000,56000,100000 --reps 3 --out gpurun_out/attn_time9.json > gpurun_out/attn_time9.log 2>&1; tail -5 gpurun_out/attn_time9.log | cut -c1-600
timeout 900 python tools/gpu/shape_tune.py --top 45 --emit --tag r03_shape_tuning_pass2 > gpurun_out/shape_tune9_c1.log 2>&1; tail -3 gpurun_out/shape_tune9_c1.log | cut -c1-300
timeout 900 python tools/gpu/shape_tune.py --top 40 --emit --model sdxl --size 1024 --batch 4 --sampler-steps 6 --reps 2 --tag r03_shape_tuning_sdxl > gpurun_out/shape_tune9_sdxl.log 2>&1; tail -3 gpurun_out/shape_tune9_sdxl.log | cut -c1-300
timeout 900 python tools/gpu/shape_tune.py --top 40 --emit --hires --sampler-steps 6 --reps 2 --tag r03_shape_tuning_hires > gpurun_out/shape_tune9_hires.log 2>&1; tail -3 gpurun_out/shape_tune9_hires.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "conv or gemm or split" > gpurun_out/pytest_gpu9.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu9.log; tail -3 gpurun_out/pytest_gpu9.log
