#!/bin/bash
# round 3, GPU call 13: front-end test after the batch-shrink / infotext alignment
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "pil_front_end or inpaint or img2img" > gpurun_out/pytest_gpu13.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu13.log; tail -12 gpurun_out/pytest_gpu13.log | cut -c1-300
