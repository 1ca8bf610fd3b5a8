#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 4 -p no:cacheprovider --tb=short -k "${1:-lincomb or end_to_end or scheduler_choice or inpainting or registry or img2img or lora}" > gpurun_out/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -25 gpurun_out/pytest_new.log
