#!/bin/bash
# full GPU check: every gpu test + the bench line (no cpu baseline)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 4 -p no:cacheprovider --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -2 gpurun_out/bench.log | cut -c1-300
