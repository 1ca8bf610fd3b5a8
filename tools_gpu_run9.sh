#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "norm" -p no:cacheprovider --tb=short > gpurun_out/pytest_norm.log 2>&1
rc=$?; echo "pytest norm rc=$rc" >> gpurun_out/pytest_norm.log; tail -8 gpurun_out/pytest_norm.log
[ $rc -ne 0 ] && exit 0
timeout 300 python tools/bench_kernels.py norm > gpurun_out/kernels_norm.log 2>&1; cat gpurun_out/kernels_norm.log
timeout 900 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 4 -p no:cacheprovider --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -2 gpurun_out/bench.log | cut -c1-300
