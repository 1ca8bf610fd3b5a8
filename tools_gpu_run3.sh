#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 60 -p no:cacheprovider --tb=short -rA > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_kernels.py gemm > gpurun_out/kernels.log 2>&1
timeout 900 python tools/bench_kernels.py split >> gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
cat gpurun_out/kernels.log | tail -40
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
