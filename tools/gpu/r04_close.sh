#!/bin/bash
# round 4, closing pass on the final tree: the whole GPU suite (SDMI_PARITY_FULL=1), smoke(), then the driver's bench command
export TMPDIR=/tmp
mkdir -p gpurun_out
SDMI_PARITY_FULL=1 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu_close.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_close.log; tail -3 gpurun_out/pytest_gpu_close.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_close.log 2>&1; tail -1 gpurun_out/smoke_close.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_close.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_close.log; tail -2 gpurun_out/bench_close.log | cut -c1-1200
