#!/bin/bash
# round 3, GPU call 18: full-size parity + attention tests with the long-sequence default (folded shift at N, M >= 8192), c4a line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --tb=short --timeout 900 -k "fullsize or c4 or c3 or c2 or vae or attention" > gpurun_out/pytest_gpu18.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu18.log; tail -4 gpurun_out/pytest_gpu18.log | cut -c1-300
timeout 900 python bench.py --config c4a --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c4a_fold.log 2>&1; tail -1 gpurun_out/bench_c4a_fold.log | cut -c1-300
