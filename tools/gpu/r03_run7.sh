#!/bin/bash
# round 3, GPU call 7: PIL front-end replay after the x8-VAE fix, then every gpu test and a bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/gpu/debug_frontend.py > gpurun_out/debug_frontend.log 2>&1; echo "rc=$?" >> gpurun_out/debug_frontend.log
tail -5 gpurun_out/debug_frontend.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu7.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu7.log; tail -8 gpurun_out/pytest_gpu7.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench7.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench7.log; tail -2 gpurun_out/bench7.log | cut -c1-400
