#!/bin/bash
# first GPU pass: parity tests (crash-isolated through xdist) + a short bench
export TMPDIR=/tmp
mkdir -p gpurun_out
(rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; nproc; free -g | head -2) > gpurun_out/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 60 -p no:cacheprovider --tb=short -rA > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
