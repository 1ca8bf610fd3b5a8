#!/bin/bash
# whole gpu suite + smoke on the final tree (after the last kernel-file edit)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu_last3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_last3.log; tail -3 gpurun_out/pytest_gpu_last3.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_last3.log 2>&1; tail -1 gpurun_out/smoke_last3.log
