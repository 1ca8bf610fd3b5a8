#!/bin/bash
# First GPU call of round 5 (no `import torch`: ~10 s of box time, ~45 s of lease): what the vendor library reaches on the engine's GEMM
# shapes (tools/micro/hipblaslt_yardstick — build it here first with the hipcc line in its header: the binary travels with the snapshot), and the forward's per-launch-shape profile of
# this tree through the torch-free harness.  Results under gpurun_out/; copy what is kept into profiles/.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/micro/hipblaslt_yardstick 30 > gpurun_out/hipblaslt_yardstick.txt 2> gpurun_out/hipblaslt_yardstick.err; echo "yardstick rc=$?"
cat gpurun_out/hipblaslt_yardstick.txt; tail -3 gpurun_out/hipblaslt_yardstick.err
timeout 120 python tools/gpu/fwd_ab.py base --reps 4 --fwd 10 --profile --out gpurun_out/fwd_ab_base.json > gpurun_out/fwd_ab_base.log 2>&1; echo "fwd_ab rc=$?"
grep -v "^    " gpurun_out/fwd_ab_base.log | tail -5
