#!/bin/bash
# round 3, GPU call 2: GPU suite (three-group attention, single-launch small GroupNorm, LayerNorm fold with hoisted row statistics, sharded job
# with the real engine), attention timing, knob sweep, a short bench line with the drop-in path, then the FULL parity run (live oracle on all rows)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 300 python tools/gpu/attn_time.py --variants 15,20,21,30,31 --out gpurun_out/attn_time2.json > gpurun_out/attn_time2.log 2>&1; tail -6 gpurun_out/attn_time2.log | cut -c1-400
timeout 700 python tools/gpu/knob_sweep.py base gn_small=0 ln_fold=1 attn_occ=30 attn_occ=31 ln_fold=1,attn_occ=31 --profile --out gpurun_out/knob_sweep_r03_2.json > gpurun_out/knob_sweep_r03_2.log 2>&1
grep -v "^    " gpurun_out/knob_sweep_r03_2.log | tail -16 | cut -c1-200
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r03_2.log 2>&1; tail -1 gpurun_out/bench_r03_2.log | cut -c1-1500
SDMI_PARITY_FULL=1 timeout 1500 python -m pytest tests/test_gpu_c1_parity.py -m gpu -q -p no:cacheprovider --tb=short --timeout 1400 > gpurun_out/pytest_parity_full.log 2>&1
echo "parity full rc=$?" >> gpurun_out/pytest_parity_full.log; tail -5 gpurun_out/pytest_parity_full.log
