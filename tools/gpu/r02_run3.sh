#!/bin/bash
# round 2, GPU call: whole gpu suite + knob sweep of the round-2 defaults against their round-1 settings
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; grep -E "passed|failed|FAILED|transpose|c1 |c3 " gpurun_out/pytest_gpu.log | tail -40
timeout 900 python tools/gpu/knob_sweep.py base conv_korder=0 tile_order=0 attn_occ=0 vt_mode=0 \
  conv_korder=0,tile_order=0,attn_occ=0,vt_mode=0 --reps 3 --jobs 2 --profile > gpurun_out/knob_sweep2.log 2>&1
echo "sweep rc=$?" >> gpurun_out/knob_sweep2.log; grep -E "min|rc=" gpurun_out/knob_sweep2.log | head -30
cp gpurun_out/knob_sweep.json gpurun_out/knob_sweep2.json
