#!/bin/bash
# round 2: channel-block-major conv K order, old address rebuild (HEAD build) vs the per-source pointer + tap mask form, each against
# the tap-major order in its own process
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in libsdmi_old.so libsdmi.so; do
  SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 600 python tools/gpu/knob_sweep.py base conv_korder=0 --reps 2 --profile --out gpurun_out/knob_sweep_r15_$lib.$rep.json > gpurun_out/knob_sweep_r15_$lib.$rep.log 2>&1
  echo "$lib rep $rep rc=$?"; grep -E "^base|^conv_korder|conv3x3" gpurun_out/knob_sweep_r15_$lib.$rep.log | head -4
done
done
