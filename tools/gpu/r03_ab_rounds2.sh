#!/bin/bash
# per-kernel profiles of the round-2 and round-3 libraries on one box (which launches differ: the transposed-output GEMMs ran 28.8 us in
# round 2's profile and 47-52 us in every round-3 one)
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in libsdmi_r02.so libsdmi.so libsdmi_r02.so libsdmi.so; do
  SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 300 python tools/gpu/knob_sweep.py base --reps 2 --profile --out gpurun_out/ab_profile_$lib.$RANDOM.json > gpurun_out/ab_profile_$lib.log 2>&1
  echo "$lib rc=$?"; grep -E "^base|_tr M|1x1_vt" gpurun_out/ab_profile_$lib.log | head -8 | cut -c1-200
done
ls gpurun_out/ab_profile_*
