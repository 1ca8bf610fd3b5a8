#!/bin/bash
# Same-box A/B of two BUILDS (a compile-time change): each library runs the knob sweep in its own process of this one call, against
# a common reference setting, twice.  Build the comparison library first, e.g.
#   git show <commit>:stable-diffusion-webui_amd/csrc/gemm.hip > /tmp/gemm_old.hip
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Istable-diffusion-webui_amd/csrc -c /tmp/gemm_old.hip -o /tmp/gemm_old.o
#   hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_old.o stable-diffusion-webui_amd/csrc/build/{attention,norm,elementwise,engine,capi,prof}.o \
#         -o stable-diffusion-webui_amd/lib/libsdmi_old.so
# (round 2: channel-block-major conv K order with the old per-tile address rebuild vs the per-source pointer + tap-mask form, each
# against the tap-major order: profiles/r02_conv_korder.md)
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in libsdmi_old.so libsdmi.so; do
  SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 600 python tools/gpu/knob_sweep.py base conv_korder=0 --reps 2 --profile --out gpurun_out/knob_sweep_r15_$lib.$rep.json > gpurun_out/knob_sweep_r15_$lib.$rep.log 2>&1
  echo "$lib rep $rep rc=$?"; grep -E "^base|^conv_korder|conv3x3" gpurun_out/knob_sweep_r15_$lib.$rep.log | head -4
done
done
