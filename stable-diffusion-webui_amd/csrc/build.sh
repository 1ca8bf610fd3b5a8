#!/bin/bash
# Build libsdmi.so for gfx950 in-tree (stable-diffusion-webui_amd/lib/libsdmi.so).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-result ${SDMI_EXTRA_FLAGS}"     # e.g. -DSDMI_ATTN_PARTS (tools/gpu/attn_parts.py)
pids=()
for f in gemm.hip attention.hip norm.hip elementwise.hip; do
  hipcc $FLAGS -c "$f" -o "build/${f%.hip}.o" & pids+=($!)
done
# rowchain.hip: no NaN can arise in its softmax / GEGLU (finite operands, -inf only as a key mask), and without the flag every fmaxf of an
# MFMA result costs an extra canonicalising v_max_f32 in an issue-bound loop
hipcc $FLAGS -fno-honor-nans -c rowchain.hip -o build/rowchain.o & pids+=($!)
hipcc $FLAGS -x hip -c engine.cpp -o build/engine.o & pids+=($!)
hipcc $FLAGS -x hip -c capi.cpp -o build/capi.o & pids+=($!)
hipcc $FLAGS -x hip -c prof.cpp -o build/prof.o & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o "$OUT/libsdmi.so"
echo "built $OUT/libsdmi.so"
