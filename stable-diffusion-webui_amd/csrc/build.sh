#!/bin/bash
# Build libsdmi.so for gfx950 in-tree (stable-diffusion-webui_amd/lib/libsdmi.so).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" build/asm
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-result ${SDMI_EXTRA_FLAGS}"     # e.g. -DSDMI_ATTN_PARTS (tools/gpu/attn_parts.py)
# -save-temps=obj: the device assembly of every kernel file falls out of the same compile; kept as build/asm/<name>.s (listed in .gpurunignore: 75 MB the GPU box has no use for) for the ISA tests
# (tests/test_cpu_host.py::_gfx950_assembly: K-loop instruction counts, scratch / spill / register budgets), the other temporaries deleted
pids=()
for f in gemm.hip attention.hip norm.hip elementwise.hip; do
  hipcc $FLAGS -save-temps=obj -c "$f" -o "build/${f%.hip}.o" & pids+=($!)
done
# rowchain.hip: no NaN can arise in its softmax / GEGLU (finite operands, -inf only as a key mask), and without the flag every fmaxf of an
# MFMA result costs an extra canonicalising v_max_f32 in an issue-bound loop
hipcc $FLAGS -fno-honor-nans -save-temps=obj -c rowchain.hip -o build/rowchain.o & pids+=($!)
hipcc $FLAGS -x hip -c engine.cpp -o build/engine.o & pids+=($!)
hipcc $FLAGS -x hip -c capi.cpp -o build/capi.o & pids+=($!)
hipcc $FLAGS -x hip -c prof.cpp -o build/prof.o & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
for f in gemm attention norm elementwise rowchain; do
  mv -f "build/$f-hip-amdgcn-amd-amdhsa-gfx950.s" "build/asm/$f.s"
  rm -f build/$f-hip-amdgcn-amd-amdhsa-gfx950.* build/$f-host-x86_64-unknown-linux-gnu.* build/$f.hip-hip-amdgcn-amd-amdhsa.hipfb
done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o "$OUT/libsdmi.so"
echo "built $OUT/libsdmi.so"
