#!/bin/bash
# Same-box A/B of two BUILDS of the attention kernels (lib/libsdmi.so against lib/libsdmi_alt.so — e.g. attention.hip compiled with
# -mllvm -amdgpu-sched-strategy=max-ilp): the standalone self-attention timings and the whole C1 job, each library in its own process,
# interleaved twice; the alternative library's attention op tests first (same bits expected: only the instruction order differs).
export TMPDIR=/tmp
mkdir -p gpurun_out
ALT=$PWD/stable-diffusion-webui_amd/lib/libsdmi_alt.so
SDMI_LIB=$ALT timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "attn or attention" --tb=short > gpurun_out/ab_attn_pytest.log 2>&1
echo "alt attention tests rc=$?"; tail -2 gpurun_out/ab_attn_pytest.log
for rep in 1 2; do
for lib in libsdmi.so libsdmi_alt.so; do
  [ $rep = 1 ] && SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 200 python tools/gpu/attn_time.py --variants 15 --reps 4 --out gpurun_out/ab_attn_time_$lib.$rep.json > gpurun_out/ab_attn_time_$lib.$rep.log 2>&1
  [ $rep = 1 ] && grep -o "^[a-z0-9 ]*level [0-9]\|'us_min': [0-9.]*" gpurun_out/ab_attn_time_$lib.$rep.log | paste -sd' ' | cut -c1-400
  SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 300 python tools/gpu/knob_sweep.py base --reps 3 --out gpurun_out/ab_attn_job_$lib.$rep.json > gpurun_out/ab_attn_job_$lib.$rep.log 2>&1
  echo "$lib rep $rep job rc=$?"; grep -E "^base" gpurun_out/ab_attn_job_$lib.$rep.log | head -2
done
done
