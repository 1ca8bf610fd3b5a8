#!/bin/bash
# same-box A/B of whole rounds: the round-2 final library (commit 569c355 built as libsdmi_r02.so; the C ABI did not change in round 3, so
# the same Python host drives both) against this tree's, three interleaved repetitions, one process each
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/ab_rounds.jsonl
for rep in 1 2 3; do
  for lib in libsdmi_r02.so libsdmi.so; do
    SDMI_LIB=$PWD/stable-diffusion-webui_amd/lib/$lib timeout 300 python tools/gpu/job_time.py 5 2>/dev/null | grep '^{' >> gpurun_out/ab_rounds.jsonl
  done
done
cat gpurun_out/ab_rounds.jsonl
# the new GPU test of this session's last edits
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "sd15_widths or pil_front_end" > gpurun_out/pytest_gpu11.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu11.log; tail -4 gpurun_out/pytest_gpu11.log
