#!/bin/bash
# round 2, final pass: the whole GPU suite, the measurement pipeline of tools/gpu/profile.sh (bench line + rocprofv3 stats + PMC passes
# + summaries under profiles/), then the other BASELINE configs through bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; grep -E "^\[c1|^\[c3|passed|failed|rc=|Error|assert" gpurun_out/pytest_gpu.log | tail -14
bash tools/gpu/profile.sh
for c in c2 c4b c4a c3; do
  timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$c.log 2>&1
  echo "bench $c rc=$?" >> gpurun_out/bench_$c.log; tail -2 gpurun_out/bench_$c.log | cut -c1-400
done
du -sh gpurun_out
