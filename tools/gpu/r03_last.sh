#!/bin/bash
# last check of the final tree: the driver's two GPU commands (whole gpu suite, smoke, default bench line)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu_last.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_last.log; tail -3 gpurun_out/pytest_gpu_last.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_last.log 2>&1; tail -1 gpurun_out/smoke_last.log
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/bench_last.log 2> gpurun_out/bench_last.err
echo "bench rc=$?"; tail -1 gpurun_out/bench_last.log | cut -c1-300; grep "Elapsed (wall" gpurun_out/bench_last.err
python - <<'PY'
import json
line = [l for l in open('gpurun_out/bench_last.log') if l.startswith('{')][-1]
d = json.loads(line)
print("cpu_baseline:", json.dumps(d["cpu_baseline"])[:900])
PY
