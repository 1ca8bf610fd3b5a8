#!/bin/bash
# the driver's bench command on the final tree, wall-clocked
export TMPDIR=/tmp
mkdir -p gpurun_out
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_last.log 2> gpurun_out/bench_last.err
echo "bench rc=$? wall ${SECONDS}s"; tail -1 gpurun_out/bench_last.log | cut -c1-300
python - <<'PY'
import json
line = [l for l in open('gpurun_out/bench_last.log') if l.startswith('{')][-1]
d = json.loads(line)
print("cpu_baseline:", json.dumps(d["cpu_baseline"])[:1200])
PY
