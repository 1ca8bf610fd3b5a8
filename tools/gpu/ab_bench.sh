#!/bin/bash
# same-box A/B of an environment knob on the whole-job bench:  bash tools/gpu/ab_bench.sh VAR A B
export TMPDIR=/tmp
mkdir -p gpurun_out
VAR=$1; A=$2; B=$3
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done | tee gpurun_out/ab_bench.log
