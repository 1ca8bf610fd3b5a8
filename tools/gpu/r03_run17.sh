#!/bin/bash
# round 3, GPU call 17: folded-shift attention (attn_occ 17) on the hires-fix workload: parity at the 128x128-latent shapes, then c4a with / without
export TMPDIR=/tmp
mkdir -p gpurun_out
SDMI_ATTN_OCC=17 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -p no:cacheprovider --tb=short --timeout 900 -k "c4 or attention" > gpurun_out/pytest_gpu17.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu17.log; tail -4 gpurun_out/pytest_gpu17.log | cut -c1-300
cp gpurun_out/r03_parity_fullsize.json gpurun_out/r03_parity_fullsize_occ17.json 2>/dev/null
for rep in 1 2; do
  for occ in 15 17; do
    SDMI_ATTN_OCC=$occ timeout 600 python bench.py --config c4a --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('occ $occ rep $rep', d['value'], d['ms_per_step'])"
  done
done
