#!/bin/bash
# round 2: new feature tests + the other BASELINE configs through bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_boundaries.py -m gpu -q -p no:cacheprovider --tb=short \
  -k "hypernet or lycoris or nan_check or tiling or sde or adaptive or boundar or checkpoint_file or vae_decode_hook" > gpurun_out/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -6 gpurun_out/pytest_new.log
for c in c2 c4b c4a c3; do
  timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$c.log 2>&1
  echo "bench $c rc=$?" >> gpurun_out/bench_$c.log; tail -2 gpurun_out/bench_$c.log | cut -c1-1200
done
