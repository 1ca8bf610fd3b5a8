#!/bin/bash
# round 3, GPU call 8: new tests (DDIM CFG++ on a v-prediction model, two bench ranks on one GPU, img2img shape check), then the
# in-engine shape tuner with the 256x128 ping-pong tile and split-K down to K = 1024 among the candidates
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_boundaries.py -m gpu -q -p no:cacheprovider --tb=short --timeout 600 -k "v_prediction or bench_two or pil_front_end or img2img" > gpurun_out/pytest_gpu8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu8.log; tail -8 gpurun_out/pytest_gpu8.log
timeout 900 python tools/gpu/shape_tune.py --top 45 --emit > gpurun_out/shape_tune8.log 2>&1
echo "tune rc=$?" >> gpurun_out/shape_tune8.log; tail -5 gpurun_out/shape_tune8.log | cut -c1-300
