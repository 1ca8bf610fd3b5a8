#!/bin/bash
# bench + rocprofv3 kernel-trace stats + PMC (HBM traffic) passes
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 4 -p no:cacheprovider --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -2 gpurun_out/bench.log | cut -c1-1500
cd /tmp
# (1) kernel trace + stats of the same bench command (no cpu baseline / roofline pass to keep the trace to the timed path)
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_stats.log 2>&1
echo "rocprof stats rc=$?" >> $REPO/gpurun_out/prof_stats.log
# (2) PMC passes on a shortened job (2 sampler steps: per-launch traffic does not depend on the step count)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_pmc_fetch -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?" >> $REPO/gpurun_out/prof_pmc_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_pmc_write -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_write.log 2>&1
echo "pmc write rc=$?" >> $REPO/gpurun_out/prof_pmc_write.log
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $REPO/gpurun_out/prof_pmc_mfma -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_mfma.log 2>&1
echo "pmc mfma rc=$?" >> $REPO/gpurun_out/prof_pmc_mfma.log
cd $REPO
find gpurun_out -name "*.csv" | head -30
du -sh gpurun_out
# keep the merged-back payload small: drop raw kernel traces above 20 MB, keep stats
find gpurun_out -name "*kernel_trace.csv" -size +20M -exec sh -c 'head -2000 "$1" > "$1.head"; rm "$1"' _ {} \;
tail -3 gpurun_out/prof_stats.log; tail -3 gpurun_out/prof_pmc_fetch.log
