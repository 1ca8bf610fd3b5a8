#!/bin/bash
# round 4, final measurement pass on the final tree (argument: c1 = the C1 passes only, configs = c2..c4b only, suite = the GPU suite with
# SDMI_PARITY_FULL=1 only, none = C1 + configs): bench line with CPU baseline and same-run PMC traffic,
# rocprofv3 kernel-trace stats + MFMA-utilisation pass of the C1 job, then bench + rocprofv3 stats of c2 / c3 / c4a / c4b.
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
if [ "$1" = "suite" ]; then
  SDMI_PARITY_FULL=1 timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu_final.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log; tail -4 gpurun_out/pytest_gpu_final.log
  exit 0
fi
if [ "$1" != "configs" ]; then
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
# the driver's command, default flags: the line with the CPU baseline
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-2500
# the same with the two PMC passes taken by bench.py itself (roofline.traffic measured in this run)
timeout 1500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic > gpurun_out/bench_with_traffic.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_with_traffic.log; tail -2 gpurun_out/bench_with_traffic.log | cut -c1-1800
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_stats.log 2>&1
echo "rocprof stats rc=$?" >> $REPO/gpurun_out/prof_stats.log
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $REPO/gpurun_out/prof_pmc_mfma -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_pmc_mfma.log 2>&1
echo "pmc mfma rc=$?" >> $REPO/gpurun_out/prof_pmc_mfma.log
cd $REPO
python tools/summarize_profiles.py r04 > gpurun_out/summarize.log 2>&1; tail -2 gpurun_out/summarize.log
fi
if [ "$1" = "c1" ]; then
  mkdir -p gpurun_out/profiles_out && cp profiles/r04_kernel_stats*.md profiles/r04_pmc_*.md profiles/r04_pmc_*.json gpurun_out/profiles_out/ 2>/dev/null
  find gpurun_out -name "*.db" -size +8M -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete; exit 0
fi
# the other BASELINE.json workloads: bench line + rocprofv3 kernel stats each
for c in c2 c3 c4a c4b; do
  timeout 1500 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic > gpurun_out/bench_$c.log 2>&1
  echo "bench rc=$?" >> gpurun_out/bench_$c.log; tail -2 gpurun_out/bench_$c.log | cut -c1-400
  cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_$c.json 2>/dev/null
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats_$c -o bench -- python $REPO/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_stats_$c.log 2>&1)
done
python tools/summarize_profiles.py r04 --configs c2 c3 c4a c4b >> gpurun_out/summarize.log 2>&1; tail -2 gpurun_out/summarize.log
mkdir -p gpurun_out/profiles_out && cp profiles/r04_kernel_stats*.md profiles/r04_pmc_*.md profiles/r04_pmc_*.json profiles/r04_parity*.json gpurun_out/profiles_out/ 2>/dev/null
find gpurun_out -name "*.db" -size +8M -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
du -sh gpurun_out
