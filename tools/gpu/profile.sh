#!/bin/bash
# round-2 measurement pass: bench line (with cpu baseline) + rocprofv3 kernel-trace stats + PMC (HBM traffic, MFMA) passes of the SAME
# workload, then the summaries under profiles/ and a second bench line that carries the measured traffic.
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -2 gpurun_out/bench.log | cut -c1-1800
cd /tmp
# (1) kernel trace + stats of the same job
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_stats.log 2>&1
echo "rocprof stats rc=$?" >> $REPO/gpurun_out/prof_stats.log
# (2) PMC passes over the whole C1 job (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_pmc_fetch -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?" >> $REPO/gpurun_out/prof_pmc_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_pmc_write -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_write.log 2>&1
echo "pmc write rc=$?" >> $REPO/gpurun_out/prof_pmc_write.log
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $REPO/gpurun_out/prof_pmc_mfma -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_mfma.log 2>&1
echo "pmc mfma rc=$?" >> $REPO/gpurun_out/prof_pmc_mfma.log
# (3) K-order A/B of the conv traffic on a 2-step job: tap-major (default) vs channel-block-major (SDMI_CONV_KORDER=1)
for k in 0 1; do
  SDMI_CONV_KORDER=$k timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_pmc_fetch_k$k -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_fetch_k$k.log 2>&1
  SDMI_CONV_KORDER=$k timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_pmc_write_k$k -o pmc -- python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_pmc_write_k$k.log 2>&1
done
cd $REPO
python tools/summarize_profiles.py r02 > gpurun_out/summarize.log 2>&1; tail -2 gpurun_out/summarize.log
# (4) the bench line again, now carrying roofline.traffic measured on this box for this workload
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_with_traffic.log 2>&1
tail -1 gpurun_out/bench_with_traffic.log | cut -c1-1500
mkdir -p gpurun_out/profiles_out && cp profiles/r02_* gpurun_out/profiles_out/ 2>/dev/null   # only gpurun_out/ travels back
# keep the merged-back payload small: raw traces and databases stay on the box, the summaries are already under profiles/
find gpurun_out -name "*.db" -size +8M -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
du -sh gpurun_out
