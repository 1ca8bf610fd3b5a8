#!/bin/bash
# rocprofv3 kernel-trace stats of the C1 job on the final tree -> profiles/r04_kernel_stats.md (tools/summarize_profiles.py)
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_stats.log 2>&1
echo "rocprof stats rc=$?" >> $REPO/gpurun_out/prof_stats.log; tail -2 $REPO/gpurun_out/prof_stats.log | cut -c1-300
cd $REPO
python tools/summarize_profiles.py r04 > gpurun_out/summarize.log 2>&1; tail -3 gpurun_out/summarize.log
mkdir -p gpurun_out/profiles_out && cp profiles/r04_kernel_stats.md gpurun_out/profiles_out/ 2>/dev/null
find gpurun_out -name "*.db" -size +8M -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
