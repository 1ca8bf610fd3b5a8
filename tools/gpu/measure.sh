#!/bin/bash
# The measurement passes of a round, one script for every round:  bash tools/gpu/measure.sh <tag> <what> [pytest -k expression]
#   tag    prefix of the summaries written under profiles/ (r05, ...)
#   what   suite    the whole GPU suite with SDMI_PARITY_FULL=1 (or the tests selected by the third argument)
#          close    suite + __graft_entry__.smoke() + the driver's bench command (python bench.py --gpus 1 --steps 20 --warmup 5)
#          c1       smoke, the default bench line (CPU baseline, same-run PMC traffic), rocprofv3 --kernel-trace --stats and the
#                   MFMA-utilisation PMC pass of the C1 job, summaries through tools/summarize_profiles.py
#          stats    the rocprofv3 kernel-trace stats of the C1 job only
#          configs  bench line (+ PMC traffic) and rocprofv3 stats of c2 / c3 / c4a / c4b
#          micro    the C-ABI-only checks (conv_check, ring_check, hipblaslt_yardstick) and the torch-free forward profile (fwd_ab.py)
# (rounds 2-4 ran these steps from per-round copies of this script; their outputs are the profiles/r0N_* files.)
export TMPDIR=/tmp
REPO=$(pwd)
TAG=${1:?tag}; WHAT=${2:?what}; SEL=$3
mkdir -p gpurun_out
trim() { find gpurun_out -name "*.db" -size +8M -delete; find gpurun_out -name "*kernel_trace.csv" -size +8M -delete; }
suite() {
  if [ -n "$SEL" ]; then
    timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 900 -k "$SEL" > gpurun_out/pytest_gpu.log 2>&1
  else
    SDMI_PARITY_FULL=1 timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
  fi
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
}
smoke() { python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log; }
stats() {   # $1 = config (none: the C1 job, one warm-up job)
  local W=1; [ -n "$1" ] && W=0
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats${1:+_$1} -o bench -- python $REPO/bench.py ${1:+--config $1} --steps 1 \
      --warmup $W --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_stats${1:+_$1}.log 2>&1)
  echo "rocprof stats rc=$?" >> gpurun_out/prof_stats${1:+_$1}.log
}
case "$WHAT" in
  suite) suite ;;
  close)
    suite; smoke
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_close.log 2>&1
    echo "bench rc=$?" >> gpurun_out/bench_close.log; tail -2 gpurun_out/bench_close.log | cut -c1-1200 ;;
  stats)
    stats; python tools/summarize_profiles.py $TAG > gpurun_out/summarize.log 2>&1; tail -3 gpurun_out/summarize.log; trim ;;
  c1)
    smoke
    timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
    echo "bench rc=$?" >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-2500
    stats
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $REPO/gpurun_out/prof_pmc_mfma -o pmc -- \
        python $REPO/bench.py --steps 1 --warmup 0 --sampler-steps 2 --no-cpu-baseline --no-roofline --no-dropin > $REPO/gpurun_out/prof_pmc_mfma.log 2>&1)
    echo "pmc mfma rc=$?" >> gpurun_out/prof_pmc_mfma.log
    python tools/summarize_profiles.py $TAG > gpurun_out/summarize.log 2>&1; tail -2 gpurun_out/summarize.log; trim ;;
  configs)
    for c in c2 c3 c4a c4b; do
      timeout 1500 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic > gpurun_out/bench_$c.log 2>&1
      echo "bench rc=$?" >> gpurun_out/bench_$c.log; tail -2 gpurun_out/bench_$c.log | cut -c1-400
      cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_$c.json 2>/dev/null
      stats $c
    done
    python tools/summarize_profiles.py $TAG --configs c2 c3 c4a c4b >> gpurun_out/summarize.log 2>&1; tail -2 gpurun_out/summarize.log; trim ;;
  micro)
    for t in conv_check ring_check hipblaslt_yardstick; do
      [ -x tools/micro/$t ] && { timeout 150 tools/micro/$t 20 > gpurun_out/${TAG}_$t.txt 2>&1; echo "$t rc=$?"; tail -3 gpurun_out/${TAG}_$t.txt; }
    done
    timeout 120 python tools/gpu/fwd_ab.py base --reps 4 --fwd 10 --profile --out gpurun_out/${TAG}_fwd_ab_base.json > gpurun_out/${TAG}_fwd_ab_base.log 2>&1
    timeout 120 python tools/gpu/fwd_ab.py base --what vae --rows 8 --reps 4 --fwd 5 --profile --out gpurun_out/${TAG}_fwd_ab_vae.json > gpurun_out/${TAG}_fwd_ab_vae.log 2>&1; grep -v "^    " gpurun_out/${TAG}_fwd_ab_vae.log | tail -3
    grep -v "^    " gpurun_out/${TAG}_fwd_ab_base.log | tail -4 ;;
  *) echo "unknown pass $WHAT"; exit 2 ;;
esac
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/ 2>/dev/null
du -sh gpurun_out
