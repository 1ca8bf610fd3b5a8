#!/bin/bash
# round 3, GPU call 15: hardware counters of the level-0 attention kernel alone (separate --pmc passes, kernel trace only)
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out/attn_pmc
cd /tmp
rocprofv3 -L > $REPO/gpurun_out/attn_pmc/avail.txt 2>&1 || rocprofv3 --list-avail > $REPO/gpurun_out/attn_pmc/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $REPO/gpurun_out/attn_pmc/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $REPO/gpurun_out/attn_pmc/p$i -o pmc -- python $REPO/tools/gpu/attn_only.py > $REPO/gpurun_out/attn_pmc/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
cd $REPO
python - <<'PY'
import glob, sqlite3, json, collections
out = {}
for d in sorted(glob.glob('gpurun_out/attn_pmc/p*/')):
    dbs = glob.glob(d + '**/*.db', recursive=True)
    if not dbs:
        continue
    con = sqlite3.connect(dbs[0])
    try:
        rows = list(con.execute("select kernel_name, counter_name, value, duration from counters_collection"))
    except Exception as e:
        out[d] = str(e); continue
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for kn, cn, v, dur in rows:
        if 'attn_mfma_kernel' in kn:
            a = acc[cn]; a[0] += 1; a[1] += v; a[2] += dur
    for cn, (n, v, dur) in acc.items():
        out[cn] = {"launches": n, "mean_value": v / max(n, 1), "mean_duration_us": dur / max(n, 1) / 1e3}
json.dump(out, open('gpurun_out/attn_pmc/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY
find gpurun_out/attn_pmc -name "*.db" -size +4M -delete
