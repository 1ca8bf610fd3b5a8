export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" | sort -u > gpurun_out/pmc_avail.txt; wc -l gpurun_out/pmc_avail.txt
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_ring$i -o p -- $R/tools/micro/ring_check 3 > $R/gpurun_out/pmc_ring$i.log 2>&1; echo "pass $i ($C) rc=$?")
done
python - <<'PY'
import csv,glob,collections
out=open("gpurun_out/pmc_ring_summary.txt","w")
for d in sorted(glob.glob("gpurun_out/pmc_ring[0-9]")):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: out.write(d+": no counter file\n"); continue
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k=(r["Kernel_Name"][:110],r["Grid_Size"],r["Counter_Name"])
        a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,a in agg.items():
        out.write("%-110s grid %8s %-34s n=%3d avg %.4g\n"%(k[0],k[1],k[2],a[0],a[1]/a[0]))
out.close()
PY
rm -rf gpurun_out/pmc_ring[0-9]; wc -l gpurun_out/pmc_ring_summary.txt
