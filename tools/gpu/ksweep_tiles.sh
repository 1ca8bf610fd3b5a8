for cfg in 7 9 10 13 11 0 12 8; do
  ov=""
  for K in 320 640 1280 2560 5120; do ov="${ov}4096,1280,${K},1,0:${cfg}:1;"; done
  echo "=== tile cfg $cfg (split 1) ==="
  SDMI_GEMM_OVERRIDE="$ov" timeout 120 tools/micro/hipblaslt_yardstick 30 ksweep 2>&1 | grep "M4096  N1280 K.*+res" | cut -c1-86
done
