for shp in "3850 512 512" "3850 1536 512" "3850 512 2048" "3850 4096 512 f16x3 geglu"; do
 for cfg in "X=0" "PFPP_GEMM_RING=1" "PFPP_GEMM_WS=1" "PFPP_GEMM_PF2=0"; do
  echo "== $shp $cfg"; env $cfg python tools/gemm_bench.py $shp 2>/dev/null | tail -1
 done
done
