set -x
for shp in "16000 4096 512" "16000 1536 512" "16000 2048 512 f16x3 geglu" "16000 512 2048" "32000 4096 512" "16000 1024 512"; do
 for cfg in "PFPP_GEMM_BIG4=1" "PFPP_GEMM_BIG4=0" "PFPP_GEMM_BIG=0"; do
  echo "== $cfg"; env $cfg python tools/gemm_bench.py $shp
 done
done
