#!/bin/bash
# runs every W-direct probe variant over the token-GEMM shapes of the step; output -> gpurun_out/wdirect_<tag>.txt
cd "$(dirname "$0")/../.."; TAG=${1:-a}; OUT=gpurun_out/wdirect_$TAG.txt; : > $OUT
for shp in "3850 512 512" "3850 1536 512" "3850 4096 512" "3850 512 2048" "16000 512 512" "16000 1536 512" "16000 4096 512" "16000 512 2048"; do
  echo "== $shp" >> $OUT
  tools/lab/_bin/gemm_adirect_probe $shp >> $OUT 2>&1
  for b in tools/lab/_bin/wd_*; do timeout 60 $b $shp >> $OUT 2>&1; done
done
