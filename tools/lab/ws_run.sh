#!/bin/bash
# runs every probe variant over the token-GEMM shapes of the training step; output -> gpurun_out/ws_<tag>.txt
cd "$(dirname "$0")/../.."; TAG=${1:-a}; mkdir -p gpurun_out; OUT=gpurun_out/ws_$TAG.txt; : > $OUT
for shp in "3850 512 512" "3850 1536 512" "3850 512 1536" "3850 512 2048" "3850 2048 512" "3850 4096 512" "16000 512 512"; do
  echo "== $shp" >> $OUT
  timeout 60 tools/lab/_run/wdbase $shp >> $OUT 2>&1
  for b in tools/lab/_run/ws_a*; do timeout 60 $b $shp >> $OUT 2>&1 || echo "FAILED/TIMEOUT $b $shp" >> $OUT; done
done
# residual + bias path, and smaller grids (persistent walk with more tiles per workgroup)
echo "== residual / grids" >> $OUT
timeout 60 tools/lab/_run/ws_a0_p0 3850 512 512 256 1 >> $OUT 2>&1
timeout 60 tools/lab/_run/ws_a0_p0 3850 1536 512 256 1 >> $OUT 2>&1
timeout 60 tools/lab/_run/ws_a0_p0 3850 1536 512 128 0 >> $OUT 2>&1
timeout 60 tools/lab/_run/ws_a0_p0 3850 1536 512 244 0 >> $OUT 2>&1
timeout 60 tools/lab/_run/ws_a0_p0 200 512 512 256 1 >> $OUT 2>&1
timeout 60 tools/lab/_run/ws_a0_p0 37 1536 512 256 1 >> $OUT 2>&1
cat $OUT
