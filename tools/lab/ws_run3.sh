#!/bin/bash
cd "$(dirname "$0")/../.."; TAG=${1:-a}; mkdir -p gpurun_out; OUT=gpurun_out/ws_$TAG.txt; : > $OUT
for shp in "3850 512 2048" "3850 512 512"; do
  echo "== $shp" >> $OUT
  for b in tools/lab/_run/ws_a*; do timeout 60 $b $shp >> $OUT 2>&1 || echo "FAILED/TIMEOUT $b $shp" >> $OUT; done
done
cut -c1-125 $OUT
