#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
for other in fps gemm none; do timeout 600 python tools/diag/fps_race.py --iters 1500 --other $other 2>&1 | grep -v amdgpu.ids; done > $O/fps_race_default.txt
PFPP_LIB=tools/lab/_bin/libpfpp_fullbar.so timeout 600 python tools/diag/fps_race.py --iters 1500 --other fps 2>&1 | grep -v amdgpu.ids > $O/fps_race_fullbar.txt
timeout 600 python tools/diag/fps_race.py --iters 1500 --other fps --N 1024 --S 256 --F 16 2>&1 | grep -v amdgpu.ids > $O/fps_race_n1024.txt
timeout 600 python tools/diag/fps_race.py --iters 1500 --other fps --N 256 --S 128 --F 16 2>&1 | grep -v amdgpu.ids > $O/fps_race_n256.txt
tail -n 4 $O/fps_race_*.txt
: > $O/wdirect_kt.txt
for shp in "3850 512 512" "3850 1536 512" "3850 512 1536" "3850 512 2048" "3850 2048 512" "16000 512 512"; do
  echo "== $shp" >> $O/wdirect_kt.txt
  for b in tools/lab/_bin/wd_*; do timeout 60 $b $shp >> $O/wdirect_kt.txt 2>&1; done
done
grep -v "^GPU core\|^Memory access" $O/wdirect_kt.txt | sed 's/(fp32-grade), //' | cut -c1-140
