#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 40; }
{
echo "== default, other=gemm"; run --iters 8000 --other gemm
echo "== default, other=gemm N=1024"; run --iters 8000 --other gemm --N 1024 --S 256 --F 16
} > $O/fps_race_detail.txt 2>&1
cat $O/fps_race_detail.txt
