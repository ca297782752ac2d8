#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 4; }
{
echo "== default, other=gemm"; run --iters 5000 --other gemm
echo "== full barrier, other=gemm"; PFPP_LIB=tools/lab/_bin/libpfpp_FULL_BARRIER.so run --iters 5000 --other gemm
echo "== selections through LDS, raw barrier, other=gemm"; PFPP_LIB=tools/lab/_bin/libpfpp_LDS_OUT.so run --iters 5000 --other gemm
echo "== default, N=1024 (4 waves), other=gemm"; run --iters 5000 --other gemm --N 1024 --S 256 --F 16
echo "== default, N=2048 (4 waves x 8), other=gemm"; run --iters 3000 --other gemm --N 2048 --S 256 --F 16
} > $O/fps_race_variants.txt 2>&1
cat $O/fps_race_variants.txt
