#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 8; }
{
echo "== step-tagged exchange entries, other=gemm N=1024"; PFPP_LIB=tools/lab/_bin/libpfpp_CHECK.so run --iters 8000 --other gemm --N 1024 --S 256 --F 16
echo "== step-tagged exchange entries, other=gemm N=512"; PFPP_LIB=tools/lab/_bin/libpfpp_CHECK.so run --iters 8000 --other gemm
echo "== step-tagged exchange entries, other=none N=1024"; PFPP_LIB=tools/lab/_bin/libpfpp_CHECK.so run --iters 3000 --other none --N 1024 --S 256 --F 16
for o in gemm planes wd fps none; do timeout 600 python tools/diag/lds_canary.py --other $o 2>&1 | grep -v amdgpu.ids | tail -n 2; done
} > $O/fps_check.txt 2>&1
cat $O/fps_check.txt
