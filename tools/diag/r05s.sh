#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 3; }
{
echo "== default"; run --iters 8000 --other gemm --N 1024 --S 256 --F 16
echo "== no packed fp32"; PFPP_LIB=tools/lab/_bin/libpfpp_NOPK.so run --iters 8000 --other gemm --N 1024 --S 256 --F 16
echo "== dword point loads"; PFPP_LIB=tools/lab/_bin/libpfpp_VOLATILE_LOAD.so run --iters 8000 --other gemm --N 1024 --S 256 --F 16
} > $O/fps_variants2.txt 2>&1
cat $O/fps_variants2.txt
