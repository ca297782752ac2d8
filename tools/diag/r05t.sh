#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 1; }
{
for rep in 1 2 3; do
echo "== round $rep"
run --iters 8000 --other gemm --N 1024 --S 256 --F 16
PFPP_LIB=tools/lab/_bin/libpfpp_PKNOP.so run --iters 8000 --other gemm --N 1024 --S 256 --F 16
PFPP_LIB=tools/lab/_bin/libpfpp_NOPKFEAT.so run --iters 8000 --other gemm --N 1024 --S 256 --F 16
run --iters 8000 --other gemm
PFPP_LIB=tools/lab/_bin/libpfpp_PKNOP.so run --iters 8000 --other gemm
PFPP_LIB=tools/lab/_bin/libpfpp_NOPKFEAT.so run --iters 8000 --other gemm
done
} > $O/fps_pk_hazard.txt 2>&1
cat $O/fps_pk_hazard.txt
