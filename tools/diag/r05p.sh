#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
run() { timeout 900 python tools/diag/fps_race.py "$@" 2>&1 | grep -v amdgpu.ids | tail -n 2; }
{
for other in ew ln gemm planes wd fps none; do echo "== other=$other"; run --iters 5000 --other $other; done
} > $O/fps_race_others.txt 2>&1
cat $O/fps_race_others.txt
