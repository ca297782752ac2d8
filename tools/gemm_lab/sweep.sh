#!/bin/bash
# every (variant, split) choice of pfpp_gemm_planes on the training step's shapes -> gpurun_out/lab_sweep.txt (tuning data of pl_choose)
cd $GRAFT_REPO_ROOT/tools/gemm_lab
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab_sweep.txt
: > $OUT
for S in nt,3850,512,512 nt,3850,1536,512 nt,3850,4096,512 nt,3850,512,2048 nn,3850,512,512 nn,3850,512,1536 nn,3850,512,4096 nn,3850,2048,512 tn,512,512,3850 tn,1536,512,3850 tn,4096,512,3850 tn,512,2048,3850; do
  echo "== $S auto" >> $OUT
  LAB_WS=1 ./lab2 30 $S 2>&1 | cut -c1-75 >> $OUT
  for V in 2 3 6; do
    ARGS=""
    for SP in 1 2 3 4 6 8 12 16; do ARGS="$ARGS $S,$V,$SP"; done
    LAB_WS=1 ./lab2 30 $ARGS 2>&1 | cut -c1-75 >> $OUT
  done
done
