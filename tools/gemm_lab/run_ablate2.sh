#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/gemm_lab
OUT=$GRAFT_REPO_ROOT/gpurun_out
SHAPES="${SHAPES:-4096,4096,4096 16000,512,2048 3850,1536,512}"
for D in ${DBGS:-0 32 7 39 23 15}; do
  echo "=== PFPP_GEMM_DBG=$D"
  PFPP_GEMM_DBG=$D LAB_VARIANTS=${LAB_VARIANTS:-123} timeout 200 ./lab 10 $SHAPES | grep -v default
done 2>&1 | tee $OUT/lab_ablate2.txt
