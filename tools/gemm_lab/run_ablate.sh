#!/bin/bash
# on the GPU box: ablation sweep + counter passes of the plane GEMM (developer tool)
cd $GRAFT_REPO_ROOT/tools/gemm_lab
OUT=$GRAFT_REPO_ROOT/gpurun_out
SHAPES="${SHAPES:-3850,1536,512 16000,512,2048 4096,4096,4096 16000,4096,512}"
for D in 0 1 2 3 7; do
  echo "=== PFPP_GEMM_DBG=$D"
  PFPP_GEMM_DBG=$D LAB_VARIANTS=${LAB_VARIANTS:-123} timeout 200 ./lab 10 $SHAPES
done 2>&1 | tee $OUT/lab_ablate.txt
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  LAB_VARIANTS=${LAB_VARIANTS:-123} timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_$i -- $GRAFT_REPO_ROOT/tools/gemm_lab/lab 3 ${PMC_SHAPE:-16000,512,2048} > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pmc_$i -name "*_results.db" | head -1) $OUT/lab_pmc_$i.csv --pmc
done
