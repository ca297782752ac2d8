#!/bin/bash
# A/B of the grouped weight-gradient launch in the training step (PFPP_TRAIN_DW_GROUP: 0 = one plane GEMM + slab reduction per weight,
# 1 = one launch per block, 2 = two launches per block) and of its tile variant (PFPP_TRAIN_DW_GROUP_VARIANT)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
run() { echo "$1: $(env $1 python bench.py --steps 40 --warmup 8 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['extra'].get('final_loss'))")"; }
for rep in 1 2; do
  run "PFPP_TRAIN_DW_GROUP=0"
  for v in 3 6 7 2; do run "PFPP_TRAIN_DW_GROUP=1 PFPP_TRAIN_DW_GROUP_VARIANT=$v"; done
  run "PFPP_TRAIN_DW_GROUP=2 PFPP_TRAIN_DW_GROUP_VARIANT=6"
  run "PFPP_TRAIN_DW_GROUP=2 PFPP_TRAIN_DW_GROUP_VARIANT=3"
done
