#!/bin/bash
# the iteration's exposed tail: AdaLN-linear gradients / AdamW per block on the side stream (PFPP_TRAIN_ADA_IN_C) and the timestep tables'
# untouched rows updated at the start of the backward (PFPP_TRAIN_TABLES_EARLY)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
run() { echo "$1 $2: $(env $1 python bench.py --steps 40 --warmup 8 $B $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['extra'].get('final_loss'))")"; }
for rep in 1 2 3; do
run "PFPP_TRAIN_TABLES_EARLY=0 PFPP_TRAIN_DW_GROUP=1" ""
run "PFPP_TRAIN_TABLES_EARLY=1 PFPP_TRAIN_DW_GROUP=1" ""
run "PFPP_TRAIN_TABLES_EARLY=0 PFPP_TRAIN_DW_GROUP=3" ""
run "PFPP_TRAIN_TABLES_EARLY=1 PFPP_TRAIN_DW_GROUP=3" ""
run "PFPP_TRAIN_TABLES_EARLY=1 PFPP_TRAIN_DW_GROUP=3 PFPP_TRAIN_ADA_IN_C=1" ""
done
