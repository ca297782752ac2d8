#!/bin/bash
# A/B of the weight-direct GEMMs in the training step (PFPP_TRAIN_WD)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for wd in 0 1; do
  echo "PFPP_TRAIN_WD=$wd train: $(PFPP_TRAIN_WD=$wd python bench.py --steps 30 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
done
