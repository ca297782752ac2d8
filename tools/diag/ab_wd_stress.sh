#!/bin/bash
# A/B of the weight-direct GEMMs in the stress step (single-pass fp16 mode; PFPP_EVAL_WD)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for v in 0 1; do
  echo "PFPP_EVAL_WD=$v stress: $(PFPP_EVAL_WD=$v python bench.py --mode stress --steps 8 --warmup 2 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
done
