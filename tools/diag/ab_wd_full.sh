#!/bin/bash
# A/B of the weight-direct GEMMs in the all-slots eval forward (PFPP_EVAL_WD_FULL): sampler step with padded slots evaluated
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for v in 0 1; do
  echo "PFPP_EVAL_WD_FULL=$v sampler (all slots): $(PFPP_EVAL_WD_FULL=$v python bench.py --mode sample --steps 30 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
done
