#!/bin/bash
# A/B of the eval-mode level 3 on the rows kernels (PFPP_SA_EVAL_ROWS): sampler step, compact and all slots
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for v in 0 1; do
  for mode in "--compact" ""; do
  echo "PFPP_SA_EVAL_ROWS=$v sampler $mode: $(PFPP_SA_EVAL_ROWS=$v python bench.py --mode sample $mode --steps 30 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
done
