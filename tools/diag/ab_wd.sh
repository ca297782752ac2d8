#!/bin/bash
# A/B of the weight-direct eval GEMM (PFPP_EVAL_WD): compact sampler step, full auto_aggl loop
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for wd in 0 1; do
  echo "PFPP_EVAL_WD=$wd sampler compact: $(PFPP_EVAL_WD=$wd python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
done
