#!/bin/bash
# A/B of eval-mode level 2 on the rows kernels (PFPP_SA_EVAL_ROWS2): compact sampler step, auto_aggl with 1 and 32 puzzles in flight
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
cat > /tmp/aggl_once.py <<PY
import sys
sys.path[:0] = ["$GRAFT_REPO_ROOT", "$GRAFT_REPO_ROOT/puzzlefusion-plusplus_amd"]
import torch, bench
dev = torch.device("cuda:0")
a = bench.aggl_puzzles_per_s(dev, n_puzzles=6)
b = bench.aggl_puzzles_per_s(dev, n_puzzles=64, in_flight=32)
print(a["value"], b["value"])
PY
for rep in 1 2; do
for v in 0 1; do
  echo "PFPP_SA_EVAL_ROWS2=$v sampler compact: $(PFPP_SA_EVAL_ROWS2=$v python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")  aggl 1 / 32 in flight: $(PFPP_SA_EVAL_ROWS2=$v python /tmp/aggl_once.py 2>/dev/null | tail -1)"
done
done
