#!/bin/bash
# is the training iteration host-bound?  a busy-wait on the host before every backward (PFPP_DIAG_HOST_DELAY_US): an iteration that
# does not get longer has at least that much host slack
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
for d in 0 200 400 800 0; do
  echo "host delay $d us: $(PFPP_DIAG_HOST_DELAY_US=$d python bench.py --steps 40 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done
