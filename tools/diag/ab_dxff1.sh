#!/bin/bash
# tile of the GEGLU projection's input gradient (3850 x 512 x 4096, split K) next to the 144 KB-LDS workgroups of the grouped weight gradients
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
run() { echo "$1 $2: $(env $1 python bench.py --steps 40 --warmup 8 $B $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['extra'].get('final_loss'))")"; }
for rep in 1 2; do
run "PFPP_X=0" ""
run "PFPP_TRAIN_DXFF1_VARIANT=3" ""
run "PFPP_TRAIN_DXFF1_VARIANT=6" ""
run "PFPP_TRAIN_DXFF1_VARIANT=3 PFPP_TRAIN_DXFF1_SPLITS=2" ""
run "PFPP_TRAIN_DXFF1_VARIANT=3 PFPP_TRAIN_DXFF1_SPLITS=4" ""
run "PFPP_TRAIN_DXFF1_VARIANT=6 PFPP_TRAIN_DXFF1_SPLITS=2" ""
run "PFPP_TRAIN_DXFF1_VARIANT=2 PFPP_TRAIN_DXFF1_SPLITS=2" ""
done
