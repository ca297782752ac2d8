#!/bin/bash
# where the overlapped iteration's time goes (lab switches: WRONG gradients with PFPP_LAB_SKIP_DW_LAYERS) and a sweep of the CU masks of the
# encoder / weight-gradient streams with the grouped weight-gradient launch (160 one-per-CU workgroups per block)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
run() { echo "$1 $2: $(env $1 python bench.py --steps 40 --warmup 8 $B $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"; }
run "PFPP_X=0" ""
run "PFPP_X=0" "--latents-given"
run "PFPP_LAB_SKIP_DW_LAYERS=6" ""
run "PFPP_LAB_SKIP_DW_LAYERS=6" "--latents-given"
run "PFPP_X=0" "--serial"
for side in 50 62 75; do for enc in 40 50; do
  run "PFPP_SIDE_CU_FRACTION_PCT=$side PFPP_ENC_CU_FRACTION_PCT=$enc" ""
done; done
run "PFPP_SIDE_CU_FRACTION_PCT=35 PFPP_ENC_CU_FRACTION_PCT=40" ""
run "PFPP_SIDE_CU_FRACTION_PCT=40 PFPP_ENC_CU_FRACTION_PCT=50" ""
run "PFPP_ENC_CU_FRACTION_PCT=40" ""
run "PFPP_ENC_CU_FRACTION_PCT=60" ""
run "PFPP_X=0" ""
