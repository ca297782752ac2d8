#!/bin/bash
# where the next iteration's encoder is issued: before this iteration's forward (default) or between its forward and backward
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-roofline"
run() { echo "$1 $2: $(env $1 python bench.py --steps 40 --warmup 8 $B $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"; }
for rep in 1 2; do
run "PFPP_BENCH_ENC_AFTER_FWD=0" ""
run "PFPP_BENCH_ENC_AFTER_FWD=1" ""
run "PFPP_BENCH_ENC_AFTER_FWD=1 PFPP_ENC_CU_FRACTION_PCT=60" ""
run "PFPP_BENCH_ENC_AFTER_FWD=1 PFPP_ENC_CU_FRACTION_PCT=40" ""
run "PFPP_BENCH_ENC_AFTER_FWD=1 PFPP_ENC_CU_FRACTION_PCT=0" ""
run "PFPP_BENCH_ENC_AFTER_FWD=1 PFPP_TRAIN_DW_GROUP_VARIANT=6" ""
done
