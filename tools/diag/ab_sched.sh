# scheduling knobs of the training iteration, same box back to back (ms per iteration)
cd $GRAFT_REPO_ROOT
B="--steps 30 --no-cpu-baseline --no-roofline"
run() { echo "$1: $(env $1 python bench.py $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"; }
run "PFPP_X=0"
run "PFPP_SIDE_CU_FRACTION_PCT=30"
run "PFPP_SIDE_CU_FRACTION_PCT=50"
run "PFPP_SIDE_CU_FRACTION_PCT=70"
run "PFPP_ENC_CU_FRACTION_PCT=45"
run "PFPP_ENC_CU_FRACTION_PCT=55"
run "PFPP_BENCH_OPT_IN_BWD=0"
run "PFPP_X=1"
