# A/B of the encoder stream's CU mask and the persistent-grid cap on the training bench (same box, back to back)
cd $GRAFT_REPO_ROOT
B="--steps 30 --no-cpu-baseline --no-roofline"
run() { echo "$1: $(env $1 python bench.py $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"; }
for pct in 70 60 50 40 30; do run "PFPP_ENC_CU_FRACTION_PCT=$pct"; done
run "PFPP_ENC_CU_FRACTION_PCT=70 PFPP_ENC_WGS_AUTO=0"
run "PFPP_ENC_CU_FRACTION_PCT=50 PFPP_ENC_WGS_AUTO=0"
run "PFPP_ENC_CU_FRACTION_PCT=0"
run "PFPP_ENC_CU_FRACTION_PCT=70"
echo "latents given: $(python bench.py $B --latents-given 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
