cd $GRAFT_REPO_ROOT
B="--steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
run() { echo "$1: $(env $1 python bench.py $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"; }
for r in 1 2; do for pct in 60 50 45 40 35; do run "PFPP_ENC_CU_FRACTION_PCT=$pct"; done; done
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "encoder or fused_zero" 2>&1 | tail -2
