cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do for m in "" "--compact"; do echo -n "evaltab=$v $m: "; PFPP_SA_EVAL_UTAB=$v python bench.py --mode sample $m --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; done; done; done
for v in 0 1; do echo -n "stress evaltab=$v: "; PFPP_SA_EVAL_UTAB=$v python bench.py --mode stress --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; done
