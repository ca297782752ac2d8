cd $GRAFT_REPO_ROOT
B="--steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
run() { echo "$1: $(env $1 python bench.py $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"; }
for r in 1 2; do for v in 8 4 5 13 25; do run "PFPP_LN_BWD_ROWS=$v"; done; done
