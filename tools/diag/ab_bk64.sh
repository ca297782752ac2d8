# K-tile of 64 (PFPP_PL_BK64) on / off: training iteration, sampler step (all slots / compact), per-GEMM times
cd $GRAFT_REPO_ROOT
B="--steps 30 --no-cpu-baseline --no-roofline"
run() { echo "$1 $2: $(env $1 python bench.py $B $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"; }
for v in 1 0 1 0; do run "PFPP_PL_BK64=$v" ""; done
for v in 1 0; do run "PFPP_PL_BK64=$v" "--mode sample"; run "PFPP_PL_BK64=$v" "--mode sample --compact"; run "PFPP_PL_BK64=$v" "--serial"; done
PFPP_PL_BK64=1 python tools/diag/gemm_calls.py 2>/dev/null | head -24
