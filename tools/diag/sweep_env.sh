# one environment variable over a list of values, two rounds, the benchmarked iteration: bash tools/diag/sweep_env.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
N=$1; shift
for r in 1 2; do for v in "$@"; do
  env $N=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', l['ms_per_step'], l['value'])"
done; done
