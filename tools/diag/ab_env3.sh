# alternating A/B of one environment switch on one box: bash tools/diag/ab_env3.sh NAME [rounds]
cd $GRAFT_REPO_ROOT
N=$1; R=${2:-3}
for r in $(seq 1 $R); do for v in 0 1; do
  env $N=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', l['ms_per_step'], l['value'])"
done; done
