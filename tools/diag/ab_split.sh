for cfg in "X=0" "PFPP_SPLIT_ACT=1"; do
  echo "== $cfg"
  env $cfg python bench.py --mode sample --compact --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact', d['ms_per_step'])"
  env $cfg python bench.py --mode sample --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'])"
done
