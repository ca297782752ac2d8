for cfg in "PFPP_SPLIT_DX=1" "PFPP_SPLIT_DX=0"; do
  echo "== $cfg"
  for rep in 1 2; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --latents-given 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('latents-given', d['ms_per_step'])"
done
