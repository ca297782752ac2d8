for cfg in "PFPP_GRAD_WG=192 PFPP_GRAD_MINK=768" "PFPP_GRAD_WG=160 PFPP_GRAD_MINK=768" "PFPP_GRAD_WG=144 PFPP_GRAD_MINK=640" "PFPP_GRAD_WG=192 PFPP_GRAD_MINK=1024" "PFPP_GRAD_WG=160 PFPP_GRAD_MINK=1024" "PFPP_GRAD_WG=224 PFPP_GRAD_MINK=640"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
done
python tools/diag/tail_events.py 2>&1 | tail -7
