for cfg in "PFPP_GEMM_BN_TILE=0" "PFPP_GEMM_BN_TILE=1" "PFPP_GEMM_BN_TILE=2" "PFPP_GEMM_BN_TILE=3"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
  env $cfg python bench.py --steps 40 --warmup 5 --serial --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])"
done
