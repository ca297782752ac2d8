for cfg in "X=0" "PFPP_GEMM_SMALL=512" "PFPP_GEMM_SMALL=2048" "PFPP_TRAIN_GROUP_DW=1" "PFPP_TRAIN_GROUP_DW=1 PFPP_GRAD_GROUP_WG=512" "PFPP_GEMM_BN_TILE=3" "PFPP_GEMM_DEEP=0" "PFPP_GEMM_DEEP=2"; do
  echo "== $cfg"
  for rep in 1 2; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
done
