python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2
for cfg in "PFPP_GEMM_1STAGE=0" "PFPP_GEMM_1STAGE=1"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
  env $cfg BENCH_GEMM_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep "1261568, 64, 4,"
done
