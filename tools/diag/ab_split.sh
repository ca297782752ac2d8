timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
PFPP_SPLIT_ACT=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "denoiser or sampler or verifier or auto_aggl" 2>&1 | tail -3
for cfg in "PFPP_SPLIT_ACT=0" "PFPP_SPLIT_ACT=1"; do
  echo "== $cfg"
  for rep in 1 2; do
  env $cfg python bench.py --mode sample --compact --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact', d['ms_per_step'])"
  env $cfg python bench.py --mode sample --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'])"
  done
done
