timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
PFPP_BN_FUSED=0 timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k batchnorm 2>&1 | tail -2
for cfg in "X=1" "PFPP_BN_FUSED=0"; do
  echo "== $cfg"; env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra'])"
done
