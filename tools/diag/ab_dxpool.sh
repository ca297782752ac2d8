python -m pytest tests/test_gpu_train.py -q -x -m gpu 2>&1 | tail -2
for cfg in "PFPP_TRAIN_DX_POOL=0" "PFPP_TRAIN_DX_POOL=1"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
done
