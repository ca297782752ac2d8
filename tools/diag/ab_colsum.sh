python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py -q -x -m gpu 2>&1 | tail -3
for cfg in "PFPP_TRAIN_FUSE_COLSUM=0" "PFPP_TRAIN_FUSE_COLSUM=1"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'], d['extra']['final_loss'])"
  done
  env $cfg python bench.py --steps 40 --warmup 5 --serial --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])"
done
