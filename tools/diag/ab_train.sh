for cfg in "X=1" "PFPP_TRAIN_SIDE_STREAM=0"; do
  echo "== $cfg"; env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra'])"
done
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
