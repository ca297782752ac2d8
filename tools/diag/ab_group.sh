set -x
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
for cfg in "PFPP_TRAIN_GROUP_DW=0" "PFPP_TRAIN_GROUP_DW=1" "PFPP_GRAD_GROUP_WG=320" "PFPP_GRAD_GROUP_WG=768" "PFPP_GRAD_GROUP_WG=1024" "PFPP_GRAD_GROUP_WG=1536"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --serial | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])"
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --latents-given | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('latents-given', d['ms_per_step'])"
done
