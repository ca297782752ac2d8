# weight-gradient / split input-gradient GEMMs: tile-major vs chunk-major assignment of the split-K workgroups to the XCDs
python -m pytest tests/test_gpu_train_ops.py -q -x -m gpu 2>&1 | tail -2
PFPP_GRAD_KXCD=1 python -m pytest tests/test_gpu_train_ops.py -q -x -m gpu 2>&1 | tail -2
for cfg in "PFPP_GRAD_KXCD=0" "PFPP_GRAD_KXCD=1" "PFPP_GRAD_KXCD=1 PFPP_GRAD_MINK=512" "PFPP_GRAD_KXCD=1 PFPP_GRAD_WG=512 PFPP_GRAD_MINK=512" "PFPP_GRAD_KXCD=1 PFPP_GRAD_WG=384 PFPP_GRAD_MINK=768" "PFPP_GRAD_KXCD=1 PFPP_GRAD_WG=512 PFPP_GRAD_MINK=1024"; do
  echo "== $cfg"
  for rep in 1 2; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
  env $cfg python bench.py --steps 40 --warmup 5 --serial --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])"
done
