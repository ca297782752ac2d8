for cfg in "X=0" "PFPP_GRAD_SMALL=0" "PFPP_GRAD_SMALL=24" "PFPP_GRAD_SMALL=80" "PFPP_SPLIT_DX=0" "PFPP_MAIN_HIGH=0" "PFPP_SIDE_PRIORITY=-1" "PFPP_TRAIN_GROUP_DW=1" "PFPP_LN_BWD_ROWS=6"; do
  echo "== $cfg"
  for rep in 1 2 3; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'])"
  done
done
