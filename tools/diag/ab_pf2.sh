for c in "PFPP_GEMM_PF2=1" "PFPP_GEMM_PF2=0"; do
  echo "== $c"
  env $c python tools/diag/graph_time.py 2>&1 | grep eager
  env $c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra']['sampler_step_compact']['ms_per_step'], d['extra']['sampler_step']['ms_per_step'], d['extra']['auto_aggl_full_loop']['value'])"
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
