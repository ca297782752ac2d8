for c in "PFPP_GEMM_SPLITK=1" "PFPP_GEMM_SPLITK=0"; do
  echo "== $c"
  env $c python tools/diag/graph_time.py 2>&1 | grep -E "eager|graphs"
  env $c python tools/gemm_bench.py 125 1536 512
  env $c python tools/gemm_bench.py 125 512 2048
  env $c python tools/gemm_bench.py 125 4096 512 f16x3 geglu
  env $c python tools/gemm_bench.py 3850 512 2048
  env $c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
