set -x
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
PFPP_GEMM=f32 timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/grad_gemm_bench.py 2>&1 | tail -20
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v.get('ms_per_step',v.get('value')) for k,v in d['extra'].items() if isinstance(v,dict)})"
