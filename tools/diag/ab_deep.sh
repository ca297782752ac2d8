for m in 1 2; do
PFPP_GEMM_DEEP=$m timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or denoiser or verifier or sampler" 2>&1 | tail -2
done
for cfg in "PFPP_GEMM_DEEP=0" "PFPP_GEMM_DEEP=1" "PFPP_GEMM_DEEP=2"; do
  echo "== $cfg"
  for shp in "250 512 512" "250 1536 512" "250 512 2048" "250 4096 512 f16x3 geglu" "3850 512 512" "3850 1536 512" "3850 512 2048"; do
    env $cfg python tools/gemm_bench.py $shp 2>/dev/null | tail -1
  done
  env $cfg python - <<'PY' 2>/dev/null
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'puzzlefusion-plusplus_amd')
import torch, bench
dev = torch.device('cuda:0')
r = bench.aggl_puzzles_per_s(dev, n_puzzles=6); print('aggl single', r['value'])
PY
  env $cfg python bench.py --mode sample --compact --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact sampler', d['ms_per_step'])"
done
