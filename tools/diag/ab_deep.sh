for cfg in "PFPP_GEMM_DEEP=1" "PFPP_GEMM_DEEP=3"; do
  echo "== $cfg"
  for shp in "250 512 2048" "125 512 2048" "500 512 2048" "250 512 1024" "3850 512 2048"; do
    env $cfg python tools/gemm_bench.py $shp 2>/dev/null | tail -1
  done
  env $cfg python - <<'PY' 2>/dev/null
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'puzzlefusion-plusplus_amd')
import torch, bench
dev = torch.device('cuda:0')
r = bench.aggl_puzzles_per_s(dev, n_puzzles=6); print('aggl single', r['value'])
PY
done
PFPP_GEMM_DEEP=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or denoiser or verifier or sampler" 2>&1 | tail -2
