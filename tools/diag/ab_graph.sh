for cfg in "X=0" "PFPP_AGGL_GRAPHS=1"; do
  echo "== $cfg"
  env $cfg python - <<'PY'
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'puzzlefusion-plusplus_amd')
import torch, bench
dev = torch.device('cuda:0')
print(json.dumps(bench.aggl_puzzles_per_s(dev, n_puzzles=6)))
print(json.dumps(bench.aggl_puzzles_per_s(dev, n_puzzles=64, in_flight=32)))
PY
done
