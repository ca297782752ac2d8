"""where does the batched auto_aggl loop spend its time?  python tools/diag/aggl_profile.py"""
import cProfile, pstats, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import config, synthetic
from puzzlefusion_plusplus.auto_aggl import AutoAgglomerative

dev = torch.device("cuda:0")
torch.manual_seed(4321)
model = AutoAgglomerative(config.auto_aggl_config()).to(dev).eval()
with torch.no_grad():
    model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
puzzles = []
for i in range(33):
    b = {k: v.to(dev) for k, v in synthetic.make_batch(500 + i, 1, num_points=1000).items()}
    b.update(synthetic.make_matching(b, seed=i))
    puzzles.append(b)
model.test_step(puzzles[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = model.test_batch(puzzles[1:])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"32 puzzles in {dt:.3f} s = {32 / dt:.1f} puzzles/s, steps {sum(o['steps'] for o in outs)}")
pr = cProfile.Profile(); pr.enable()
outs = model.test_batch(puzzles[1:])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
