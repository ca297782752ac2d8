"""how much of a training step is host enqueue time?  python tools/diag/host_time.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    wl.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e2 * (t1 - t0):.2f} ms/step, total {1e2 * (t2 - t0):.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    wl.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
