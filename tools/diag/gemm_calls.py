"""per-call GEMM durations of one training iteration (serial execution, HIP events), grouped by (kernel, shape)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import collections
import torch
import bench
from pfpp_hip import ops

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev, pipeline=False) if "pipeline" in bench.TrainWorkload.__init__.__code__.co_varnames else bench.TrainWorkload(32, 1024, None, 0, dev)
wl.engine.single_stream()
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
ops.GEMM_TRACE = []
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
agg = collections.OrderedDict()
for e0, e1, flops, name, shape in tr:
    k = (name.split("(")[0][-70:], shape)
    a = agg.setdefault(k, [0, 0.0, flops])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = 0.0
for (name, shape), (n, ms, flops) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / 5
    print(f"{ms / n * 1e3:8.1f} us x {n / 5:5.1f}/step = {ms / 5:6.3f} ms  {flops / (ms / n * 1e-3) / 1e12:6.1f} TF/s  {shape}  {name}")
print(f"total {tot:.3f} ms/step")
