"""the frozen encoder in train mode alone (BASELINE configs[1] batch): wall time per pass on the whole chip and the per-launch
table of the set-abstraction stages (ops.GEMM_TRACE), for PFPP_SA_TRAIN_WIDE / PFPP_SA_TRAIN_CHAIN A/Bs"""
import sys, time, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
import bench
from pfpp_hip import ops

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev, pipeline=False)
m, d = wl.model, wl.data
def one():
    with torch.no_grad():
        return m._extract_features(d["part_pcs"], d["part_valids"], wl.gt)
for _ in range(5): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): one()
torch.cuda.synchronize()
print(f"encoder pass {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
ops.GEMM_TRACE = []
for _ in range(5): one()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, flops, name, shape in ops.GEMM_TRACE:
    agg.setdefault((name, shape[:3]), []).append(e0.elapsed_time(e1))
tot = 0.0
for (name, shape), ts in agg.items():
    us = sum(ts) / 5 * 1e3
    tot += us
    print(f"{us:9.1f} us/pass  x{len(ts) // 5:3d}  {shape}  {name}")
print(f"{tot:9.1f} us/pass traced")
