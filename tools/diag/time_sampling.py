import sys
sys.path.insert(0, "."); sys.path.insert(0, "puzzlefusion-plusplus_amd")
import torch
from pfpp_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for F in (8, 154):
    pts = (torch.rand(F, 1024, 3, generator=g) * 2 - 1).to(dev)
    lv = ((256, 0.2, 32), (128, 0.4, 64), (25, 0.8, 64))
    def sep():
        xyz = pts
        for S, r, ns in lv:
            fi, nx = ops.fps(xyz, S); ops.ball_query(xyz, nx, r, ns); xyz = nx
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    print(F, "fused us", timed(lambda: ops.sample_levels(pts, lv)), "separate us", timed(sep),
          "fps1 only", timed(lambda: ops.fps(pts, 256)))
