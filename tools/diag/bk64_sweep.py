"""slope / intercept of the plane GEMM against K: 128 x 64 tile with the K-tile of 64 (variant 6) vs 32 (variant 13), 128 x 128 (3 vs 14)"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import planes as P
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 3850
for N in (512, 1536):
    for form in ("nt", "nn"):
        for K in (512, 1024, 2048, 4096):
            x = P.split(torch.randn(M, K, device=dev))
            w = P.split(torch.randn(N, K, device=dev)) if form == "nt" else P.split(torch.randn(K, N, device=dev))
            out = torch.empty(M, N, device=dev)
            row = []
            for v in (15, 6, 16, 3, 17, 18):
                row.append(t(lambda: P.gemm(x, w, out, M=M, N=N, K=K, w_kmajor=(form == "nn"), splits=1, variant=v)))
            print(f"{form} M{M} N{N:5d} K{K:5d}:  128x64 BK64 {row[0]:6.1f}  BK32 {row[1]:6.1f}  2 K-groups {row[4]:6.1f} | 128x128 BK64 {row[2]:6.1f}  BK32 {row[3]:6.1f}  2 K-groups {row[5]:6.1f} us", flush=True)
