"""single-shape GEMM microbenchmark (developer tool):  python tools/gemm_bench.py M N K [mode] [act] [iters]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch

from pfpp_hip import ops
from pfpp_hip.packing import PW, pack_geglu

M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
act = sys.argv[5] if len(sys.argv) > 5 else "none"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
if act == "geglu":
    W, b = pack_geglu(W, b)
pw = PW(W)
for _ in range(3):
    ops.linear(A, pw, b, act=act, mode=mode)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.linear(A, pw, b, act=act, mode=mode)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"M{M} N{N} K{K} {mode} {act}: {ms * 1e3:.1f} us  {2.0 * M * N * K / (ms * 1e-3) / 1e12:.1f} TFLOP/s (algorithmic)")
