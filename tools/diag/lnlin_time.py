"""LayerNorm + linear for few rows: the fused launch (csrc/lnlin_small.hip) against the two launches it replaces, back to back on one stream"""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import ops
from pfpp_hip.packing import PW, pack_geglu
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, L = 512, 25
for M in (125, 250, 500):
    x = torch.randn(M, C, generator=g).to(dev)
    mod = (torch.randn(1, 2 * C, generator=g) * 0.3).to(dev)
    fb = torch.zeros((M + L - 1) // L, dtype=torch.int32, device=dev)
    n = ops.SplitAct.empty(M, C, dev)
    for N, geglu in ((1536, False), (4096, True)):
        W = (torch.randn(N, C, generator=g) / math.sqrt(C)).to(dev)
        b = torch.randn(N, generator=g).to(dev) * 0.1
        if geglu:
            W, b = pack_geglu(W, b)
        pw = PW(W.contiguous())
        u = ops.SplitAct.empty(M, N // 2, dev)
        def two():
            ops.layernorm_grouped(x, mod, fb, L, out=n)
            return ops.linear(n, pw, b, act="geglu", out=u) if geglu else ops.linear(n, pw)
        def one():
            return ops.layernorm_linear_small(x, pw, mod=mod, group_batch=fb, group_rows=L, bias=b if geglu else None, geglu=geglu)
        for name, fn in (("two launches", two), ("fused", one)):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"M {M:4d} N {N:5d} {'geglu' if geglu else 'plain'} {name:13s}: {e0.elapsed_time(e1) * 10:.1f} us per call")
