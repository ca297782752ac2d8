"""per-fragment self-attention: the dedicated block-diagonal kernels vs the ragged dense kernels with one 25-token sequence per fragment"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import ops, train_ops as T

dev = torch.device("cuda:0")
for Fv in (154, 10):
    L, H, dh = 25, 8, 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(Fv * L, 3 * H * dh, generator=g).to(dev)
    dO = (torch.randn(Fv * L, H * dh, generator=g) * 1e-3).to(dev)
    so = (torch.arange(Fv, dtype=torch.int32) * L).to(dev)
    sl = torch.full((Fv,), L, dtype=torch.int32, device=dev)
    scale = 1 / math.sqrt(dh)

    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    a = ops.attn_blockdiag(qkv, Fv, L, H, dh, scale)
    b = ops.attn_dense(qkv, so, sl, L, H, dh, scale, None)
    print(f"Fv={Fv}: fwd max diff {(a - b).abs().max().item():.2e}",
          f"blockdiag {timeit(lambda: ops.attn_blockdiag(qkv, Fv, L, H, dh, scale)):.1f} us",
          f"dense-per-fragment {timeit(lambda: ops.attn_dense(qkv, so, sl, L, H, dh, scale, None)):.1f} us")
    out, lse = T.attn_dense_train(qkv, so, sl, L, H, dh, scale)
    ga = T.attn_blockdiag_bwd(qkv, dO, Fv, L, H, dh, scale)
    gb = T.attn_dense_bwd(qkv, out, dO, lse, so, sl, L, H, dh, scale)
    print(f"       bwd max diff {(ga - gb).abs().max().item():.2e} (|g| {ga.abs().max().item():.2e})",
          f"blockdiag_bwd {timeit(lambda: T.attn_blockdiag_bwd(qkv, dO, Fv, L, H, dh, scale)):.1f} us",
          f"dense_bwd-per-fragment {timeit(lambda: T.attn_dense_bwd(qkv, out, dO, lse, so, sl, L, H, dh, scale)):.1f} us",
          f"(+ train fwd {timeit(lambda: T.attn_dense_train(qkv, so, sl, L, H, dh, scale)):.1f} us)")
