import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import ops
from pfpp_hip.packing import PW
dev = torch.device("cuda:0")
for M, B in ((25, 1), (64, 1), (125, 1), (125, 4)):
    g = torch.Generator().manual_seed(M)
    C, L = 512, 25
    x = (torch.randn(M, C, generator=g) * 2 + 0.3).to(dev)
    mod = (torch.randn(B, 2 * C, generator=g) * 0.3).to(dev)
    fb = torch.randint(0, B, ((M + L - 1) // L,), generator=g).to(torch.int32).to(dev)
    W = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    pw = PW(W.to(dev).contiguous())
    got = ops.layernorm_linear_small(x, pw, mod=mod, group_batch=fb, group_rows=L)
    got2 = ops.layernorm_linear_small(x, pw, mod=mod, group_batch=fb, group_rows=L)
    n = ops.SplitAct.empty(M, C, dev)
    ops.layernorm_grouped(x, mod, fb, L, out=n)
    two = ops.linear(n, pw)
    err = (got - two).abs()
    bad = err > 1e-4 * two.abs().max()
    rows = bad.any(1).nonzero().flatten().tolist()
    cols = bad.any(0).nonzero().flatten().tolist()
    print(M, B, "max err", float(err.max()), "same twice", torch.equal(got, got2), "bad rows", rows[:40], "n bad cols", len(cols), cols[:8])
from pfpp_hip.packing import pack_geglu
for M in (25, 64, 125):
    g = torch.Generator().manual_seed(M)
    C, inner = 512, 2048
    x = (torch.randn(M, C, generator=g) * 2 + 0.3).to(dev)
    W1 = torch.randn(2 * inner, C, generator=g) / math.sqrt(C)
    b1 = torch.randn(2 * inner, generator=g) * 0.1
    gamma, beta = torch.randn(C, generator=g).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    w1p, b1p = pack_geglu(W1.to(dev), b1.to(dev))
    pw1 = PW(w1p)
    u = ops.layernorm_linear_small(x, pw1, gamma=gamma, beta=beta, bias=b1p, geglu=True)
    ub = ops.layernorm_linear_small(x, pw1, gamma=gamma, beta=beta, bias=b1p, geglu=True)
    n = ops.SplitAct.empty(M, C, dev)
    ops.layernorm(x, gamma=gamma, beta=beta, out=n)
    u2 = ops.SplitAct.empty(M, inner, dev)
    ops.linear(n, pw1, b1p, act="geglu", out=u2)
    err = (u.float() - u2.float()).abs()
    bad = err > 1e-4 * u2.float().abs().max()
    print("geglu", M, "max err", float(err.max()), "same twice", torch.equal(u.hi, ub.hi) and torch.equal(u.lo, ub.lo), "bad rows", bad.any(1).nonzero().flatten().tolist()[:40],
          "bad cols", bad.any(0).nonzero().flatten().tolist()[:16], int(bad.sum()))
