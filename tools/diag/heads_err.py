"""diagnostic: fused output heads (csrc/heads.hip) against float64, run-to-run determinism, and which split term an error looks like"""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import train_ops as T
from pfpp_hip._lib import HeadGrads
from pfpp_hip.packing import PW
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
R, C, C2, L, G = 154, 512, 256, 25, 4096.0
pooled = torch.randn(R, C, generator=g).to(dev)
hs, gs, gbufs, raw = [], [], [], []
for n_out in (3, 4):
    W0 = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev); W2 = (torch.randn(C2, C, generator=g) / math.sqrt(C)).to(dev)
    W4 = (torch.randn(n_out, C2, generator=g) / 16).to(dev); b0, b2, b4 = (torch.randn(n, generator=g).to(dev) * 0.1 for n in (C, C2, n_out))
    p0, p2 = PW(W0), PW(W2)
    hs.append(T.head_params(p0, p2, W4, b0, b2, b4)); raw.append((p0, p2, W4, b0, b2, b4))
    gb = [torch.zeros(n_out, C2, device=dev), torch.zeros(n_out, device=dev), torch.zeros(C2, device=dev), torch.zeros(C, device=dev)]
    gbufs.append(gb); gs.append(HeadGrads(*(t.data_ptr() for t in gb)))
out = torch.zeros(R, 7, device=dev)
saved = T.heads_fwd(pooled, hs[0], hs[1], out, save=True)
dout = (torch.randn(R, 7, generator=g) * 1e-3).to(dev)
runs = []
for rep in range(3):
    for gb in gbufs:
        for t in gb: t.zero_()
    da0, da1, dx = T.heads_bwd(dout, hs[0], hs[1], saved, gs[0], gs[1], G, L)
    torch.cuda.synchronize()
    runs.append((da0.clone(), da1.clone(), dx.clone()))
print("deterministic:", [all(torch.equal(a, b) for a, b in zip(runs[0], r)) for r in runs[1:]])
a0, v0, a1, v1 = (t.double() for t in saved)
def sg(v):
    s = torch.sigmoid(v); return s * (1 + v * (1 - s))
for hd, (p0, p2, W4, b0, b2, b4) in enumerate(raw):
    c0, n = (0, 3) if hd == 0 else (3, 4)
    W2e = (p2.hi.double() + p2.lo.double()) / p2.scale; W0e = (p0.hi.double() + p0.lo.double()) / p0.scale
    W2h = p2.hi.double() / p2.scale
    da1_ref = (dout[:, c0:c0 + n].double() @ W4.double()) * sg(a1[hd])
    da0_ref = (da1_ref @ W2e) * sg(a0[hd])
    da0_hi = (da1_ref @ W2h) * sg(a0[hd])
    rel = lambda x, y: float((x.double() - y).abs().max() / y.abs().max())
    print(f"head {hd}: da1 {rel(runs[0][1][hd], da1_ref):.2e}  da0 {rel(runs[0][0][hd], da0_ref):.2e}  (da0 vs W-hi-only reference {rel(runs[0][0][hd], da0_hi):.2e}; hi-only ref vs full {rel(da0_hi, da0_ref):.2e})")
    e = (runs[0][0][hd].double() - da0_ref).abs()
    print("   worst columns:", torch.topk(e.max(0).values, 8).indices.tolist(), " worst rows:", torch.topk(e.max(1).values, 5).indices.tolist())
    print("   err by column block of 32:", [f"{float(e[:, 32*i:32*i+32].max()):.1e}" for i in range(16)])
# timing (HIP events over 50 back-to-back launches)
for name, fn in (("heads_fwd", lambda: T.heads_fwd(pooled, hs[0], hs[1], out, save=True)),
                 ("heads_bwd", lambda: T.heads_bwd(dout, hs[0], hs[1], saved, gs[0], gs[1], G, L))):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (R = {R})")
