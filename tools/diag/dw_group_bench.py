"""The six weight gradients of one transformer block (C = 512, inner = 2048) at K = 3,850 tokens: six pfpp_gemm_planes launches (K split
through slabs + a reduction each: what pfpp_tlayers_bwd issued through round 4) against ONE pfpp_gemm_dw_group launch, per tile variant.
Checks the grouped result against the separate launches (association of the partial sums differs: <= 2e-6 of the max) and against
float64 on a sample, prints us per block and TFLOP/s.    usage: python tools/diag/dw_group_bench.py [K]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p_ in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import torch

from pfpp_hip import planes as P

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3850
dev = torch.device("cuda:0")
C, inner = 512, 2048
G = 4096.0
shapes = [("ff2", C, inner, True), ("ff1", 2 * inner, C, True), ("o2", C, C, True), ("qkv2", 3 * C, C, False), ("o1", C, C, True), ("qkv1", 3 * C, C, False)]
torch.manual_seed(0)
probs = []
for name, m, n, bias in shapes:
    dy = torch.randn(K, m, device=dev) * 1e-3
    x = torch.randn(K, n, device=dev)
    probs.append((name, dy, x, P.split(dy, G), P.split(x, 1.0), bias))
flops = sum(2.0 * K * d.shape[1] * x.shape[1] for _, d, x, _, _, _ in probs)


def separate(gws, gbs):
    for (name, dy, x, dyp, xp, bias), gw, gb in zip(probs, gws, gbs):
        P.gemm(dyp, xp, gw, M=gw.shape[0], N=gw.shape[1], K=K, a_kmajor=True, w_kmajor=True, accumulate=True, colsum=gb if bias else None)


def grouped(gws, gbs, variant, parts=1):
    jobs = [(dyp, xp, gw, gb if bias else None) for (name, dy, x, dyp, xp, bias), gw, gb in zip(probs, gws, gbs)]
    if parts == 1:
        P.dw_group(jobs, K, variant)
    else:
        P.dw_group(jobs[:2], K, variant)
        P.dw_group(jobs[2:], K, variant)


def fresh():
    return ([torch.zeros(d.shape[1], x.shape[1], device=dev) for _, d, x, _, _, _ in probs], [torch.zeros(d.shape[1], device=dev) for _, d, _, _, _, _ in probs])


def timed(fn, n=30):
    gws, gbs = fresh()
    for _ in range(3):
        fn(gws, gbs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(gws, gbs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref_w, ref_b = fresh()
separate(ref_w, ref_b)
torch.cuda.synchronize()
# float64 on a sample of rows of each problem
for (name, dy, x, dyp, xp, bias), gw, gb in zip(probs, ref_w, ref_b):
    rows = torch.arange(0, dy.shape[1], max(1, dy.shape[1] // 16), device=dev)[:16]
    w64 = dy[:, rows].double().t() @ x.double()
    e = float((gw[rows].double() - w64).abs().max() / w64.abs().max())
    print(f"separate {name}: vs float64 {e:.2e}")
t_sep = timed(separate)
print(f"K={K}: separate launches {t_sep:.1f} us per block = {flops / t_sep / 1e6:.1f} TFLOP/s")
for variant in (3, 6, 7, 2):
    for parts in (1, 2):
        gws, gbs = fresh()
        grouped(gws, gbs, variant, parts)
        torch.cuda.synchronize()
        ew = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(gws, ref_w))
        eb = max(float((a - b).abs().max() / max(1e-30, float(b.abs().max()))) for (a, b), pr in zip(zip(gbs, ref_b), probs) if pr[5])
        gws2, gbs2 = fresh()
        grouped(gws2, gbs2, variant, parts)
        torch.cuda.synchronize()
        det = all(torch.equal(a, b) for a, b in zip(gws, gws2)) and all(torch.equal(a, b) for a, b in zip(gbs, gbs2))
        t = timed(lambda a, b: grouped(a, b, variant, parts))
        print(f"grouped variant {variant} in {parts} launch(es): {t:.1f} us = {flops / t / 1e6:.1f} TFLOP/s; vs separate: dW {ew:.2e} db {eb:.2e}; "
              f"run-to-run bit-identical: {det}")
# accumulate semantics: a second call adds
gws, gbs = fresh()
grouped(gws, gbs, 3)
grouped(gws, gbs, 3)
torch.cuda.synchronize()
print("accumulates:", max(float((a - 2 * b).abs().max() / b.abs().max()) for a, b in zip(gws, ref_w)))
