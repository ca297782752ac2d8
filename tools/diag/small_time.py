"""few-token GEMM (pfpp_gemm_small) at the one-puzzle-in-flight shapes: launches back to back on one stream, each reading the previous
one's output as its residual (a dependent chain like the sampler's), with PFPP_GEMM_SMALL_KS = 0 / 1 in one process"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import ops
from pfpp_hip.packing import PW

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for M in (200, 500, 1000, 2000):
    for N, K in ((512, 512), (512, 2048)):
        x = torch.randn(M, K, generator=g).to(dev)
        from pfpp_hip import planes as P
        pl = P.split(x)
        a = ops.SplitAct(pl.hi, pl.lo)
        pw = PW((torch.randn(N, K, generator=g) / K ** 0.5).to(dev).contiguous())
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N, device=dev)
        res = {}
        line = f"M {M:5d} N {N} K {K:4d}:"
        for ks in ("0", "1", "0", "1"):
            os.environ["PFPP_GEMM_SMALL_KS"] = ks
            out.zero_()
            ops.gemm_small(a, pw, bias=bias, residual=out, out=out)
            res.setdefault(ks, out.clone())
            for _ in range(10): ops.gemm_small(a, pw, bias=bias, residual=out, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): ops.gemm_small(a, pw, bias=bias, residual=out, out=out)
            e1.record(); torch.cuda.synchronize()
            line += f"  ks={ks} {e0.elapsed_time(e1) * 5:6.2f} us"
        ref = x.double().cpu() @ pw.f32.double().cpu().t() + bias.double().cpu()
        line += f"   |ks - chain| {float((res['0'] - res['1']).abs().max()):.2e}, |ks - f64| {float((res['1'].double().cpu() - ref).abs().max()):.2e} of {float(ref.abs().max()):.2f}"
        print(line)
