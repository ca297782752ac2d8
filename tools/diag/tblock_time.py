"""duration of the small-token transformer kernel (pfpp_tblock_small) against the layer-wise loop, and with single phases left out
(PFPP_TBLOCK_SKIP bit mask: 1 qkv1, 2 block-diagonal attention, 4 out-proj 1, 8 qkv2, 16 global attention, 32 out-proj 2, 64 GEGLU
projection, 128 second feed-forward linear)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import config, ops
from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DenoiserTransformer(config.denoiser_config()).to(dev).eval()
m.compact_padded = True
def run(n, small, reps=40):
    ops.TBLOCK_SMALL = small
    valid = torch.zeros(1, 20, device=dev); valid[0, :n] = 1
    x = torch.randn(1, 20, 7, device=dev); lat = torch.randn(1, 20, 25, 64, device=dev); xyz = torch.rand(1, 20, 25, 3, device=dev)
    sc = torch.rand(1, 20, 1, device=dev) + 0.5; ref = torch.zeros(1, 20, dtype=torch.bool, device=dev); ref[0, 0] = True
    ts = torch.full((1,), 500, device=dev); ts._pfpp_t = 500
    with torch.no_grad():
        for _ in range(5):
            m(x, ts, lat, xyz, valid, sc, ref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m(x, ts, lat, xyz, valid, sc, ref)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (2, 8, 20):
    print(f"fragments {n:2d} ({25 * n:3d} tokens): layer-wise forward {run(n, False):7.1f} us, persistent kernel forward {run(n, True):7.1f} us", flush=True)

# the kernel alone (HIP events around pfpp_tblock_small)
from pfpp_hip import denoiser as D
pk = m.packed()
for n in (2, 8, 20):
    M = 25 * n
    h = torch.randn(M, 512, device=dev)
    mods = torch.randn(12, 1, 1024, device=dev) * 0.1
    frag_b = torch.zeros(n, dtype=torch.int32, device=dev)
    so = torch.zeros(1, dtype=torch.int32, device=dev); sl = torch.full((1,), M, dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.tblock_small(pk, h.clone(), mods, frag_b, so, sl, L=25, num_layers=6, num_heads=8, att_scale=0.125)
    torch.cuda.synchronize()
    hs = [h.clone() for _ in range(40)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for hh in hs:
        ops.tblock_small(pk, hh, mods, frag_b, so, sl, L=25, num_layers=6, num_heads=8, att_scale=0.125)
    e1.record(); torch.cuda.synchronize()
    print(f"pfpp_tblock_small alone, {M:3d} tokens: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per launch", flush=True)

# hot-weight probe: ONE layer per launch (its 21 MB of planes stay in the caches between launches)
for n in (2, 8, 20):
    M = 25 * n
    h = torch.randn(M, 512, device=dev)
    mods = torch.randn(12, 1, 1024, device=dev) * 0.1
    frag_b = torch.zeros(n, dtype=torch.int32, device=dev)
    so = torch.zeros(1, dtype=torch.int32, device=dev); sl = torch.full((1,), M, dtype=torch.int32, device=dev)
    pk.pop("_tblock_layers", None)
    for _ in range(5):
        ops.tblock_small(pk, h.clone(), mods[:2], frag_b, so, sl, L=25, num_layers=1, num_heads=8, att_scale=0.125)
    torch.cuda.synchronize()
    hs = [h.clone() for _ in range(40)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for hh in hs:
        ops.tblock_small(pk, hh, mods[:2], frag_b, so, sl, L=25, num_layers=1, num_heads=8, att_scale=0.125)
    e1.record(); torch.cuda.synchronize()
    print(f"one layer per launch (hot weights), {M:3d} tokens: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per launch", flush=True)
    pk.pop("_tblock_layers", None)
