"""dense attention backward: time of the dq / dk-dv passes against the sequence length (uniform lengths, 32 sequences x 8 heads)
-> fixed cost and cost per 32-row tile of the walk"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import _lib, train_ops as T
from pfpp_hip.ops import _ptr, _stream, check
dev = torch.device("cuda:0")
H, dh = 8, 64
scale = 1 / math.sqrt(dh)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, Tl in ((32, 32), (32, 64), (32, 128), (32, 256), (32, 512), (16, 256), (64, 128), (8, 2500)):
    lens = [Tl] * B
    rows = sum(lens)
    qkv = torch.randn(rows, 3 * H * dh, device=dev)
    dO = torch.randn(rows, H * dh, device=dev) * 1e-3
    so = torch.arange(B, dtype=torch.int32, device=dev) * Tl
    sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    out, lse = T.attn_dense_train(qkv, so, sl, Tl, H, dh, scale)
    dq = torch.empty_like(qkv); dvec = torch.empty_like(lse)
    def part(bits):
        check(_lib.load().pfpp_attn_dense_bwd_parts(_ptr(qkv), _ptr(out), _ptr(dO), _ptr(lse), _ptr(dvec), _ptr(dq), _ptr(so), _ptr(sl),
                                                    None, 0, B, Tl, H, dh, scale, bits, _stream()), "parts")
    part(1)
    print(f"B {B:3d} T {Tl:5d}: D {timeit(lambda: part(1)):6.1f} us  dq {timeit(lambda: part(2)):6.1f} us  dkv {timeit(lambda: part(4)):6.1f} us  "
          f"all {timeit(lambda: T.attn_dense_bwd(qkv, out, dO, lse, so, sl, Tl, H, dh, scale)):6.1f} us  fwd {timeit(lambda: T.attn_dense_train(qkv, so, sl, Tl, H, dh, scale)):6.1f} us")
