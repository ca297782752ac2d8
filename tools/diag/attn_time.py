"""dense (global) attention of the benchmarked batch stand-alone: forward (with lse), backward (dq + dk/dv passes) timed with one event
pair around N back-to-back calls, with an environment switch at 0 / 1 in the same process (PFPP_ATTN_AB names it; round 6 used it for two
experiments that are not in the library: a distance-2 row prefetch and 8-wave workgroups) and results compared bit for bit; then the time
by sequence length.     PFPP_ATTN_AB=PFPP_SOME_SWITCH python tools/diag/attn_time.py [reps]"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
from pfpp_hip import synthetic, train_ops as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
valid = synthetic.make_batch(0, 32, num_points=16)["part_valids"].sum(1).int()      # bench.py's batch: fragments per puzzle
seq_len = (valid * 25).to(torch.int32)
seq_off = (torch.cumsum(seq_len, 0) - seq_len).to(torch.int32)
M, H, dh = int(seq_len.sum()), 8, 64
max_len = int(seq_len.max())
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(M, 3 * H * dh, device=dev, generator=g)
dout = torch.randn(M, H * dh, device=dev, generator=g) * 1e-3
seq_off, seq_len = seq_off.to(dev), seq_len.to(dev)
scale = dh ** -0.5
print(f"{M} tokens, {len(valid)} sequences, longest {max_len}")


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {}
for pf in ("0", "1", "0", "1"):
    os.environ[os.environ.get("PFPP_ATTN_AB", "PFPP_ATTN_AB_UNUSED")] = pf
    out, outp, lse = T.attn_dense_train_planes(qkv, seq_off, seq_len, max_len, H, dh, scale)
    dq = T.attn_dense_bwd_planes(qkv, out, dout, lse, seq_off, seq_len, max_len, H, dh, scale, 4096.0)
    t_f = timed(lambda: T.attn_dense_train_planes(qkv, seq_off, seq_len, max_len, H, dh, scale))
    t_b = timed(lambda: T.attn_dense_bwd_planes(qkv, out, dout, lse, seq_off, seq_len, max_len, H, dh, scale, 4096.0))
    print(f"switch={pf}: forward {t_f:6.1f} us   backward (dq + dkv) {t_b:6.1f} us")
    res.setdefault(pf, (out.clone(), lse.clone(), dq.hi.clone(), dq.lo.clone()))
same = all(torch.equal(a, b) for a, b in zip(res["0"], res["1"]))
print("results bit-identical between the two:", same)
if not same:
    for nm, a, b in zip(("out", "lse", "dq.hi", "dq.lo"), res["0"], res["1"]):
        print(f"  {nm}: max |diff| {float((a.float() - b.float()).abs().max()):.3e} of {float(a.float().abs().max()):.3e}")

# per-tile cost of the walk: 32 sequences of one uniform length each (32 keys per tile)
print("uniform lengths (32 sequences): tokens per sequence -> forward us, backward us")
for Tn in (32, 64, 128, 192, 256, 384, 512):
    sl = torch.full((32,), Tn, dtype=torch.int32)
    so = (torch.cumsum(sl, 0) - sl).to(torch.int32)
    Mu = 32 * Tn
    q2 = torch.randn(Mu, 3 * H * dh, device=dev, generator=g)
    d2 = torch.randn(Mu, H * dh, device=dev, generator=g) * 1e-3
    sl, so = sl.to(dev), so.to(dev)
    o2, _, l2 = T.attn_dense_train_planes(q2, so, sl, Tn, H, dh, scale)
    tf = timed(lambda: T.attn_dense_train_planes(q2, so, sl, Tn, H, dh, scale))
    tb = timed(lambda: T.attn_dense_bwd_planes(q2, o2, d2, l2, so, sl, Tn, H, dh, scale, 4096.0))
    print(f"  {Tn:4d}: {tf:6.1f}  {tb:6.1f}")
