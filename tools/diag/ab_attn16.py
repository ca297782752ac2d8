"""dense attention: exact-fp32 MFMA kernel vs the split-f16 kernel (PFPP_ATTN_F16X3=1), accuracy against float64 and time"""
import os, sys, math, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) == 1:
    for v in ("1",):
        subprocess.run([sys.executable, __file__, v], env=dict(os.environ, PFPP_ATTN_F16X3=v))
    sys.exit()
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import ops, synthetic
dev = torch.device("cuda:0")
print("== PFPP_ATTN_F16X3 =", sys.argv[1])
H, dh = 8, 64
for name, lens, masked in (("compact (ragged)", None, False), ("all slots (32 x 500, masked)", [500] * 32, True)):
    g = torch.Generator().manual_seed(1)
    if lens is None:
        pv = synthetic.make_batch(0, 32, num_points=64)["part_valids"].sum(1).long().tolist()
        lens = [int(n) * 25 for n in pv]
    rows = sum(lens)
    qkv = torch.randn(rows, 3 * H * dh, generator=g)
    offs = [sum(lens[:i]) for i in range(len(lens))]
    so = torch.tensor(offs, dtype=torch.int32, device=dev); sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    kv = None
    if masked:
        kvb = torch.rand(len(lens), max(lens), generator=g) < 0.3
        kvb[:, 0] = True
        kv = kvb.to(torch.uint8).contiguous().to(dev)
    qd = qkv.to(dev)
    scale = 1 / math.sqrt(dh)
    out = ops.attn_dense(qd, so, sl, max(lens), H, dh, scale, kv)
    # float64 reference on a few sequences
    err = 0.0
    for b in (0, len(lens) // 2, len(lens) - 1):
        x = qkv[offs[b]:offs[b] + lens[b]].double().view(lens[b], 3, H, dh)
        q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
        sc = q @ k.transpose(1, 2) * scale
        if masked:
            sc = sc.masked_fill(~kvb[b, :lens[b]][None, None, :].expand(H, lens[b], lens[b]), float("-inf"))
        ref = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(lens[b], H * dh)
        err = max(err, float((out[offs[b]:offs[b] + lens[b]].double().cpu() - ref).abs().max() / ref.abs().max()))
    for _ in range(5): ops.attn_dense(qd, so, sl, max(lens), H, dh, scale, kv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.attn_dense(qd, so, sl, max(lens), H, dh, scale, kv)
    e1.record(); torch.cuda.synchronize()
    print(f"  {name}: rows {rows}, rel err vs float64 {err:.2e}, {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
