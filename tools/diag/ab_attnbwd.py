"""dense attention backward: exact-fp32 kernels vs split-f16 kernels on the ragged compact token list"""
import os, sys, math, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) == 1:
    for v in ("0", "1"):
        subprocess.run([sys.executable, __file__, v], env=dict(os.environ, PFPP_ATTN_F16X3=v))
    sys.exit()
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import ops, synthetic, train_ops as T
dev = torch.device("cuda:0")
print("== PFPP_ATTN_F16X3 =", sys.argv[1])
H, dh = 8, 64
g = torch.Generator().manual_seed(1)
pv = synthetic.make_batch(0, 32, num_points=64)["part_valids"].sum(1).long().tolist()
lens = [int(n) * 25 for n in pv]
rows = sum(lens)
qkv = torch.randn(rows, 3 * H * dh, generator=g)
dO = torch.randn(rows, H * dh, generator=g) * 1e-3
offs = [sum(lens[:i]) for i in range(len(lens))]
so = torch.tensor(offs, dtype=torch.int32, device=dev); sl = torch.tensor(lens, dtype=torch.int32, device=dev)
scale = 1 / math.sqrt(dh)
qd, dd = qkv.to(dev), dO.to(dev)
out, lse = T.attn_dense_train(qd, so, sl, max(lens), H, dh, scale)
got = T.attn_dense_bwd(qd, out, dd, lse, so, sl, max(lens), H, dh, scale)
err = 0.0
for b in (0, 7, len(lens) - 1):
    x = qkv[offs[b]:offs[b] + lens[b]].double().view(lens[b], 3, H, dh).clone().requires_grad_(True)
    q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    o = (torch.softmax(q @ k.transpose(1, 2) * scale, -1) @ v).transpose(0, 1).reshape(lens[b], H * dh)
    o.backward(dO[offs[b]:offs[b] + lens[b]].double())
    ref = x.grad.reshape(lens[b], 3 * H * dh)
    err = max(err, float((got[offs[b]:offs[b] + lens[b]].double().cpu() - ref).abs().max() / ref.abs().max()))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"  rows {rows}: bwd rel err vs float64 {err:.2e}, bwd {timeit(lambda: T.attn_dense_bwd(qd, out, dd, lse, so, sl, max(lens), H, dh, scale)):.1f} us, "
      f"fwd(train) {timeit(lambda: T.attn_dense_train(qd, so, sl, max(lens), H, dh, scale)):.1f} us")
