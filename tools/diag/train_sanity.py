"""end-to-end sanity of the training path as bench.py runs it (optimizer in the backward, fused gradient clear, dynamic gradient
scale, next iteration's encoder on its own stream): N iterations on one fixed synthetic batch — the loss must stay finite and fall"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev)
losses = []
for i in range(n):
    wl.step()
    if i % 25 == 0 or i == n - 1:
        torch.cuda.synchronize()             # the step runs on its own high-priority stream: .item() on the default stream would not wait for it
        losses.append((i, float(wl.last_loss), wl.engine.grad_scale))
torch.cuda.synchronize()
for i, l, g in losses:
    print(f"iteration {i:4d}  loss {l:.5f}  grad_scale 2^{int(torch.log2(torch.tensor(g)))}")
assert all(torch.isfinite(torch.tensor(l)) for _, l, _ in losses)
first = sum(l for _, l, _ in losses[:3]) / 3
last = sum(l for _, l, _ in losses[-3:]) / 3
print("mean of the first / last three samples:", round(first, 4), round(last, 4))
assert last < 0.8 * first, "the loss did not fall"
print("params finite:", bool(torch.isfinite(wl.engine.flat.params).all()))
