"""engine-level training iteration (bench.TrainWorkload) against the module-surface loop (Denoiser.training_schedule / training_step /
backward / optimizer.step / zero_grad from the default stream) in ONE fresh process, BASELINE configs[1]: best of three 20-iteration
windows of the first, best of five of the second; one JSON line.  tests/test_gpu_train.py runs it as a subprocess."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from pfpp_hip import config, synthetic  # noqa: E402
from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, first_id=0, dev=dev)
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
t_engine = float("inf")
for _ in range(3):          # best of three windows on both sides: one window is at the mercy of whatever else the box does
    t0 = time.perf_counter()
    for _ in range(20):
        wl.step()
    torch.cuda.synchronize()
    t_engine = min(t_engine, (time.perf_counter() - t0) / 20)
del wl

torch.manual_seed(1234)
model = Denoiser(config.denoiser_config()).to(dev)
with torch.no_grad():
    model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
for p_ in model.encoder.parameters():
    p_.requires_grad = False
model.train()
opt = model.configure_optimizers()
assert opt.in_backward
data = {k: v.to(dev) for k, v in synthetic.make_batch(0, 32, num_points=1024).items()}
losses = []


def loop(n, skip=0):
    """-> seconds per iteration of iterations skip .. n - 1 (the first `skip` fill the encoder pipeline of this pass over the loader)"""
    t0 = time.perf_counter()
    for i, batch in enumerate(model.training_schedule([data] * n)):
        if i == skip:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = model.training_step(batch, i)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n - skip)


loop(6)
# steady state of ONE pass over a loader, like the engine-level windows above (a fresh pass starts with an in-line encoder: that
# start-up is the loader's, not the iteration's)
t_module = min(loop(26, skip=6) for _ in range(5))
ls = torch.stack(losses).cpu()
print(json.dumps({"engine_ms": t_engine * 1e3, "module_ms": t_module * 1e3, "losses_finite": bool(torch.isfinite(ls).all()),
                  "loss_first5": float(ls[:5].mean()), "loss_last5": float(ls[-5:].mean())}))
