"""host time per iteration (enqueue only: the loop returns before the GPU is done) of the engine-level loop and of the module-surface loop"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch, bench
from pfpp_hip import config, synthetic
from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser
dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, first_id=0, dev=dev)
for _ in range(8): wl.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30): wl.step()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"engine: host {th / 30 * 1e3:.3f} ms, done {ta / 30 * 1e3:.3f} ms")
torch.manual_seed(1234)
model = Denoiser(config.denoiser_config()).to(dev)
with torch.no_grad():
    model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
for p_ in model.encoder.parameters():
    p_.requires_grad = False
model.train()
opt = model.configure_optimizers()
data = {k: v.to(dev) for k, v in synthetic.make_batch(0, 32, num_points=1024).items()}
tt = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0, "next": 0.0}
def loop(n):
    it = iter(model.training_schedule([data] * n))
    i = 0
    while True:
        a = time.perf_counter()
        try: batch = next(it)
        except StopIteration: break
        b = time.perf_counter(); loss = model.training_step(batch, i)
        c = time.perf_counter(); loss.backward()
        d = time.perf_counter(); opt.step(); opt.zero_grad()
        e = time.perf_counter()
        tt["next"] += b - a; tt["fwd"] += c - b; tt["bwd"] += d - c; tt["opt"] += e - d; i += 1
loop(8); torch.cuda.synchronize()
for rep in range(3):
    for k in tt: tt[k] = 0.0
    t0 = time.perf_counter(); loop(30); th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"module: host {th / 30 * 1e3:.3f} ms, done {ta / 30 * 1e3:.3f} ms; " + ", ".join(f"{k} {v / 30 * 1e3:.2f}" for k, v in tt.items()))
# the engine loop once more, after the module loop: tells a drifting clock from a difference between the two loops
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30): wl.step()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"engine again: host {th / 30 * 1e3:.3f} ms, done {ta / 30 * 1e3:.3f} ms")
for rep in range(2):
    for k in tt: tt[k] = 0.0
    t0 = time.perf_counter(); loop(30); th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"module again: host {th / 30 * 1e3:.3f} ms, done {ta / 30 * 1e3:.3f} ms")
with torch.autograd.set_multithreading_enabled(False):
    loop(8); torch.cuda.synchronize()
    for rep in range(4):
        for k in tt: tt[k] = 0.0
        t0 = time.perf_counter(); loop(30); th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
        print(f"module, backward on the calling thread: host {th / 30 * 1e3:.3f} ms, done {ta / 30 * 1e3:.3f} ms; " + ", ".join(f"{k} {v / 30 * 1e3:.2f}" for k, v in tt.items()))
