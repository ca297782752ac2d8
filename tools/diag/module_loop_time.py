"""host-side enqueue time per phase of the module-surface training loop (Denoiser.training_schedule) vs its wall time per iteration"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import config, synthetic
from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = Denoiser(config.denoiser_config()).to(dev)
with torch.no_grad():
    model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
for p_ in model.encoder.parameters():
    p_.requires_grad = False
model.train()
opt = model.configure_optimizers()
data = {k: v.to(dev) for k, v in synthetic.make_batch(0, 32, num_points=1024).items()}
ph = {}
def loop(n, rec):
    it = iter(model.training_schedule([data] * n))
    i = 0
    while True:
        t0 = time.perf_counter()
        try:
            batch = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter(); loss = model.training_step(batch, i)
        t2 = time.perf_counter(); loss.backward()
        t3 = time.perf_counter(); opt.step()
        t4 = time.perf_counter(); opt.zero_grad()
        t5 = time.perf_counter()
        if rec:
            for k, v in (("next(prepare)", t1 - t0), ("training_step", t2 - t1), ("backward", t3 - t2), ("opt.step", t4 - t3), ("zero_grad", t5 - t4)):
                ph[k] = ph.get(k, 0.0) + v
        i += 1
loop(6, False)
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(20, True); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"wall {dt * 1e3:.3f} ms/iteration; host enqueue per phase (ms): " + ", ".join(f"{k} {v / 20 * 1e3:.3f}" for k, v in ph.items()),
      f"sum {sum(ph.values()) / 20 * 1e3:.3f}")

# the same loop with the backward on the calling thread (no autograd worker-thread hand-over)
ph.clear()
with torch.autograd.set_multithreading_enabled(False):
    loop(6, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(20, True); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"single-thread autograd: wall {dt * 1e3:.3f} ms/iteration; " + ", ".join(f"{k} {v / 20 * 1e3:.3f}" for k, v in ph.items()),
      f"sum {sum(ph.values()) / 20 * 1e3:.3f}")

# engine-level loop (bench.TrainWorkload): host time per step() vs wall
import bench
wl = bench.TrainWorkload(32, 1024, None, first_id=0, dev=dev)
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); host = 0.0
for _ in range(20):
    a = time.perf_counter(); wl.step(); host += time.perf_counter() - a
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"engine-level: wall {dt * 1e3:.3f} ms/iteration, host {host / 20 * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    wl.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45); st.sort_stats("cumtime").print_stats(45)
