"""the module-surface training loop alone (for rocprofv3): 26 iterations through Denoiser.training_schedule"""
import sys, time
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
def loop(n):
    for i, batch in enumerate(model.training_schedule([data] * n)):
        loss = model.training_step(batch, i); loss.backward(); opt.step(); opt.zero_grad()
loop(6); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(20); torch.cuda.synchronize()
print(f"module loop {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/iteration")
if len(sys.argv) > 1 and sys.argv[1] == "profile":       # host time of the loop's own thread (the encoder is issued from a second one)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); loop(20); t1 = time.perf_counter()
    pr.disable(); torch.cuda.synchronize()
    print(f"enqueue {(t1 - t0) / 20 * 1e3:.3f} ms/iteration")
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
