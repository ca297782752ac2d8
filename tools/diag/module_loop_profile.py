"""host-side profile of the module-surface training loop (cProfile over 30 iterations): where the Python time of an iteration goes"""
import sys, time, cProfile, pstats
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
loop(8); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(30); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"module loop: host returns after {t_host / 30 * 1e3:.3f} ms/iteration, GPU done after {t_all / 30 * 1e3:.3f} ms/iteration")
pr = cProfile.Profile(); pr.enable(); loop(30); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
