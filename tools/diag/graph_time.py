import sys, time, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
from pfpp_hip import config, synthetic
from pfpp_hip.denoiser import CompactLayout
from puzzlefusion_plusplus.auto_aggl import AutoAgglomerative
dev = torch.device("cuda:0")
m = AutoAgglomerative(config.auto_aggl_config(), use_graphs=True).to(dev).eval()
b = {k: v.to(dev) for k, v in synthetic.make_batch(500, 1, num_points=1000).items()}
x = torch.randn(1, 20, 7, device=dev)
lay = CompactLayout(b["part_valids"], 25)
print("valid fragments", lay.Fv)
for graphs in (False, True):
    m.use_graphs = graphs
    f = m._make_step(b["part_pcs"], b["part_valids"], b["part_scale"], b["ref_part"], lay, x)
    for _ in range(3):
        f(x, 500)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        f(x, 500)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("graphs" if graphs else "eager ", f"enqueue {(t1 - t0) * 20:.3f} ms/step, total {(t2 - t0) * 20:.3f} ms/step")
