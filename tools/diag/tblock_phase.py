"""one layer per launch of pfpp_tblock_small with phases left out (PFPP_TBLOCK_SKIP) -> per-phase cost at hot weights"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import config, ops
from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DenoiserTransformer(config.denoiser_config()).to(dev).eval()
pk = m.packed()
nl = int(os.environ.get("NL", "1"))
out = []
for n in (2, 8, 20):
    M = 25 * n
    h = torch.randn(M, 512, device=dev)
    mods = torch.randn(12, 1, 1024, device=dev) * 0.1
    frag_b = torch.zeros(n, dtype=torch.int32, device=dev)
    so = torch.zeros(1, dtype=torch.int32, device=dev); sl = torch.full((1,), M, dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.tblock_small(pk, h.clone(), mods[:2 * nl], frag_b, so, sl, L=25, num_layers=nl, num_heads=8, att_scale=0.125)
    torch.cuda.synchronize()
    hs = [h.clone() for _ in range(40)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for hh in hs:
        ops.tblock_small(pk, hh, mods[:2 * nl], frag_b, so, sl, L=25, num_layers=nl, num_heads=8, att_scale=0.125)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M} tok {e0.elapsed_time(e1) / 40 * 1e3:6.1f}")
print(f"skip={os.environ.get('PFPP_TBLOCK_SKIP', '0'):>3s} layers={nl}: " + " | ".join(out))
