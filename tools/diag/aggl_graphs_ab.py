"""auto_aggl full loop, one puzzle in flight: eager launches vs HIP-graph replay of the per-step work (PFPP_AGGL_GRAPHS)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch, bench
dev = torch.device("cuda:0")
for rep in range(2):
    for g in ("0", "1"):
        os.environ["PFPP_AGGL_GRAPHS"] = g
        r = bench.aggl_puzzles_per_s(dev, n_puzzles=4)
        print(f"graphs={g}: {r['value']} puzzles/s, {r['ddpm_steps']} steps")
