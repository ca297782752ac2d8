"""auto_aggl loop, one puzzle in flight: puzzles/s and ms per DDPM step with / without the small-token transformer kernel"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
import bench
from pfpp_hip import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for flag in (False, True, False, True):
    ops.TBLOCK_SMALL = flag
    r = bench.aggl_puzzles_per_s(dev, n_puzzles=6)
    print(f"TBLOCK_SMALL={int(flag)}: {r['value']} puzzles/s, {r['ddpm_steps']} steps, {1e3 * r['puzzles'] / r['value'] / r['ddpm_steps']:.3f} ms per DDPM step")
