"""auto_aggl loop, one puzzle in flight: puzzles/s and ms per DDPM step under an A/B of one module attribute or environment switch.
usage: python tools/diag/aggl_ab.py encoder.SA_EVAL_ROWS_MIN 51200 0     (module.attribute value_a value_b; values are ints)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import importlib
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mod_name, attr = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("pfpp_hip." + mod_name)
vals = [int(v) for v in sys.argv[2:4]]
for v in vals * 2:
    setattr(mod, attr, v)
    r = bench.aggl_puzzles_per_s(dev, n_puzzles=6)
    print(f"{sys.argv[1]}={v}: {r['value']} puzzles/s, {r['ddpm_steps']} steps, {1e3 * r['puzzles'] / r['value'] / r['ddpm_steps']:.3f} ms per DDPM step", flush=True)
