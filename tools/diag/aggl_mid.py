"""auto_aggl with a handful of puzzles in flight (600-4000 tokens per step): where should the few-token kernels hand over to the tiled GEMMs?
   PFPP_EVAL_LNLIN_ROWS=<rows> python tools/diag/aggl_mid.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch, bench
dev = torch.device("cuda:0")
for k in tuple(int(x) for x in os.environ.get("KS", "2,4,8").split(",")):
    r = bench.aggl_puzzles_per_s(dev, n_puzzles=2 * k, in_flight=k)
    print("rows", os.environ.get("PFPP_EVAL_LNLIN_ROWS", "512"), "in flight", k, r["value"], "puzzles/s")
