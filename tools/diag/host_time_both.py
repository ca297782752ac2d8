"""host enqueue time per training iteration against the iteration itself, with and without the encoder in the loop
(python tools/diag/host_time_both.py): `enqueue` = wall time of issuing N iterations without a device synchronisation (the launch queues
are far deeper than an iteration), `total` = until the device is idle"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

dev = torch.device("cuda:0")
for lg in (False, True):
    wl = bench.TrainWorkload(32, 1024, None, 0, dev, latents_given=lg)
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    for n in (10, 30):
        t0 = time.perf_counter()
        for _ in range(n):
            wl.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"latents given {lg}: {n} iterations: enqueue {1e3 * (t1 - t0) / n:.2f} ms/iteration, total {1e3 * (t2 - t0) / n:.2f} ms/iteration", flush=True)
    del wl.engine, wl
