"""which host<->device synchronisations does one training / sampler step contain?  python tools/diag/sync_check.py"""
import sys, time, warnings
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

dev = torch.device("cuda:0")
for name, wl in (("train", bench.TrainWorkload(32, 1024, None, 0, dev)),
                 ("train latents given", bench.TrainWorkload(32, 1024, None, 0, dev, latents_given=True)),
                 ("sampler", bench.SamplerWorkload(32, 1024, None, 0, dev))):
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"== {name}: enqueue {1e2 * (t1 - t0):.2f} ms/step, total {1e2 * (t2 - t0):.2f} ms/step", flush=True)
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        wl.step()
    torch.cuda.set_sync_debug_mode("default")
    for x in w:
        print("  sync:", x.filename.split("repo/")[-1], x.lineno, str(x.message)[:80])
    torch.cuda.synchronize()
