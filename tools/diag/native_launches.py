"""Which lines of OUR host code still launch torch-native kernels (fills, copies, index ops, RNG) inside the benchmarked training
iteration: torch.profiler over a few iterations, every non-pfpp kernel attributed to the innermost frame under pfpp_hip/ or bench.py.
    python tools/diag/native_launches.py [iterations]"""
import collections
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import bench  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, first_id=0, dev=dev)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(n_it):
        wl.step()
    torch.cuda.synchronize()
by_site = collections.Counter()
time_site = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    names = [k.name for k in ev.kernels]
    if all(("pfpp" in n or "anonymous namespace" in n or "_GLOBAL__N_" in n) for n in names):
        continue
    site = "?"
    for fr in ev.stack or []:
        if "pfpp_hip/" in fr or "bench.py" in fr or "puzzlefusion_plusplus/" in fr:
            site = fr.split("puzzlefusion-plusplus_amd/")[-1]
            break
    key = (site, ev.name, names[0][:60])
    by_site[key] += 1
    time_site[key] += sum(k.duration for k in ev.kernels)
print(f"torch-native launches per iteration (over {n_it} iterations):")
tot_n = tot_t = 0
for key, n in sorted(by_site.items(), key=lambda kv: -time_site[kv[0]]):
    print(f"  {n / n_it:5.2f} x {time_site[key] / n:6.1f} us  {key[1]:28s} {key[2]:62s} {key[0]}")
    tot_n += n
    tot_t += time_site[key]
print(f"total {tot_n / n_it:.1f} launches, {tot_t / n_it:.1f} us per iteration")
