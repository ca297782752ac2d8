"""experiment: the sampler step of 32 puzzles as two independent groups of 16 on two streams (inference shards by puzzle: no exchange)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "puzzlefusion-plusplus_amd")
import torch
import bench
dev = torch.device("cuda:0")
compact = "--compact" in sys.argv
def run(groups, steps=30, warm=5):
    B = 32 // groups
    wls = [bench.SamplerWorkload(B, 1024, None, first_id=g * B, dev=dev, compact=compact) for g in range(groups)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
    def step():
        for wl, st in zip(wls, streams):
            with torch.cuda.stream(st):
                wl.step()
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nf = sum(w.n_frag for w in wls)
    return dt * 1e3, nf / dt
for g in (1, 2, 4, 1, 2):
    ms, v = run(g)
    print(f"groups {g}: {ms:.3f} ms per step of 32 puzzles, {v:.0f} fragment*steps/s")
