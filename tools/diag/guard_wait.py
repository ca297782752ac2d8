"""how long does the main chain stall on the write-after-read guard of the in-place residual-gradient update?
python tools/diag/guard_wait.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev)
eng = wl.engine
orig = eng._before_inplace_update
pairs = []


def guarded():
    if getattr(eng, "_dy_read", None) is None:
        return orig()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    orig()
    e1.record(torch.cuda.current_stream())
    pairs.append((e0, e1))


eng._before_inplace_update = guarded
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
pairs.clear()
n = 20
for _ in range(n):
    wl.step()
torch.cuda.synchronize()
waits = [a.elapsed_time(b) for a, b in pairs]
print(f"{len(waits) / n:.1f} guarded updates per iteration, stall {sum(waits) / n:.3f} ms per iteration, max {max(waits):.3f} ms")
per = len(waits) // n
for j in range(per):
    print(f"  guard {j}: {sum(waits[j::per]) / n * 1e3:7.1f} us")
