"""fill most of the GPU's free memory with a bit pattern and exit: what a fresh process's torch.empty() may then see (a test for reads of
uninitialised device memory: run it before a test that passes on a clean box).  usage: python tools/diag/pollute.py [value] [fraction]"""
import sys
import torch
val = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
free, total = torch.cuda.mem_get_info()
n = int(free * frac) // 4
chunks = []
left = n
while left > 0:
    m = min(left, 1 << 30)
    chunks.append(torch.full((m,), val, dtype=torch.float32, device="cuda"))
    left -= m
torch.cuda.synchronize()
print(f"filled {n * 4 / 2**30:.1f} GiB with {val}")
