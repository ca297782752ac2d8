"""lnlin_small under rocprofv3: what does a launch cost with warm / cold weights and warm / cold instruction cache?
   python tools/diag/lnlin_cases.py <case>   case: same | rotate | rotate_evict"""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
import torch
from pfpp_hip import ops
from pfpp_hip.packing import PW
case = sys.argv[1]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, L, M, N = 512, 25, 400, 1536
x = torch.randn(M, C, generator=g).to(dev)
mod = (torch.randn(1, 2 * C, generator=g) * 0.3).to(dev)
fb = torch.zeros((M + L - 1) // L, dtype=torch.int32, device=dev)
n_sets = 1 if case == "same" else 48
pws = [PW((torch.randn(N, C, generator=g) / math.sqrt(C)).to(dev).contiguous()) for _ in range(n_sets)]
big = torch.randn(4096, 4096, device=dev)
for i in range(300):
    ops.layernorm_linear_small(x, pws[i % n_sets], mod=mod, group_batch=fb, group_rows=L)
    if case == "rotate_evict":      # a few unrelated kernels with code of their own in between
        y = torch.sort(big[:64], dim=1)[0]; y = torch.cumsum(y, 1); y = torch.softmax(y, 1); y = y @ big[:, :64]; y = torch.erfinv(y.clamp(-0.9, 0.9))
torch.cuda.synchronize()
