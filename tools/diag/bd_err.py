import sys, math, os
sys.path[:0] = ['.', 'puzzlefusion-plusplus_amd']
import torch
from pfpp_hip import ops
dev = torch.device("cuda:0")
Fv, L, H, dh = 40, 25, 8, 64
g = torch.Generator().manual_seed(0)
for amp in (1.0, 0.3):
    qkv = (torch.randn(Fv * L, 3 * H * dh, generator=g) * amp)
    scale = 1 / math.sqrt(dh)
    a = ops.attn_blockdiag(qkv.to(dev), Fv, L, H, dh, scale).cpu().double()
    x = qkv.double().view(Fv, L, 3, H, dh)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(Fv * L, H * dh)
    d = (a - ref).abs()
    print(f"amp {amp}: max |err| {d.max().item():.3e}  mean {d.mean().item():.3e}  max |ref| {ref.abs().max().item():.2f}")
    big = (d > 5e-6).nonzero()
    tok = big[:, 0] % L; col = big[:, 1] % dh; head = big[:, 1] // dh
    print("  n big", big.shape[0], "tokens", sorted(set(tok.tolist())), "dims", sorted(set(col.tolist()))[:40], "heads", sorted(set(head.tolist())))
