"""per-fragment attention backward against float64 autograd: maximum and mean error, and how many rows carry an error far above the mean"""
import sys, math
sys.path[:0] = ['.', 'puzzlefusion-plusplus_amd']
import torch
from pfpp_hip import train_ops as T
dev = torch.device("cuda:0")
Fv, L, H, dh = 40, 25, 8, 64
g = torch.Generator().manual_seed(0)
for amp in (1.0, 0.3):
    qkv = torch.randn(Fv * L, 3 * H * dh, generator=g) * amp
    dO = torch.randn(Fv * L, H * dh, generator=g) * 1e-3
    scale = 1 / math.sqrt(dh)
    got = T.attn_blockdiag_bwd(qkv.to(dev), dO.to(dev), Fv, L, H, dh, scale).cpu().double()
    x = qkv.double().view(Fv, L, 3, H, dh).clone().requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    o = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(Fv * L, H * dh)
    o.backward(dO.double())
    ref = x.grad.reshape(Fv * L, 3 * H * dh)
    d = (got - ref).abs()
    print(f"amp {amp}: max |err| {d.max().item():.3e}  mean {d.mean().item():.3e}  max |ref| {ref.abs().max().item():.3e}  elements above 100 x mean: {int((d > 100 * d.mean()).sum())}")
