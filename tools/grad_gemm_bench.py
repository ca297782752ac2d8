"""backward-GEMM microbenchmark (developer tool): python tools/grad_gemm_bench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch

from pfpp_hip import train_ops as T

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 3850
for n_out, k_in in ((1536, 512), (4096, 512), (512, 2048), (512, 512)):
    dY = torch.randn(tokens, n_out, device=dev) * 1e-3
    X = torch.randn(tokens, k_in, device=dev)
    W = torch.randn(n_out, k_in, device=dev) * 0.03
    dW = torch.zeros(n_out, k_in, device=dev)
    fl = 2.0 * tokens * n_out * k_in
    for split in (1, 2, 4, 8, 15, 0):
        us = timeit(lambda: T.gemm_grad(dY, X, dW, M=n_out, N=k_in, K=tokens, lda=n_out, ldw=k_in, ldc=k_in, a_kmajor=True,
                                        w_kmajor=True, accumulate=True, split_k=split, a_scale=4096.0))
        print(f"dW [{n_out}x{k_in}] K={tokens} split {split:2d}: {us:7.1f} us {fl / us / 1e6:6.1f} TF/s")
    dX = torch.zeros(tokens, k_in, device=dev)
    for split in (1, 2, 4, 0):
        us = timeit(lambda: T.gemm_grad(dY, W, dX, M=tokens, N=k_in, K=n_out, lda=n_out, ldw=k_in, ldc=k_in, w_kmajor=True,
                                        accumulate=True, split_k=split, a_scale=4096.0))
        print(f"dX [{tokens}x{k_in}] K={n_out} split {split:2d}: {us:7.1f} us {fl / us / 1e6:6.1f} TF/s")
