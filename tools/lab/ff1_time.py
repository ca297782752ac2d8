"""ff.net.0.proj + GEGLU + dropout of the training forward: pfpp_gemm_planes then pfpp_geglu_p against pfpp_ff1_geglu_train, each
back to back on one stream (torch events), at the benchmarked 3,850 x 512 -> 2 x 2048."""
import sys
import torch
sys.path.insert(0, "puzzlefusion-plusplus_amd")
from pfpp_hip import planes as P, train_ops as TO

dev = torch.device("cuda:0")
M, K, inner = 3850, 512, 2048
x = torch.randn(M, K, device=dev); w = torch.randn(2 * inner, K, device=dev) / K ** 0.5; b = torch.randn(2 * inner, device=dev) * 0.1
xp, wp = P.split(x, 1.0), P.split(w, 4096.0)
z0 = torch.empty(M, 2 * inner, device=dev)

def timed(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for p_drop in (0.2, 0.0):
    g = timed(lambda: P.gemm(xp, wp, z0, M=M, N=2 * inner, K=K, bias=b))
    a = timed(lambda: TO.geglu_planes(z0, p_drop, 77, 9))
    both = timed(lambda: (P.gemm(xp, wp, z0, M=M, N=2 * inner, K=K, bias=b), TO.geglu_planes(z0, p_drop, 77, 9)))
    f = timed(lambda: TO.ff1_geglu_train(xp, wp, b, inner, p_drop, 77, 9))
    print(f"p_drop {p_drop}: gemm {g:.1f} us, geglu {a:.1f} us, gemm+geglu {both:.1f} us, fused {f:.1f} us (incl. two torch.empty per call)")
