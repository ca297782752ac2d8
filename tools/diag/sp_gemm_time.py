"""single-pass fp16 plane GEMMs of the stress workload (BASELINE configs[4]: 20,000 denoiser tokens x 512, 39,600 verifier tokens x 256):
tiled plane kernel (ops.linear with ops.SINGLE_PASS) against the weight-direct kernel's single-pass form, per shape, with the HBM floor of
the shape (A hi plane + W hi plane + fp32 output [+ residual])  —  python tools/diag/sp_gemm_time.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import _lib, ops
from pfpp_hip.packing import PW, split_f16

dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.SINGLE_PASS = True


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K, res in ((20000, 1536, 512, False), (20000, 512, 512, True), (20000, 512, 2048, True), (39600, 768, 256, False),
                     (39600, 256, 256, True), (39600, 256, 1024, True), (39600, 1024, 256, False)):
    A = torch.randn(M, K, device=dev)
    a = ops.SplitAct(*split_f16(A))
    pw = PW(torch.randn(N, K, device=dev) / K ** 0.5)
    b = torch.randn(N, device=dev)
    h = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    t_pl = timeit(lambda: ops.gemm(a, pw, M=M, N=N, K=K, lda=K, out=out, ldc=N, bias=b, residual=h, ldr=N) if res else ops.linear(a, pw, b, out=out))
    k_pl = _lib.load().pfpp_last_gemm_kernel().decode()[:60]
    t_wd = timeit(lambda: ops.gemm_wd(a, pw, bias=b, residual=h, out=out, single_pass=True)) if (N % 128 == 0 and K % 32 == 0) else float("nan")
    bytes_ = M * K * 2 + N * K * 2 + M * N * 4 * (2 if res else 1)
    fl = 2.0 * M * N * K
    print(f"M {M} N {N} K {K} res {int(res)}: tiled {t_pl:7.1f} us ({fl / t_pl / 1e6:6.0f} TF/s)  weight-direct {t_wd:7.1f} us ({fl / t_wd / 1e6:6.0f} TF/s)  "
          f"HBM floor {bytes_ / 5.8e6:6.1f} us at 5.8 TB/s, matrix floor {fl / 2500e6:5.1f} us   [{k_pl}]", flush=True)
