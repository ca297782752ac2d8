"""single-shape GEMM microbenchmark (developer tool):  python tools/gemm_bench.py M N K [mode] [act] [iters] [a=planes|fp32]

a=planes (default in the f16x3 mode): the A operand is handed over as split-f16 planes, as the model's kernels do -> the LDS-DMA
plane kernel (csrc/gemm_pl.hip); a=fp32: fp32 activations -> the register-staged kernel that splits them while staging.
Prints both, the kernel instantiation that ran, and whether the two results are bit-identical."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch

from pfpp_hip import _lib, ops
from pfpp_hip.packing import PW, pack_geglu, split_f16

M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
act = sys.argv[5] if len(sys.argv) > 5 else "none"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
which = sys.argv[7].split("=")[-1] if len(sys.argv) > 7 else ("both" if mode == "f16x3" else "fp32")
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
if act == "geglu":
    W, b = pack_geglu(W, b)
pw = PW(W)


def run(a_op, label):
    for _ in range(3):
        out = ops.linear(a_op, pw, b, act=act, mode=mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear(a_op, pw, b, act=act, mode=mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    kern = _lib.load().pfpp_last_gemm_kernel().decode() or "register-staged kernel (csrc/gemm.hip)"
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print(f"M{M} N{N} K{K} {mode} {act} A={label}: {ms * 1e3:.1f} us  {tf:.1f} TFLOP/s algorithmic"
          + (f" = {tf / (2500.0 / 3.0):.3f} of the f16 dense peak / 3" if mode == "f16x3" else "") + f"   [{kern}]")
    return out


outs = {}
if which in ("both", "fp32"):
    outs["fp32"] = run(A, "fp32")
if which in ("both", "planes") and mode == "f16x3":
    outs["planes"] = run(ops.SplitAct(*split_f16(A)), "planes")
if len(outs) == 2:
    print("bit-identical:", bool(torch.equal(outs["fp32"], outs["planes"])))
