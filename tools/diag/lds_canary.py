"""Round 5 lab: LDS canary workgroups (tools/lab/lds_canary.hip) on the current stream next to a kernel of the library on a second
stream — do the canaries' LDS words ever change?    usage: python tools/diag/lds_canary.py [--other gemm|planes|wd|fps|none] [--iters N]"""
import argparse
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p_ in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--other", default="gemm")
    ap.add_argument("--lds", type=int, default=12368)
    a = ap.parse_args()
    from pfpp_hip import ops
    from pfpp_hip.packing import PW

    lib = ctypes.CDLL(str(ROOT / "tools/lab/_bin/liblds_canary.so"))
    lib.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    big = torch.randn(4096, 512, generator=g).to(dev)
    wbig = PW(torch.randn(512, 512, generator=g).to(dev))
    x1 = (torch.rand(16, 1024, 3, generator=g) * 2 - 1).to(dev)
    bigp = ops.SplitAct.empty(4096, 512, dev)
    ops.layernorm(big, gamma=torch.ones(512, device=dev), beta=torch.zeros(512, device=dev), out=bigp)
    bad = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for it in range(a.iters):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if a.other == "gemm":
                o = ops.linear(big, wbig)
            elif a.other == "planes":
                o = ops.linear(bigp, wbig)
            elif a.other == "wd":
                o = ops.gemm_wd(bigp, wbig)
            elif a.other == "fps":
                o = ops.fps(x1, 256)
        rc = lib.canary_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 512, a.lds, 40, ctypes.c_void_p(bad.data_ptr()))
        assert rc == 0
        if it % 16 == 15:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"other={a.other}: LDS words found changed {int(bad[0])} (canary workgroups run: {int(bad[1])})")


if __name__ == "__main__":
    main()
