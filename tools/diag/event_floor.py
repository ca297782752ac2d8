"""what a HIP-event pair reads around (a) nothing, (b) a one-element kernel, (c) a ~20 us GEMM — torch events against raw HIP events created with
hipEventDisableSystemFence / hipEventReleaseToDevice (python tools/diag/event_floor.py)"""
import ctypes as C
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import ops, planes as P
from pfpp_hip.packing import PW

dev = torch.device("cuda:0")
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]


class Raw:
    def __init__(self, flags):
        self.e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(self.e), flags) == 0

    def record(self):
        assert hip.hipEventRecord(self.e, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0

    def elapsed_time(self, other):
        ms = C.c_float()
        assert hip.hipEventElapsedTime(C.byref(ms), self.e, other.e) == 0
        return ms.value


one = torch.zeros(1, device=dev)
g = torch.Generator().manual_seed(0)
pl = P.split(torch.randn(3850, 512, generator=g).to(dev))
a = ops.SplitAct(pl.hi, pl.lo)
pw = PW((torch.randn(512, 512, generator=g) / 22.6).to(dev).contiguous())
out = torch.empty(3850, 512, device=dev)
work = {"nothing": lambda: None, "one-element fill": lambda: one.fill_(1.0), "gemm_wd 3850x512x512": lambda: ops.gemm_wd(a, pw, out=out)}
kinds = {"torch.cuda.Event": lambda: torch.cuda.Event(enable_timing=True), "raw default": lambda: Raw(0), "raw DisableSystemFence": lambda: Raw(0x20000000),
         "raw ReleaseToDevice": lambda: Raw(0x40000000)}
for wn, fn in work.items():
    for kn, mk in kinds.items():
        pairs = []
        for _ in range(120):
            e0, e1 = mk(), mk()
            e0.record(); fn(); e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs[20:])
        print(f"{wn:24s} {kn:24s}: median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f}  p90 {v[int(len(v) * 0.9)]:7.2f}", flush=True)
