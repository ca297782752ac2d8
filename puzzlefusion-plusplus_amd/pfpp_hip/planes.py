"""Split-f16 planes as first-class operands: the tensors the plane GEMM (csrc/gemm_pl.hip, pfpp_gemm_planes) consumes
and the training kernels that produce them directly (include/pfpp.h, "plane-producing forms").

A `Planes` is (hi, lo) fp16 tensors with hi + lo == scale * x to 22 bits; `scale` is a power of two (1 for forward
activations and weights, the gradient scale for dY).  Same bytes as the fp32 tensor it stands for.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT, GemmPlanesArgs, PlanesC, SlabJob, check
from .ops import _chk, _ptr, _stream, raw_stream_id

_f32 = torch.float32
_f16 = torch.float16


class Planes:
    __slots__ = ("hi", "lo", "scale")

    def __init__(self, hi: torch.Tensor, lo: torch.Tensor, scale: float = 1.0):
        if hi.dtype != _f16 or lo.dtype != _f16 or hi.shape != lo.shape or not (hi.is_contiguous() and lo.is_contiguous()):
            raise ValueError("Planes: hi / lo must be contiguous fp16 tensors of one shape")
        if hi.device.type != "cuda":
            raise ValueError("Planes: tensors must live on the GPU (there is no CPU path)")
        self.hi, self.lo, self.scale = hi, lo, float(scale)

    @staticmethod
    def empty(rows: int, cols: int, device, scale: float = 1.0) -> "Planes":
        return Planes(torch.empty((rows, cols), dtype=_f16, device=device), torch.empty((rows, cols), dtype=_f16, device=device), scale)

    @property
    def shape(self):
        return self.hi.shape

    def float(self) -> torch.Tensor:
        """the represented tensor (hi + lo) / scale"""
        return (self.hi.float() + self.lo.float()) / self.scale

    def record_stream(self, s) -> None:
        self.hi.record_stream(s)
        self.lo.record_stream(s)

    def c(self) -> PlanesC:
        return PlanesC(self.hi.data_ptr(), self.lo.data_ptr(), self.scale)


def _pl(p: Optional[Planes]):
    return None if p is None else C.byref(p.c())


def split(x: torch.Tensor, scale: float = 1.0, out: Optional[Planes] = None) -> Planes:
    """planes of scale * x (pfpp_split_planes)"""
    _chk(x, _f32, "x")
    if out is None:
        out = Planes(torch.empty(x.shape, dtype=_f16, device=x.device), torch.empty(x.shape, dtype=_f16, device=x.device), scale)
    out.scale = float(scale)
    check(_lib.load().pfpp_split_planes(_ptr(x), x.numel(), _pl(out), _stream()), "pfpp_split_planes")
    return out


def colsum(p: Planes, out: torch.Tensor) -> torch.Tensor:
    """out[c] += sum over rows of the tensor `p` stands for (bias gradient of a dY given as planes)"""
    _chk(out, _f32, "out")
    rows, cols = p.shape
    check(_lib.load().pfpp_colsum_planes(_ptr(p.hi), _ptr(p.lo), _ptr(out), rows, cols, cols, 1.0 / p.scale, _stream()),
          "pfpp_colsum_planes")
    return out


_WS = {}


def _workspace(device):
    """K-split workspace per (device, stream): launches on one stream are ordered, different streams must not share it"""
    key = (device.index, raw_stream_id(device.index))
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty((24 * 1024 * 1024,), dtype=_f32, device=device)       # 96 MB
        _WS[key] = ws
    return ws


def workspace_for(device, stream_handle: int) -> torch.Tensor:
    """the K-split workspace of launches on the raw stream `stream_handle` (same table as _workspace: a stream has one)"""
    key = (device.index, int(stream_handle))
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty((24 * 1024 * 1024,), dtype=_f32, device=device)
        _WS[key] = ws
    return ws


def gemm(A: Planes, W: Planes, out: torch.Tensor, *, M: int, N: int, K: int, a_kmajor: bool = False, w_kmajor: bool = False,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: str = "none",
         accumulate: bool = False, splits: int = 0, variant: int = 0, alpha: Optional[float] = None, use_ws: bool = True,
         single_pass: bool = False, colsum: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
         defer: Optional["SlabJob"] = None) -> torch.Tensor:
    """pfpp_gemm_planes: out [M, N] = act(alpha * A.W + bias) + residual, or += with accumulate.
       forward  : A [M, K],              W [N, K]
       dX       : A = dY [M, K],         W [K, N] (w_kmajor)
       dW       : A = dY [K, M] (a_kmajor), W = X [K, N] (w_kmajor)
    alpha defaults to 1 / (A.scale * W.scale).  colsum (dW form): colsum[m] += sum over rows of the tensor A stands for (the bias
    gradient), computed inside the same kernel."""
    _chk(out, _f32, "out")
    a = GemmPlanesArgs()
    a.a_hi, a.a_lo, a.w_hi, a.w_lo = A.hi.data_ptr(), A.lo.data_ptr(), W.hi.data_ptr(), W.lo.data_ptr()
    a.C = out.data_ptr()
    a.bias = 0 if bias is None else bias.data_ptr()
    a.residual = 0 if residual is None else residual.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw = A.hi.shape[-1], W.hi.shape[-1]
    a.ldc = out.shape[-1]
    a.ldr = 0 if residual is None else residual.shape[-1]
    a.a_kmajor, a.w_kmajor = int(a_kmajor), int(w_kmajor)
    a.act = ACT[act]
    a.accumulate = int(accumulate)
    a.splits, a.variant = splits, variant
    a.alpha = (1.0 / (A.scale * W.scale)) if alpha is None else alpha
    a.single_pass = int(single_pass)
    if colsum is not None:
        _chk(colsum, _f32, "colsum")
        if colsum.numel() != M or not colsum.is_contiguous():
            raise ValueError("colsum: contiguous fp32 [M] expected")
        a.colsum, a.colsum_alpha = colsum.data_ptr(), 1.0 / A.scale
    if ws is not None:         # the caller's own stretch of workspace (deferred reductions keep theirs until slab_reduce_group)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    elif use_ws:
        ws = _workspace(out.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    if defer is not None:
        a.defer = C.addressof(defer)
    from . import ops

    if ops.GEMM_TRACE is not None:        # bench.py: HIP events around the launch on its stream, attributed to the kernel name
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_gemm_planes(C.byref(a), _stream()), "pfpp_gemm_planes")
        e1.record()
        ops.GEMM_TRACE.append((e0, e1, 2.0 * M * N * K, _lib.load().pfpp_last_gemm_kernel().decode(),
                               (M, N, K, 1, "tn" if a_kmajor else ("nn" if w_kmajor else "nt"), 0)))
        return out
    check(_lib.load().pfpp_gemm_planes(C.byref(a), _stream()), "pfpp_gemm_planes")
    return out


def slab_reduce_group(jobs) -> None:
    """pfpp_slab_reduce_group: the K-split reductions gemm(..., defer=job) handed back, all in one launch"""
    arr = (SlabJob * len(jobs))(*jobs)
    check(_lib.load().pfpp_slab_reduce_group(arr, len(jobs), _stream()), "pfpp_slab_reduce_group")


def reblock_many(items):
    """ONE pfpp_reblock_planes launch for up to PFPP_REBLOCK_MAX weights: items = [(W planes as stored [rows, cols], transposed)] ->
    [(fhi, flo)] — what pfpp_tlayers_fwd / _bwd do at the top of a six-layer call (the traced pass of bench.py issues the same two
    launches per iteration as the timed region)"""
    from . import ops
    from ._lib import PlanesC, ReblockJob

    n = len(items)
    jobs = (ReblockJob * n)()
    outs = []
    for j, (W, transposed) in enumerate(items):
        rows, cols = W.hi.shape
        fhi = torch.empty(rows * cols, dtype=torch.float16, device=W.hi.device)
        flo = torch.empty(rows * cols, dtype=torch.float16, device=W.hi.device)
        jobs[j] = ReblockJob(PlanesC(W.hi.data_ptr(), W.lo.data_ptr(), W.scale), rows, cols, W.hi.shape[-1], fhi.data_ptr(), flo.data_ptr(),
                             int(transposed))
        outs.append((fhi, flo))
    trace = ops.GEMM_TRACE
    if trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.load().pfpp_reblock_planes(jobs, n, _stream()), "pfpp_reblock_planes")
    if trace is not None:
        e1.record()
        trace.append((e0, e1, 0.0, "reblock_kernel", (n, 0, 0, 1, "reblock_many", 0)))
    return outs


def gemm_wd(A: "Planes", W: "Planes", out: torch.Tensor, *, M: int, N: int, K: int, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, transposed: bool = False, frag=None) -> torch.Tensor:
    """the same linear as gemm(...) / gemm(..., w_kmajor=True) through the weight-direct kernel (csrc/gemm_wd.hip), the way
    pfpp_tlayers_fwd / _bwd run it: W's planes are fragment-blocked first (pfpp_reblock_planes; transposed: W is [K, N] and its
    transpose is blocked — the input-gradient form), then pfpp_gemm_wd.  Only the traced pass of bench.py issues it from Python (the
    product path is the C sequencer): both launches are recorded in ops.GEMM_TRACE under their kernel names."""
    from . import ops
    from ._lib import PlanesC, PwC, ReblockJob

    _chk(out, _f32, "out")
    lib = _lib.load()
    rows, cols = (K, N) if transposed else (N, K)               # W as stored
    if frag is not None:                                        # blocked beforehand (reblock_many)
        fhi, flo = frag
    else:
        fhi = torch.empty(N * K, dtype=torch.float16, device=out.device)
        flo = torch.empty(N * K, dtype=torch.float16, device=out.device)
        job = ReblockJob(PlanesC(W.hi.data_ptr(), W.lo.data_ptr(), W.scale), rows, cols, W.hi.shape[-1], fhi.data_ptr(), flo.data_ptr(),
                         int(transposed))
    pw = PwC(None, W.hi.data_ptr(), W.lo.data_ptr(), W.scale, W.hi.shape[-1], fhi.data_ptr(), flo.data_ptr())
    ap = PlanesC(A.hi.data_ptr(), A.lo.data_ptr(), A.scale)
    trace = ops.GEMM_TRACE
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if trace is not None else None
    if ev:
        ev[0].record()
    if frag is None:
        check(lib.pfpp_reblock_planes(C.byref(job), 1, _stream()), "pfpp_reblock_planes")
    if ev:
        ev[1].record(); ev[2].record()
    check(lib.pfpp_gemm_wd(C.byref(ap), A.hi.shape[-1], C.byref(pw), None if bias is None else bias.data_ptr(),
                           None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[-1],
                           out.data_ptr(), out.shape[-1], M, N, K, _stream()), "pfpp_gemm_wd")
    if ev:
        ev[3].record()
        big = N % 256 == 0 and ((M + 127) // 128) * (N // 256) >= 240
        if frag is None:
            trace.append((ev[0], ev[1], 0.0, "reblock_kernel", (rows, cols, 0, 1, "reblock_t" if transposed else "reblock", 0)))
        trace.append((ev[2], ev[3], 2.0 * M * N * K, ops.wd_kernel_name(big, False, (M, N, K)), (M, N, K, 1, "nn" if transposed else "nt", 0)))
    return out


def dw_group(problems, K: int, variant: int = 0) -> None:
    """pfpp_gemm_dw_group: for every (dY planes [K, M], X planes [K, N], gw [M, N], gb [M] or None) of `problems` (at most 8):
    gw += dY^T . X / (dY.scale * X.scale), gb += colsum(dY) / dY.scale — the weight gradients of one transformer block in ONE launch,
    each output tile over the whole contraction in one accumulator chain (no K split, no workspace)."""
    from ._lib import DwJob, PlanesC

    n = len(problems)
    jobs = (DwJob * n)()
    for j, (dy, x, gw, gb) in enumerate(problems):
        _chk(gw, _f32, "gw")
        if gb is not None:
            _chk(gb, _f32, "gb")
        M, N = gw.shape
        if dy.hi.shape[-1] != M or x.hi.shape[-1] != N or dy.hi.shape[0] < K or x.hi.shape[0] < K:
            raise ValueError("dw_group: plane shapes do not match the gradient's")
        jobs[j] = DwJob(PlanesC(dy.hi.data_ptr(), dy.lo.data_ptr(), dy.scale), PlanesC(x.hi.data_ptr(), x.lo.data_ptr(), x.scale),
                        gw.data_ptr(), None if gb is None else gb.data_ptr(), M, N)
    from . import ops

    if ops.GEMM_TRACE is not None:        # bench.py: HIP events around the launch, attributed to the instantiation the library picked
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_gemm_dw_group(jobs, n, K, variant, _stream()), "pfpp_gemm_dw_group")
        e1.record()
        flops = sum(2.0 * K * gw.shape[0] * gw.shape[1] for _, _, gw, _ in problems)
        ops.GEMM_TRACE.append((e0, e1, flops, _lib.load().pfpp_last_gemm_kernel().decode(),
                               (sum(gw.shape[0] * gw.shape[1] for _, _, gw, _ in problems), n, K, 1, "tn-group", 0)))
        return
    check(_lib.load().pfpp_gemm_dw_group(jobs, n, K, variant, _stream()), "pfpp_gemm_dw_group")
