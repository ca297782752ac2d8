"""Tensor-level wrappers over the C ABI (include/pfpp.h).

Each wrapper validates device / dtype / contiguity / shape on the host (the C side
only sees raw pointers), allocates the outputs with torch's caching allocator and
enqueues the kernel on torch's current HIP stream.  Nothing here computes on the
CPU: tensors that are not on a CUDA(HIP) device are rejected.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT, GemmArgs, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> C.c_void_p:
    """torch's current HIP stream of the current device (raw handle; torch.cuda.current_stream() costs ~8 us of
    Python per call, and every kernel wrapper asks)"""
    if STREAM_OVERRIDE is not None:
        return C.c_void_p(STREAM_OVERRIDE)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_cur_dev()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# raw stream handle every wrapper launches on instead of torch's current stream (pfpp_hip.hipstream: side-stream regions of the
# training engine that only launch kernels into existing buffers); None = torch's current stream
STREAM_OVERRIDE: Optional[int] = None


_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device      # the C call, without torch.cuda's lazy-init wrapper


def raw_stream_id(device_index: int) -> int:
    """raw handle of torch's current stream on a device (a cheap dictionary key for per-stream workspaces)"""
    if STREAM_OVERRIDE is not None:
        return STREAM_OVERRIDE
    if _raw_stream is not None:
        return int(_raw_stream(device_index))
    return int(torch.cuda.current_stream(device_index).cuda_stream)


def _chk(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# --------------------------------------------------------------------------- SE(3)
def se3_rotate_gather(part_pcs: torch.Tensor, pose: torch.Tensor, slot: torch.Tensor) -> torch.Tensor:
    """part_pcs [n_slots,N,3], pose [n_slots,7], slot int32 [F] -> [F,N,3] (include/pfpp.h a1)"""
    _chk(part_pcs, torch.float32, "part_pcs"); _chk(pose, torch.float32, "pose"); _chk(slot, torch.int32, "slot")
    n_slots, N, three = part_pcs.shape
    if three != 3 or pose.shape != (n_slots, 7):
        raise ValueError("se3_rotate_gather: part_pcs [n,N,3] and pose [n,7] expected")
    F = slot.numel()
    out = torch.empty((F, N, 3), dtype=torch.float32, device=part_pcs.device)
    check(_lib.load().pfpp_se3_rotate_gather(_ptr(part_pcs), _ptr(pose), _ptr(slot), _ptr(out), F, N, _stream()),
          "pfpp_se3_rotate_gather")
    return out


def pose_apply(pts: torch.Tensor, pose: torch.Tensor, scale: Optional[torch.Tensor] = None,
               normalise: bool = True) -> torch.Tensor:
    """pts [n,N,3], pose [n,7] (t,q) -> R(q)(scale*p)+t  (include/pfpp.h a19)"""
    _chk(pts, torch.float32, "pts"); _chk(pose, torch.float32, "pose")
    n, N, _ = pts.shape
    if pose.shape != (n, 7):
        raise ValueError("pose_apply: pose must be [n,7]")
    if scale is not None:
        _chk(scale, torch.float32, "scale")
        if scale.numel() != n:
            raise ValueError("pose_apply: scale must have n elements")
    out = torch.empty_like(pts)
    check(_lib.load().pfpp_pose_apply(_ptr(pts), _ptr(pose), _ptr(scale), _ptr(out), n, N, int(normalise), _stream()),
          "pfpp_pose_apply")
    return out


# --------------------------------------------------------------------------- PointNet++
def fps(xyz: torch.Tensor, npoint: int, start: Optional[torch.Tensor] = None):
    """xyz [F,N,3] -> (idx int32 [F,S], new_xyz [F,S,3])  (include/pfpp.h a2); start: int32 [F] first index
    (torch_cluster's random_start), default 0"""
    _chk(xyz, torch.float32, "xyz")
    F, N, three = xyz.shape
    if three != 3:
        raise ValueError("fps: xyz must be [F,N,3]")
    if not (1 <= npoint <= N):
        raise ValueError(f"fps: need 1 <= npoint <= N (npoint={npoint}, N={N})")
    idx = torch.empty((F, npoint), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((F, npoint, 3), dtype=torch.float32, device=xyz.device)
    if start is not None:
        _chk(start, torch.int32, "start")
        if start.numel() != F:
            raise ValueError("fps: start must have one index per fragment")
        check(_lib.load().pfpp_fps_start(_ptr(xyz), _ptr(idx), _ptr(new_xyz), F, N, npoint, _ptr(start), _stream()), "pfpp_fps_start")
        return idx, new_xyz
    check(_lib.load().pfpp_fps(_ptr(xyz), _ptr(idx), _ptr(new_xyz), F, N, npoint, _stream()), "pfpp_fps")
    return idx, new_xyz


def check_fps_ratio(npoint: int, N: int) -> None:
    """The reference does not ask torch_cluster.fps for `npoint` samples but for the RATIO npoint / N as a float64 tensor
    (utils/pn2_utils.py:131-134), and torch_cluster takes ceil(ratio * N) points per cloud.  For the
    reference's shapes the round trip is exact (256/1000, 128/256, 25/128; also N in {512, 1024, 2048}); for an N where it is
    not, the reference would sample a different number of points than `npoint` and everything downstream would change shape —
    refuse instead of silently diverging."""
    m = int(torch.ceil(torch.tensor(npoint / N, dtype=torch.float64) * N).item())
    if m != npoint:
        raise ValueError(f"fps: ceil(float64({npoint}/{N}) * {N}) = {m} != {npoint}: the reference's torch_cluster.fps(ratio=npoint/N) "
                         f"call would return {m} points per cloud for this N")


def ball_query(xyz: torch.Tensor, new_xyz: torch.Tensor, radius: float, nsample: int) -> torch.Tensor:
    """-> idx int32 [F,S,nsample]  (include/pfpp.h a3).  r^2 is rounded to fp32 exactly as the
    reference's `sqrdists > radius ** 2` comparison does (python double -> float32 scalar)."""
    _chk(xyz, torch.float32, "xyz"); _chk(new_xyz, torch.float32, "new_xyz")
    F, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if new_xyz.shape[0] != F:
        raise ValueError("ball_query: batch mismatch")
    r2 = torch.tensor(radius ** 2, dtype=torch.float64).to(torch.float32).item()
    idx = torch.empty((F, S, nsample), dtype=torch.int32, device=xyz.device)
    # grid.y carries the fragment index: chunk very large batches
    step = 65535
    lib = _lib.load()
    for f0 in range(0, F, step):
        f1 = min(F, f0 + step)
        check(lib.pfpp_ball_query(_ptr(xyz[f0:f1]), _ptr(new_xyz[f0:f1]), _ptr(idx[f0:f1]), f1 - f0, N, S,
                                  nsample, r2, _stream()), "pfpp_ball_query")
    return idx


def sample_levels(xyz: torch.Tensor, levels):
    """FPS + ball query of the three set-abstraction levels in one launch (pfpp_sample_levels).  levels = ((S, radius, nsample),) * 3
    -> [(fps_idx int32 [F,S], new_xyz [F,S,3], ball_idx int32 [F,S,nsample])] * 3, bit-identical to fps() + ball_query() per level"""
    _chk(xyz, torch.float32, "xyz")
    F, N, three = xyz.shape
    if three != 3 or len(levels) != 3:
        raise ValueError("sample_levels: xyz [F,N,3] and exactly three levels expected")
    arr = (_lib.SampleLevel * 3)()
    out = []
    prev = N
    for l, (S, radius, ns) in enumerate(levels):
        check_fps_ratio(S, prev)
        fi = torch.empty((F, S), dtype=torch.int32, device=xyz.device)
        nx = torch.empty((F, S, 3), dtype=torch.float32, device=xyz.device)
        bi = torch.empty((F, S, ns), dtype=torch.int32, device=xyz.device)
        arr[l].S, arr[l].nsample = S, ns
        arr[l].r2 = torch.tensor(radius ** 2, dtype=torch.float64).to(torch.float32).item()
        arr[l].fps_idx, arr[l].new_xyz, arr[l].ball_idx = fi.data_ptr(), nx.data_ptr(), bi.data_ptr()
        out.append((fi, nx, bi))
        prev = S
    check(_lib.load().pfpp_sample_levels(_ptr(xyz), F, N, arr, _stream()), "pfpp_sample_levels")
    return out


def sample_levels_supported(N: int, levels) -> bool:
    return len(levels) == 3 and N <= 2048 and levels[0][0] <= 256 and levels[1][0] <= 128 and all(ns <= 64 for _, _, ns in levels)


def group_gather(xyz: torch.Tensor, new_xyz: torch.Tensor, feats: Optional[torch.Tensor],
                 idx: torch.Tensor) -> torch.Tensor:
    """-> A operand [F*S*ns, D+4] = [feats | rel_xyz | 0]  (include/pfpp.h a4)"""
    _chk(xyz, torch.float32, "xyz"); _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    D = 0
    if feats is not None:
        _chk(feats, torch.float32, "feats")
        if feats.shape[:2] != (F, N):
            raise ValueError("group_gather: feats must be [F,N,D]")
        D = feats.shape[2]
    ldo = D + 4
    out = torch.empty((F * S * ns, ldo), dtype=torch.float32, device=xyz.device)
    check(_lib.load().pfpp_group_gather(_ptr(xyz), _ptr(new_xyz), _ptr(feats), _ptr(idx), _ptr(out), F, N, S, ns, D,
                                        ldo, _stream()), "pfpp_group_gather")
    return out


# --------------------------------------------------------------------------- GEMM
# "f16x3": split-f16 contraction (3 x v_mfma_f32_32x32x16_f16, fp32-grade error, needs |x| < 65504);
# "f32": exact fp32 MFMA.  PFPP_GEMM=f32 in the environment selects the exact path.
import os as _os

GEMM_MODE = _os.environ.get("PFPP_GEMM", "f16x3")
PRECISION = {"f32": 0, "f16x3": 1, "f16": 2}
# single-pass fp16 (PFPP_GEMM_F16: hi planes only, one MFMA per product) for every GEMM whose operands both arrive as split planes
# — the perf mode of BASELINE configs[4] (bench.py --mode stress), ~1e-3 relative error; the parity mode is f16x3
SINGLE_PASS = _os.environ.get("PFPP_GEMM_SINGLE_PASS", "0") == "1"

import contextlib as _ctx


import threading as _threading

_EXACT_DEPTH = 0
_ATTN_MODE_SET = _threading.local()      # what THIS thread last told the library (its attention mode is per calling thread)


def _sync_attention_mode() -> None:
    """the library's attention arithmetic for this thread's launches (pfpp_set_attention_mode) follows this module's state: exact fp32 inside
    exact_fp32(), single-pass fp16 while SINGLE_PASS is on (the perf mode of BASELINE configs[4]: plane GEMMs AND attention forward
    on one fp16 matrix instruction per product), the kernels' defaults otherwise.  Called by the attention wrappers; a foreign call
    only when the state changed."""
    want = 0 if _EXACT_DEPTH > 0 else (2 if (SINGLE_PASS and GEMM_MODE == "f16x3") else -1)
    if want != getattr(_ATTN_MODE_SET, "mode", None):
        check(_lib.load().pfpp_set_attention_mode(want), "pfpp_set_attention_mode")
        _ATTN_MODE_SET.mode = want


@_ctx.contextmanager
def exact_fp32():
    """run the enclosed calls with the exact fp32 MFMA GEMMs (PFPP_GEMM=f32) — the fallback the drop-in sampler loops take
    when a split-f16 run produced non-finite poses (an operand at or beyond the fp16 range, |v| >= 65504).  The attention
    forward kernels split raw q / k / v the same way, so they are switched to their exact-fp32 form for the duration as well
    (pfpp_set_attention_mode(0) through _sync_attention_mode)."""
    global GEMM_MODE, _EXACT_DEPTH
    prev, GEMM_MODE = GEMM_MODE, "f32"
    _EXACT_DEPTH += 1
    try:
        yield
    finally:
        GEMM_MODE = prev
        _EXACT_DEPTH -= 1


def f16x3_range_fallback(x: torch.Tensor) -> bool:
    """True when a result computed in the split-f16 mode is non-finite and should be recomputed under exact_fp32().
    Reads one flag back from the device: call it where the host synchronises anyway (end of a sampling loop)."""
    return GEMM_MODE == "f16x3" and not bool(torch.isfinite(x).all())


class SplitAct:
    """an activation travelling as split-f16 planes (hi, lo = x - hi), each fp16 [rows, C]: produced by the
    LayerNorm / attention kernels and GEMM epilogues, consumed as the A operand of the split-f16 GEMM with
    no conversion work (csrc/gemm_pl.hip)"""

    __slots__ = ("hi", "lo")

    def __init__(self, hi: torch.Tensor, lo: torch.Tensor):
        self.hi, self.lo = hi, lo

    @staticmethod
    def empty(rows: int, cols: int, device) -> "SplitAct":
        return SplitAct(torch.empty((rows, cols), dtype=torch.float16, device=device),
                        torch.empty((rows, cols), dtype=torch.float16, device=device))

    @property
    def shape(self):
        return self.hi.shape

    def float(self) -> torch.Tensor:
        return self.hi.float() + self.lo.float()


def split_mode() -> bool:
    """the activations our own kernels produce for the next GEMM (normalised rows, attention outputs, GEGLU products) as
    pre-split fp16 planes — same bytes as fp32.  The consuming GEMM is then the LDS-DMA staged plane kernel
    (csrc/gemm_pl.hip: no staging registers, no conversions, no ds_write in its K loop), bit-identical to the
    register-staged kernel and 1.3-1.6x faster on the 3,850-row shapes.  PFPP_SPLIT_ACT=0 restores fp32 activations."""
    return GEMM_MODE == "f16x3" and _os.environ.get("PFPP_SPLIT_ACT", "1") == "1"


# When set to a list, every pfpp_gemm launch appends (start_event, end_event, flops, kernel_name):
# bench.py uses it to time the dominant kernel with HIP events on the launch stream.
GEMM_TRACE = None


_SPLIT_WS = {}


def _split_workspace(device):
    """split-K workspace lent to pfpp_gemm: one (fp32 partials, int32 tickets) pair per (device, stream) — launches on one
    stream are ordered, launches on different streams must not share it"""
    key = (device.index, raw_stream_id(device.index))
    ws = _SPLIT_WS.get(key)
    if ws is None:
        ws = (torch.empty((16 * 1024 * 1024,), dtype=torch.float32, device=device),        # 64 MB
              torch.zeros((1024,), dtype=torch.int32, device=device))
        if STREAM_OVERRIDE is not None:
            torch.cuda.synchronize(device)        # the fill ran on torch's current stream, the first use is on the override stream
        _SPLIT_WS[key] = ws
    return ws


def gemm_kernel_name(M: int, N: int, act: str, pool: int, batch: int, w_kmajor: bool, f16x3: bool = False, K: int = 0,
                     presplit: bool = False, a_presplit: bool = False, fused_bn: bool = False, a_aff: bool = False) -> str:
    """the template instantiation pfpp_gemm dispatches to (mirror of the choice in csrc/gemm.hip with the
    default environment) — used to attribute HIP-event timings to the kernel names rocprofv3 reports"""
    wide = N > 64 or act == "geglu"
    if f16x3 and not w_kmajor:
        if a_presplit:
            if M >= 8192 and N >= 1024 and pool == 0:
                return "gemm_f16x3_apre_kernel<4, 2, 2, 4, false>"
            if M >= 8192 and pool != 32:
                return "gemm_f16x3_apre_kernel<2, 2, 4, 2, true>"
            t128 = ((M + 127) // 128) * ((N + 127) // 128) * batch
            if t128 < 1024 and act != "geglu" and pool == 0:
                return "gemm_f16x3_apre_kernel<2, 1, 2, 2, true>"
            return "gemm_f16x3_apre_kernel<2, 2, 2, 2, true>"
        if presplit and wide and M >= 8192 and N >= 1024 and pool == 0:
            return "gemm_f16x3_kernel<4, 2, true, 2, 4, false, false>"
        if presplit and wide and M >= 8192 and pool != 32:
            return f"gemm_f16x3_kernel<2, 2, true, 4, 2, true, {'true' if a_aff else 'false'}>"
        tiles128 = ((M + 127) // 128) * ((N + 127) // 128) * batch
        if presplit and not fused_bn and pool == 0 and ((M + 63) // 64) * ((N + 63) // 64) * batch <= 256:
            return "gemm_f16x3_deep_kernel<1, 2, 2, 2, 4>" if act == "geglu" else "gemm_f16x3_deep_kernel<1, 1, 2, 2, 8>"
        deep = not fused_bn and tiles128 < 2048
        if presplit and wide and tiles128 < 1024 and act != "geglu" and pool == 0:
            return f"gemm_f16x3_kernel<2, 1, true, 2, 2, {'true' if deep else 'false'}, false>"
        if presplit and deep:
            return f"gemm_f16x3_kernel<2, {2 if wide else 1}, true, 2, 2, true, false>"
        return f"gemm_f16x3_kernel<2, {2 if wide else 1}, {'true' if presplit else 'false'}, 2, 2, false, false>"
    return f"gemm_f32_mfma_kernel<2, {2 if wide else 1}, {'true' if w_kmajor else 'false'}>"


def gemm(A: torch.Tensor, W, *, M: int, N: int, K: int, lda: int, ldw: Optional[int] = None,
         out: Optional[torch.Tensor] = None, ldc: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
         shift: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         ldr: int = 0, act: str = "none", pool: int = 0, w_kmajor: bool = False,
         batch: int = 1, zdiv: int = 1, sA=(0, 0), sW=(0, 0), sC=(0, 0), sV=(0, 0), alpha: float = 1.0,
         a_off: int = 0, w_off: int = 0, c_off: int = 0, mode: Optional[str] = None,
         a_affine=None, stats: Optional[torch.Tensor] = None, c_min: Optional[torch.Tensor] = None,
         gather=None) -> torch.Tensor:
    """Raw pfpp_gemm call.  A/W/out are base tensors; *_off are element offsets into them
    (used to address q/k/v slices of a packed projection without copies).  W is an fp32 tensor or a
    packing.PW (fp32 + pre-split fp16 planes); `mode` overrides ops.GEMM_MODE for this call."""
    from .packing import PW

    mode = mode or GEMM_MODE
    f16x3 = mode == "f16x3" and not w_kmajor
    planes = None
    a_planes = None
    if isinstance(A, SplitAct):
        if not (f16x3 and isinstance(W, PW)):
            raise ValueError("gemm: a SplitAct operand needs the f16x3 mode and a packed (PW) weight")
        a_planes = (A.hi, A.lo)
        A = None
    c_planes = None
    if isinstance(out, SplitAct):
        c_planes = (out.hi, out.lo)
        if ldc is None:
            ldc = out.hi.shape[-1]
    if isinstance(W, PW):
        if f16x3:
            planes = (W.hi, W.lo)
            alpha = alpha / W.scale         # the planes stand for scale * w (packing.plane_scale): exact power of two
            if ldw is None:
                ldw = W.hi.shape[-1]
            elif ldw != W.hi.shape[-1]:
                raise ValueError("gemm: ldw does not match the pre-split planes")
        else:
            if ldw is None:
                ldw = W.f32.shape[-1]
        W = W.f32
    if ldw is None:
        ldw = W.shape[-1]
    if gather is not None:
        if A is not None:
            _chk(A, torch.float32, "A")
    elif a_planes is None:
        _chk(A, torch.float32, "A")
    else:
        _chk(a_planes[0], torch.float16, "A.hi"); _chk(a_planes[1], torch.float16, "A.lo")
    _chk(W, torch.float32, "W")
    dev_ = W.device
    n_out_cols = N // 2 if act == "geglu" else N
    if out is None:
        rows = M // pool if pool else M
        if ldc is None:
            ldc = n_out_cols
        out = torch.empty((batch, rows, ldc) if batch > 1 else (rows, ldc), dtype=torch.float32, device=dev_)
        if batch > 1 and sC == (0, 0):
            sC = (rows * ldc * zdiv, rows * ldc) if zdiv > 1 else (rows * ldc, 0)
    elif c_planes is None:
        _chk(out, torch.float32, "out")
        if ldc is None:
            ldc = out.shape[-1]
    for t, nm in ((bias, "bias"), (scale, "scale"), (shift, "shift"), (residual, "residual")):
        if t is not None:
            _chk(t, torch.float32, nm)
    es = 4
    args = GemmArgs()
    args.A = 0 if A is None else A.data_ptr() + a_off * es
    args.a_hi = 0 if a_planes is None else a_planes[0].data_ptr() + a_off * 2
    args.a_lo = 0 if a_planes is None else a_planes[1].data_ptr() + a_off * 2
    args.W = W.data_ptr() + w_off * es
    args.w_hi = 0 if planes is None else planes[0].data_ptr() + w_off * 2
    args.w_lo = 0 if planes is None else planes[1].data_ptr() + w_off * 2
    args.precision = PRECISION["f16" if (f16x3 and SINGLE_PASS and a_planes is not None and planes is not None) else
                               "f16x3" if f16x3 else "f32"]
    args.C = 0 if c_planes is not None else out.data_ptr() + c_off * es
    args.c_hi = 0 if c_planes is None else c_planes[0].data_ptr() + c_off * 2
    args.c_lo = 0 if c_planes is None else c_planes[1].data_ptr() + c_off * 2
    args.bias = 0 if bias is None else bias.data_ptr()
    args.scale = 0 if scale is None else scale.data_ptr()
    args.shift = 0 if shift is None else shift.data_ptr()
    args.residual = 0 if residual is None else residual.data_ptr() + c_off * es
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldw, args.ldc, args.ldr = lda, ldw, ldc, ldr
    args.w_kmajor = int(w_kmajor)
    args.act = ACT[act]
    args.pool = pool
    args.batch, args.zdiv = batch, zdiv
    args.sA0, args.sA1 = sA
    args.sW0, args.sW1 = sW
    args.sC0, args.sC1 = sC
    args.sV0, args.sV1 = sV
    args.alpha = alpha
    if a_affine is not None:      # train-mode BatchNorm fusion (pfpp_gemm_args.a_mul / a_add / stats / c_min)
        _chk(a_affine[0], torch.float32, "a_mul"); _chk(a_affine[1], torch.float32, "a_add")
        if a_affine[0].numel() < K or a_affine[1].numel() < K:
            raise ValueError("gemm: a_affine vectors shorter than K")
        args.a_mul, args.a_add = a_affine[0].data_ptr(), a_affine[1].data_ptr()
    if stats is not None:
        _chk(stats, torch.float64, "stats")
        if stats.dim() != 3 or stats.shape[1] != 2 or stats.shape[2] != N:
            raise ValueError("gemm: stats must be float64 [copies, 2, N]")
        args.stats, args.stats_copies = stats.data_ptr(), stats.shape[0]
    if c_min is not None:
        _chk(c_min, torch.float32, "c_min")
        args.c_min = c_min.data_ptr()
    if gather is not None:        # fused grouping: (ball indices [F,S,ns] int32, xyz [F,N,3], centroids [F,S,3])
        g_idx, g_xyz, g_ctr = gather
        _chk(g_idx, torch.int32, "gather idx"); _chk(g_xyz, torch.float32, "gather xyz"); _chk(g_ctr, torch.float32, "gather centroids")
        Fg, Sg, nsg = g_idx.shape
        if g_xyz.shape[0] != Fg or g_ctr.shape[:2] != (Fg, Sg) or M != Fg * Sg * nsg:
            raise ValueError("gemm: gather tables do not match M = F*S*ns")
        args.gather_idx, args.gather_xyz, args.gather_ctr = g_idx.data_ptr(), g_xyz.data_ptr(), g_ctr.data_ptr()
        args.gather_N, args.gather_S, args.gather_ns = g_xyz.shape[1], Sg, nsg
    ws = _split_workspace(dev_)
    if ws is not None:
        args.split_ws, args.split_ws_bytes = ws[0].data_ptr(), ws[0].numel() * 4
        args.split_cnt, args.split_cnt_len = ws[1].data_ptr(), ws[1].numel()
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_gemm(C.byref(args), _stream()), "pfpp_gemm")
        e1.record()
        pl_name = _lib.load().pfpp_last_gemm_kernel().decode()        # set when the call went to the plane kernel (gemm_pl.hip)
        GEMM_TRACE.append((e0, e1, 2.0 * M * N * K * batch,
                           pl_name or gemm_kernel_name(M, N, act, pool, batch, w_kmajor, f16x3, K, planes is not None, a_planes is not None,
                                                       a_affine is not None or stats is not None or c_min is not None, a_affine is not None),
                           (M, N, K, batch, act, pool)))
        return out
    check(_lib.load().pfpp_gemm(C.byref(args), _stream()), "pfpp_gemm")
    return out


def linear(x: torch.Tensor, w, bias: Optional[torch.Tensor] = None, *, act: str = "none",
           scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, pool: int = 0, K: Optional[int] = None,
           mode: Optional[str] = None, out=None, a_affine=None, stats=None, c_min=None):
    """y = epilogue(x @ w^T): x [M, ldx] (first K columns used); w = packing.PW or an fp32 tensor
    [N, ldw] with ldw % 4 == 0."""
    from .packing import PW

    M, ldx = x.shape          # works for torch.Tensor and SplitAct alike
    if isinstance(w, PW):
        N = w.N
        if K is None:
            K = w.K
    else:
        N, ldw = w.shape
        if K is None:
            K = min(ldx, ldw)
    return gemm(x, w, M=M, N=N, K=K, lda=ldx, bias=bias, scale=scale, shift=shift,
                residual=residual, ldr=(residual.shape[-1] if residual is not None else 0), act=act, pool=pool,
                mode=mode, out=out, a_affine=a_affine, stats=stats, c_min=c_min)


def grouped_linear(xyz: torch.Tensor, new_xyz: torch.Tensor, feats: Optional[torch.Tensor], idx: torch.Tensor, w, bias=None, *,
                   act: str = "none", scale=None, shift=None, pool: int = 0, stats=None, out=None) -> torch.Tensor:
    """first 1x1 convolution of a set-abstraction level applied to the grouped neighbourhoods WITHOUT materialising them:
    linear(group_gather(xyz, new_xyz, feats, idx), w, ...) with the gather done by the GEMM's A loader
    (pfpp_gemm_args.gather_*; f16x3 mode, w = packing.PW)."""
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    D = 0 if feats is None else feats.shape[2]
    if feats is not None and (feats.shape[:2] != (F, N) or not feats.is_contiguous()):
        raise ValueError("grouped_linear: feats must be a contiguous [F,N,D]")
    return gemm(None if feats is None else feats.view(F * N, D), w, M=F * S * ns, N=w.N, K=D + 4, lda=D, bias=bias, scale=scale,
                shift=shift, act=act, pool=pool, stats=stats, gather=(idx, xyz, new_xyz), mode="f16x3", out=out)


_IDENT_IDX = {}


def sa_first_table(xyz: torch.Tensor, feats: torch.Tensor, w, bias: torch.Tensor) -> torch.Tensor:
    """U [F*N, C1] = the first 1x1 convolution of a set-abstraction level applied PER POINT: [feats | xyz] . W1^T + b1 — the fused
    grouping of grouped_linear with the identity index and zero centroids.  conv1 is linear, so its value on the grouped row
    (neighbourhood s, point p) is U[p] - W1_xyz . centroid_s (pfpp_sa_train_args.u_in)"""
    F, N, _ = xyz.shape
    key = (F, N, xyz.device.index)
    t = _IDENT_IDX.get(key)
    if t is None:
        idx = torch.arange(N, dtype=torch.int32, device=xyz.device).view(1, 1, N).expand(F, 1, N).contiguous()
        t = _IDENT_IDX[key] = (idx, torch.zeros((F, 1, 3), dtype=torch.float32, device=xyz.device))
    return grouped_linear(xyz, t[1], feats, t[0], w, bias=bias)


def sa_mlp3_fused(xyz: torch.Tensor, new_xyz: torch.Tensor, idx: torch.Tensor, w0, w1, w2, s0, t0, s1, t1, s2, t2) -> torch.Tensor:
    """grouping + three folded [conv, BN, ReLU] + max over nsample of a feature-less set-abstraction level in one kernel
    (pfpp_sa_mlp3_fused); w* = packing.PW, s*/t* the folded BatchNorm scale / shift -> [F*S, C3]"""
    _chk(xyz, torch.float32, "xyz"); _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    for t, nm in ((s0, "s0"), (t0, "t0"), (s1, "s1"), (t1, "t1"), (s2, "s2"), (t2, "t2")):
        _chk(t, torch.float32, nm)
    if w0.hi.shape != (w0.N, 8) or w1.hi.shape != (w1.N, w0.N) or w2.hi.shape != (w2.N, w1.N):
        raise ValueError("sa_mlp3_fused: weight planes do not chain ([C1,8], [C2,C1], [C3,C2])")
    if (w0.scale, w1.scale, w2.scale) != (1.0, 1.0, 1.0):
        raise ValueError("sa_mlp3_fused reads the planes as they are: pack these weights with PW(w, prescale=False)")
    out = torch.empty((F * S, w2.N), dtype=torch.float32, device=xyz.device)
    check(_lib.load().pfpp_sa_mlp3_fused(_ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(w0.hi), _ptr(w0.lo), _ptr(w1.hi), _ptr(w1.lo),
                                         _ptr(w2.hi), _ptr(w2.lo), _ptr(s0), _ptr(t0), _ptr(s1), _ptr(t1), _ptr(s2), _ptr(t2),
                                         _ptr(out), F, N, S, ns, w0.N, w1.N, w2.N, _stream()), "pfpp_sa_mlp3_fused")
    return out


# persistent-kernel grid cap: the number of CUs the current stream may use (None = all 256).  The train-mode chain kernels run one
# workgroup per CU with a static stride over the neighbourhoods; on a CU-masked stream (pfpp_hip.train.FeaturePipeline) a 256-wide
# grid would run as two unequal rounds
PERSISTENT_WGS: Optional[int] = None


def sa_pad_schedule(idx: torch.Tensor) -> torch.Tensor:
    """pfpp_sa_pad_schedule: idx [F, S, 64] ball-query indices -> int32 [3 F*S + 1]: the neighbourhoods with more than 32 live slots,
    then the others (each class ascending), the size of the first class at [F*S], then the live-slot counts by neighbourhood and in
    schedule order"""
    _chk(idx, torch.int32, "idx")
    F, S, ns = idx.shape
    sched = torch.empty((3 * F * S + 1,), dtype=torch.int32, device=idx.device)
    check(_lib.load().pfpp_sa_pad_schedule(_ptr(idx), F * S, ns, _ptr(sched), _stream()), "pfpp_sa_pad_schedule")
    return sched


def sa_train_stage(stage: int, xyz: torch.Tensor, new_xyz: torch.Tensor, feats: Optional[torch.Tensor], idx: torch.Tensor, ws, biases,
                   affines, stats: torch.Tensor, y_out: Optional[torch.Tensor] = None, out_max: Optional[torch.Tensor] = None,
                   out_min: Optional[torch.Tensor] = None, y_in: Optional[torch.Tensor] = None,
                   u_in: Optional[torch.Tensor] = None, sched: Optional[torch.Tensor] = None) -> None:
    """one stage of the train-mode set-abstraction chain (pfpp_sa_train_stage): batch statistics of layer `stage` by recomputation
    of layers 1..stage-1 with their finalised BatchNorm affines; ws / biases = packing.PW / conv bias per layer (at least `stage` of
    them), affines = [(a_mul, a_add)] of the finalised layers (stage - 1 of them), stats = train_ops.bn_stats_buffer(C_stage).
    u_in (levels with features, stages 1 and 2): the first convolution applied per point (sa_first_table) — stage 1 then only takes
    statistics, stage 2 gathers its rows from the table and writes y_out [F*S*ns, C2].
    sched (64-neighbour levels): sa_pad_schedule(idx) — neighbourhoods with at most 32 points in range are taken as one half; every stage
    of the level gets it or none does (the raw rows of a skipped half are neither written nor read)"""
    _chk(xyz, torch.float32, "xyz"); _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    if stats.dtype != torch.float64 or not stats.is_cuda or not stats.is_contiguous():
        raise ValueError("sa_train_stage: stats must be a contiguous float64 CUDA tensor [copies, 2, C]")
    if len(ws) < stage or len(biases) < stage or len(affines) < stage - 1:
        raise ValueError("sa_train_stage: weights / biases for layers 1..stage and affines for layers 1..stage-1 are needed")
    if u_in is not None and stage <= 2:
        if feats is None or (stage == 2 and y_out is None):
            raise ValueError("sa_train_stage: the per-point table belongs to a level with features; stage 2 writes y_out")
    elif feats is not None and feats.shape[-1] == 256:
        if stage > 1 and y_in is None:
            raise ValueError("sa_train_stage: stages 2 and 3 of the wide level read the previous layer's raw rows (y_in)")
    elif feats is not None and stage == 3 and y_out is None:
        raise ValueError("sa_train_stage: stage 3 of a level with input features reads the raw rows stage 2 wrote (y_out)")
    a = _lib.SaTrainArgs()
    a.xyz, a.new_xyz, a.idx = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr()
    D = 0
    if feats is not None:
        _chk(feats, torch.float32, "feats")
        D = feats.shape[2]
        if feats.shape[:2] != (F, N):
            raise ValueError("sa_train_stage: feats [F, N, D] expected")
        a.feats = feats.data_ptr()
    widths = [0, 0, 0]
    k_in = D + 8 if feats is not None else 8
    for i in range(stage):
        w = ws[i]
        if w.scale != 1.0:
            raise ValueError("sa_train_stage reads the planes as they are: pack these weights with PW(w, prescale=False)")
        if w.hi.shape != (w.N, k_in):
            raise ValueError(f"sa_train_stage: layer {i + 1} weight planes {tuple(w.hi.shape)} do not chain ({w.N}, {k_in})")
        _chk(biases[i], torch.float32, "bias")
        a.w_hi[i], a.w_lo[i], a.bias[i] = w.hi.data_ptr(), w.lo.data_ptr(), biases[i].data_ptr()
        widths[i] = k_in = w.N
    for i in range(stage - 1):
        am, aa = affines[i]
        _chk(am, torch.float32, "a_mul"); _chk(aa, torch.float32, "a_add")
        a.a_mul[i], a.a_add[i] = am.data_ptr(), aa.data_ptr()
    if stats.shape[1:] != (2, widths[stage - 1]):
        raise ValueError("sa_train_stage: stats [copies, 2, C_stage] expected")
    a.stats, a.stats_copies = stats.data_ptr(), stats.shape[0]
    wide = feats is not None and D == 256          # sa3: one layer per stage, y_out = this layer's raw rows, y_in = the previous layer's
    utab = u_in is not None and stage <= 2
    if utab:
        _chk(u_in, torch.float32, "u_in")
        if u_in.shape != (F * N, widths[0]):
            raise ValueError("sa_train_stage: u_in must be [F*N, C1]")
        a.u_in = u_in.data_ptr()
    if y_in is not None:
        _chk(y_in, torch.float32, "y_in")
        if not wide or y_in.shape != (F * S * ns, widths[stage - 2]):
            raise ValueError("sa_train_stage: y_in is the wide level's previous-layer rows [F*S*ns, C]")
        a.y_in = y_in.data_ptr()
    for t, nm, rows in ((y_out, "y_out", F * S * ns), (out_max, "out_max", F * S), (out_min, "out_min", F * S)):
        if t is not None:
            _chk(t, torch.float32, nm)
            cols = (widths[stage - 1] if wide else widths[1]) if nm == "y_out" else widths[stage - 1]
            if t.shape != (rows, cols):
                raise ValueError(f"sa_train_stage: {nm} must be [{rows}, {cols}]")
            setattr(a, nm, t.data_ptr())
    # the ABI fixes the supported widths; layers beyond `stage` are reported with the level's known widths
    full = (64, 64, 128) if feats is None else ((256, 256, 512) if D == 256 else (128, 128, 256))
    a.F, a.N, a.S, a.ns, a.D = F, N, S, ns, D
    a.C1, a.C2, a.C3 = (widths[0] or full[0]), (widths[1] or full[1]), (widths[2] or full[2])
    a.stage = stage
    a.max_workgroups = int(PERSISTENT_WGS or 0)
    if sched is not None:
        _chk(sched, torch.int32, "sched")
        if sched.shape != (3 * F * S + 1,):
            raise ValueError("sa_train_stage: sched must be sa_pad_schedule(idx): int32 [3 F*S + 1]")
        a.sched = sched.data_ptr()
    if GEMM_TRACE is not None:            # bench.py: HIP events around the launch; FLOPs = the layers this launch actually computes
        rows = F * S * ns
        kin = [D + 3 if feats is not None else 3, full[0], full[1]]
        layers = ([] if stage == 1 else [1]) if utab else [stage - 1] if wide else ([2] if (feats is not None and stage == 3) else range(stage))
        flops = sum(2.0 * rows * kin[i] * full[i] for i in layers)
        name = ((f"sa_first_stats_kernel<{D}>" if stage == 1 else f"sa_wide_train_kernel<{D}, 2, true>") if utab else
                f"sa1_train_kernel<64, 64, 128, {stage}>" if feats is None else f"sa_wide_train_kernel<256, {stage}>" if wide else
                ("sa_rows_train_kernel<128, 256>" if _os.environ.get("PFPP_SA_ROWS8") == "0" else "sa_rows8_train_kernel<128, 256>") if stage == 3
                else f"sa2_train_kernel<128, 128, 128, {stage}>")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_sa_train_stage(C.byref(a), _stream()), "pfpp_sa_train_stage")
        e1.record()
        GEMM_TRACE.append((e0, e1, flops, name, (rows, full[stage - 1], kin[stage - 1], 1, f"chain{stage}", ns)))
        return
    check(_lib.load().pfpp_sa_train_stage(C.byref(a), _stream()), "pfpp_sa_train_stage")


def sa_mlp2_fused(xyz: torch.Tensor, new_xyz: torch.Tensor, feats: torch.Tensor, idx: torch.Tensor, w0, w1, s0, t0, s1, t1,
                  as_planes: bool = False):
    """grouping + the first two folded [conv, BN, ReLU] of a set-abstraction level with input features in one kernel
    (pfpp_sa_mlp2_fused) -> [F*S*ns, C2], the input of the level's third convolution; as_planes: a SplitAct (split-f16 planes)"""
    _chk(xyz, torch.float32, "xyz"); _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx"); _chk(feats, torch.float32, "feats")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    D = feats.shape[2]
    for t, nm in ((s0, "s0"), (t0, "t0"), (s1, "s1"), (t1, "t1")):
        _chk(t, torch.float32, nm)
    if feats.shape[:2] != (F, N) or w0.hi.shape != (w0.N, D + 8) or w1.hi.shape != (w1.N, w0.N):
        raise ValueError("sa_mlp2_fused: shapes do not chain (feats [F,N,D], w0 planes [C1,D+8], w1 planes [C2,C1])")
    if (w0.scale, w1.scale) != (1.0, 1.0):
        raise ValueError("sa_mlp2_fused reads the planes as they are: pack these weights with PW(w, prescale=False)")
    if as_planes:
        sp = SplitAct.empty(F * S * ns, w1.N, xyz.device)
        pc = _lib.PlanesC(sp.hi.data_ptr(), sp.lo.data_ptr(), 1.0)
        check(_lib.load().pfpp_sa_mlp2_fused_p(_ptr(feats), _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(w0.hi), _ptr(w0.lo), _ptr(w1.hi),
                                               _ptr(w1.lo), _ptr(s0), _ptr(t0), _ptr(s1), _ptr(t1), None, C.byref(pc), F, N, S, ns, D, w0.N,
                                               w1.N, _stream()), "pfpp_sa_mlp2_fused_p")
        return sp
    out = torch.empty((F * S * ns, w1.N), dtype=torch.float32, device=xyz.device)
    check(_lib.load().pfpp_sa_mlp2_fused(_ptr(feats), _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(w0.hi), _ptr(w0.lo), _ptr(w1.hi), _ptr(w1.lo),
                                         _ptr(s0), _ptr(t0), _ptr(s1), _ptr(t1), _ptr(out), F, N, S, ns, D, w0.N, w1.N, _stream()),
          "pfpp_sa_mlp2_fused")
    return out


def sa_mlp2_table(xyz: torch.Tensor, new_xyz: torch.Tensor, feats: torch.Tensor, idx: torch.Tensor, w0, w1, s0, t0, s1, t1):
    """sa_mlp2_fused(as_planes=True) with the first convolution taken per point (pfpp_sa_mlp2_table_p): u = sa_first_table without bias,
    then one launch that gathers u[idx], subtracts W_xyz . centroid inside the folded affine and runs the second layer -> SplitAct"""
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx"); _chk(feats, torch.float32, "feats")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    D = feats.shape[2]
    for t, nm in ((s0, "s0"), (t0, "t0"), (s1, "s1"), (t1, "t1")):
        _chk(t, torch.float32, nm)
    if feats.shape[:2] != (F, N) or w0.hi.shape != (w0.N, D + 8) or w1.hi.shape != (w1.N, w0.N):
        raise ValueError("sa_mlp2_table: shapes do not chain (feats [F,N,D], w0 planes [C1,D+8], w1 planes [C2,C1])")
    if (w0.scale, w1.scale) != (1.0, 1.0):
        raise ValueError("sa_mlp2_table reads the planes as they are: pack these weights with PW(w, prescale=False)")
    u = sa_first_table(xyz, feats, w0, None)
    sp = SplitAct.empty(F * S * ns, w1.N, xyz.device)
    pc = _lib.PlanesC(sp.hi.data_ptr(), sp.lo.data_ptr(), 1.0)
    check(_lib.load().pfpp_sa_mlp2_table_p(_ptr(u), _ptr(new_xyz), _ptr(idx), _ptr(w0.hi), _ptr(w0.lo), _ptr(w1.hi), _ptr(w1.lo), _ptr(s0),
                                           _ptr(t0), _ptr(s1), _ptr(t1), C.byref(pc), F, N, S, ns, D, w0.N, w1.N,
                                           int(PERSISTENT_WGS or 0), _stream()), "pfpp_sa_mlp2_table_p")
    return sp


def sa_table_planes(xyz: torch.Tensor, new_xyz: torch.Tensor, feats: torch.Tensor, idx: torch.Tensor, w0, s0, t0):
    """first folded [conv, BN, ReLU] of a level with features on every grouped row, from the per-point table (pfpp_sa_table_planes):
    grouped_linear(..., scale=s0, shift=t0, act="relu") as a SplitAct without the grouped matrix work"""
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(idx, torch.int32, "idx"); _chk(feats, torch.float32, "feats")
    _chk(s0, torch.float32, "s0"); _chk(t0, torch.float32, "t0")
    F, N, _ = xyz.shape
    _, S, ns = idx.shape
    D = feats.shape[2]
    if feats.shape[:2] != (F, N) or w0.hi.shape != (w0.N, D + 8) or w0.scale != 1.0:
        raise ValueError("sa_table_planes: feats [F,N,D], w0 planes [C1, D+8] packed with PW(w, prescale=False)")
    u = sa_first_table(xyz, feats, w0, None)
    sp = SplitAct.empty(F * S * ns, w0.N, xyz.device)
    pc = _lib.PlanesC(sp.hi.data_ptr(), sp.lo.data_ptr(), 1.0)
    check(_lib.load().pfpp_sa_table_planes(_ptr(u), _ptr(new_xyz), _ptr(idx), _ptr(w0.hi), _ptr(w0.lo), _ptr(s0), _ptr(t0), C.byref(pc),
                                           F, N, S, ns, D, w0.N, _stream()), "pfpp_sa_table_planes")
    return sp


# --------------------------------------------------------------------------- VQ
def vq_encode(z_e: torch.Tensor, codebook: torch.Tensor, slot: torch.Tensor, n_slots: int,
              z_q: Optional[torch.Tensor] = None, return_codes: bool = False):
    """z_e [F, L, C] with C % 16 == 0 -> z_q scattered into zeros [n_slots, L, C]  (include/pfpp.h a7/a8)"""
    _chk(z_e, torch.float32, "z_e"); _chk(codebook, torch.float32, "codebook"); _chk(slot, torch.int32, "slot")
    F, L, Cc = z_e.shape
    n_codes, dim = codebook.shape
    if Cc % dim != 0:
        raise ValueError("vq_encode: latent width must be a multiple of the code width")
    sub = L * (Cc // dim)
    if z_q is None:
        z_q = torch.zeros((n_slots, L, Cc), dtype=torch.float32, device=z_e.device)
    codes = torch.empty((F, sub), dtype=torch.int32, device=z_e.device) if return_codes else None
    check(_lib.load().pfpp_vq_encode(_ptr(z_e), _ptr(codebook), _ptr(slot), _ptr(z_q), _ptr(codes), F, sub, dim,
                                     n_codes, _stream()), "pfpp_vq_encode")
    return (z_q, codes) if return_codes else z_q


def scatter_rows(src: torch.Tensor, slot: torch.Tensor, n_slots: int,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[slot[f]] = src[f]; `out` defaults to zeros [n_slots, ...]  (denoiser.py:72-76)"""
    _chk(src, torch.float32, "src"); _chk(slot, torch.int32, "slot")
    F = src.shape[0]
    row = src[0].numel() if F else 0
    if out is None:
        out = torch.zeros((n_slots, *src.shape[1:]), dtype=torch.float32, device=src.device)
    else:
        _chk(out, torch.float32, "out")
    if F:
        check(_lib.load().pfpp_scatter_rows(_ptr(src), _ptr(slot), _ptr(out), F, row, _stream()), "pfpp_scatter_rows")
    return out


# --------------------------------------------------------------------------- transformer pieces
def token_features(latent, xyz, scale, x, slot=None):
    """latent [n,L,64], xyz [n,L,3], scale [n], x [n,7] -> shape_feat [n*L,148], pose_feat [n,148].
    slot (int32 [F]): the inputs are the PADDED tensors (n = number of slots) and listed fragment f is read from slot[f] -> [F*L,148],
    [F,148] (the valid-fragment gather inside the kernel)"""
    for t, nm in ((latent, "latent"), (xyz, "xyz"), (scale, "scale"), (x, "x")):
        _chk(t, torch.float32, nm)
    n, L, c = latent.shape
    if c != 64 or xyz.shape != (n, L, 3) or scale.numel() != n or x.shape != (n, 7):
        raise ValueError("token_features: shape mismatch")
    if slot is not None:
        _chk(slot, torch.int32, "slot")
        n = slot.numel()
    sf = torch.empty((n * L, 148), dtype=torch.float32, device=latent.device)
    pf = torch.empty((n, 148), dtype=torch.float32, device=latent.device)
    if slot is not None:
        check(_lib.load().pfpp_token_features_slots(_ptr(latent), _ptr(xyz), _ptr(scale), _ptr(x), _ptr(slot), _ptr(sf), _ptr(pf), n, L,
                                                    _stream()), "pfpp_token_features_slots")
        return sf, pf
    check(_lib.load().pfpp_token_features(_ptr(latent), _ptr(xyz), _ptr(scale), _ptr(x), _ptr(sf), _ptr(pf), n, L,
                                          _stream()), "pfpp_token_features")
    return sf, pf


def token_combine(shape_emb, x_emb, ref_emb, ref_part_u8, pe, B, P, L):
    for t, nm in ((shape_emb, "shape_emb"), (x_emb, "x_emb"), (ref_emb, "ref_emb"), (pe, "pe")):
        _chk(t, torch.float32, nm)
    _chk(ref_part_u8, torch.uint8, "ref_part")
    Cc = shape_emb.shape[-1]
    tok = torch.empty((B * P * L, Cc), dtype=torch.float32, device=shape_emb.device)
    check(_lib.load().pfpp_token_combine(_ptr(shape_emb), _ptr(x_emb), _ptr(ref_emb), _ptr(ref_part_u8), _ptr(pe),
                                         _ptr(tok), B, P, L, Cc, _stream()), "pfpp_token_combine")
    return tok


def token_combine_list(shape_emb, x_emb, ref_emb, ref_u8, pe, frag_pos, L, slot=None):
    """token assembly for a compacted fragment list: frag_pos[f] = index of the fragment inside its puzzle; with slot, ref_u8 is the
    padded [n_slots] flag array and fragment f reads ref_u8[slot[f]]"""
    for t, nm in ((shape_emb, "shape_emb"), (x_emb, "x_emb"), (ref_emb, "ref_emb"), (pe, "pe")):
        _chk(t, torch.float32, nm)
    _chk(ref_u8, torch.uint8, "ref_part"); _chk(frag_pos, torch.int32, "frag_pos")
    n, Cc = x_emb.shape
    tok = torch.empty((n * L, Cc), dtype=torch.float32, device=shape_emb.device)
    if slot is not None:
        _chk(slot, torch.int32, "slot")
        check(_lib.load().pfpp_token_combine_slots(_ptr(shape_emb), _ptr(x_emb), _ptr(ref_emb), _ptr(ref_u8), _ptr(pe), _ptr(frag_pos),
                                                   _ptr(slot), _ptr(tok), n, L, Cc, _stream()), "pfpp_token_combine_slots")
        return tok
    check(_lib.load().pfpp_token_combine_list(_ptr(shape_emb), _ptr(x_emb), _ptr(ref_emb), _ptr(ref_u8), _ptr(pe),
                                              _ptr(frag_pos), _ptr(tok), n, L, Cc, _stream()), "pfpp_token_combine_list")
    return tok


def layernorm_grouped(x: torch.Tensor, mod: torch.Tensor, group_batch: torch.Tensor, group_rows: int,
                      eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AdaLN over a compacted token list: the batch (row of `mod`) of token r is group_batch[r // group_rows]"""
    _chk(x, torch.float32, "x"); _chk(mod, torch.float32, "mod"); _chk(group_batch, torch.int32, "group_batch")
    rows, Cc = x.shape
    if isinstance(out, SplitAct):
        check(_lib.load().pfpp_layernorm_grouped_split(_ptr(x), _ptr(out.hi), _ptr(out.lo), _ptr(mod), mod.shape[-1], _ptr(group_batch),
                                                       group_rows, rows, Cc, eps, _stream()), "pfpp_layernorm_grouped_split")
        return out
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().pfpp_layernorm_grouped(_ptr(x), _ptr(out), _ptr(mod), mod.shape[-1], _ptr(group_batch), group_rows,
                                             rows, Cc, eps, _stream()), "pfpp_layernorm_grouped")
    return out


def silu_embed(tables: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """tables [n_tab, n_emb, C], t int64 [B] -> silu(tables[:, t]) [n_tab, B, C]"""
    _chk(tables, torch.float32, "tables"); _chk(t, torch.int64, "t")
    n_tab, n_emb, Cc = tables.shape
    B = t.numel()
    out = torch.empty((n_tab, B, Cc), dtype=torch.float32, device=tables.device)
    check(_lib.load().pfpp_silu_embed(_ptr(tables), _ptr(t), _ptr(out), n_tab, n_emb, B, Cc, _stream()),
          "pfpp_silu_embed")
    return out


def layernorm(x: torch.Tensor, *, mod: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
              beta: Optional[torch.Tensor] = None, rows_per_batch: int = 1, eps: float = 1e-5,
              out=None):
    """x [rows, C]; mod [B, 2C] (AdaLN scale|shift) or gamma/beta [C]; `out` may be a SplitAct (the
    normalised rows are then written as split-f16 planes for the next GEMM)"""
    _chk(x, torch.float32, "x")
    rows, Cc = x.shape
    ld_mod = 0
    if mod is not None:
        _chk(mod, torch.float32, "mod")
        ld_mod = mod.shape[-1]
        if ld_mod != 2 * Cc:
            raise ValueError("layernorm: mod must be [B, 2C]")
    if gamma is not None:
        _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    if isinstance(out, SplitAct):
        check(_lib.load().pfpp_layernorm_split(_ptr(x), _ptr(out.hi), _ptr(out.lo), _ptr(mod), ld_mod, _ptr(gamma),
                                               _ptr(beta), rows, Cc, rows_per_batch, eps, _stream()),
              "pfpp_layernorm_split")
        return out
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().pfpp_layernorm(_ptr(x), _ptr(out), _ptr(mod), ld_mod, _ptr(gamma), _ptr(beta), rows, Cc,
                                     rows_per_batch, eps, _stream()), "pfpp_layernorm")
    return out


def embed_tokens_small(latent: torch.Tensor, xyz: torch.Tensor, scale: torch.Tensor, x: torch.Tensor, slot: Optional[torch.Tensor], w_cat,
                       bias: torch.Tensor, ref_emb: torch.Tensor, ref_u8: torch.Tensor, pe: torch.Tensor, frag_pos: torch.Tensor, n: int,
                       L: int) -> torch.Tensor:
    """the Denoiser's token embedding in one launch for few tokens (pfpp_embed_tokens_small): -> tok [n * L, C].
    w_cat = packing.PW of [W_shape | W_param | 0] [C, 320], bias = shape bias + param bias (denoiser.pack_denoiser builds both)"""
    from ._lib import PwC

    for t_, nm in ((latent, "latent"), (xyz, "xyz"), (scale, "scale"), (x, "x"), (bias, "bias"), (ref_emb, "ref_emb"), (pe, "pe")):
        _chk(t_, torch.float32, nm)
    _chk(ref_u8, torch.uint8, "ref_part"); _chk(frag_pos, torch.int32, "frag_pos")
    if slot is not None:
        _chk(slot, torch.int32, "slot")
    Cc = bias.numel()
    fh, fl = w_cat.frag()
    pw = PwC(w_cat.f32.data_ptr(), w_cat.hi.data_ptr(), w_cat.lo.data_ptr(), w_cat.scale, w_cat.hi.shape[-1], fh.data_ptr(), fl.data_ptr())
    tok = torch.empty((n * L, Cc), dtype=torch.float32, device=latent.device)
    check(_lib.load().pfpp_embed_tokens_small(_ptr(latent), _ptr(xyz), _ptr(scale), _ptr(x), _ptr(slot), C.byref(pw), _ptr(bias), _ptr(ref_emb),
                                              _ptr(ref_u8), _ptr(pe), _ptr(frag_pos), _ptr(tok), n, L, Cc, _stream()),
          "pfpp_embed_tokens_small")
    return tok


def gemm_small(a, w, *, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a . w^T (+ bias) (+ residual) for few rows (pfpp_gemm_small): a = SplitAct planes [M, K], w = packing.PW [N, K] (its
    fragment-blocked planes are built on first use); out may be the residual tensor"""
    from ._lib import PlanesC, PwC

    M, K = a.hi.shape
    if K != w.K:
        raise ValueError(f"gemm_small: A has K = {K}, the weight {w.K}")
    fh, fl = w.frag()
    pw = PwC(w.f32.data_ptr(), w.hi.data_ptr(), w.lo.data_ptr(), w.scale, w.hi.shape[-1], fh.data_ptr(), fl.data_ptr())
    for t_, nm in ((bias, "bias"), (residual, "residual"), (out, "out")):
        if t_ is not None:
            _chk(t_, torch.float32, nm)
    if out is None:
        out = torch.empty((M, w.N), dtype=torch.float32, device=a.hi.device)
    ap = PlanesC(a.hi.data_ptr(), a.lo.data_ptr(), 1.0)
    check(_lib.load().pfpp_gemm_small(C.byref(ap), a.hi.stride(0), C.byref(pw), _ptr(bias), _ptr(residual),
                                      0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), M, w.N, K, _stream()),
          "pfpp_gemm_small")
    return out


WD_PF = _os.environ.get("PFPP_WD_PF", "0") == "1"          # lab: the software-pipelined loop of csrc/gemm_wd.hip (off by default)


def wd_kernel_name(big: bool, single_pass: bool = False, shape=None) -> str:
    """the instantiation pfpp_gemm_wd / pfpp_gemm_wd_f16 launch (csrc/gemm_wd.hip gemm_wd_impl: tile by shape, X1 = single pass; the
    software-pipelined loop from 512 small tiles up — shape = (M, N, K)), spelled as rocprofv3 prints it — bench.py looks the counter
    traffic of the dominant kernel up under this name"""
    if not big and not single_pass and shape is not None and WD_PF:
        M, N, K = shape
        if ((M + 63) // 64) * (N // 128) >= 512 and K >= 224:
            return "gemm_wd_pf_kernel<2, 1, 4>"
    return "gemm_wd_kernel<%s, %s>" % ("4, 2, 4" if big else "2, 1, 3", "true" if single_pass else "false")


def gemm_wd(a, w, *, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, single_pass: bool = False) -> torch.Tensor:
    """a . w^T (+ bias) (+ residual) with the weight's fragment-blocked planes read straight into the matrix operands (pfpp_gemm_wd,
    csrc/gemm_wd.hip): a = SplitAct planes [M, K], w = packing.PW [N, K]; N % 128 == 0, K % 32 == 0; out may be the residual tensor.
    Bit-identical to the tiled plane GEMM (ops.linear on the same planes)."""
    from ._lib import PlanesC, PwC

    M, K = a.hi.shape
    if K != w.K:
        raise ValueError(f"gemm_wd: A has K = {K}, the weight {w.K}")
    fh, fl = w.frag()
    pw = PwC(w.f32.data_ptr(), w.hi.data_ptr(), w.lo.data_ptr(), w.scale, w.hi.shape[-1], fh.data_ptr(), fl.data_ptr())
    for t_, nm in ((bias, "bias"), (residual, "residual"), (out, "out")):
        if t_ is not None:
            _chk(t_, torch.float32, nm)
    if out is None:
        out = torch.empty((M, w.N), dtype=torch.float32, device=a.hi.device)
    ap = PlanesC(a.hi.data_ptr(), a.lo.data_ptr(), 1.0)
    ev = None
    if GEMM_TRACE is not None:            # bench.py: HIP events around the launch, attributed to the instantiation the library picks
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    fn = _lib.load().pfpp_gemm_wd_f16 if single_pass else _lib.load().pfpp_gemm_wd      # single_pass: hi planes only (perf mode)
    check(fn(C.byref(ap), a.hi.stride(0), C.byref(pw), _ptr(bias), _ptr(residual),
             0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), M, w.N, K, _stream()), "pfpp_gemm_wd")
    if ev is not None:
        ev[1].record()
        big = w.N % 256 == 0 and ((M + 127) // 128) * (w.N // 256) >= 240
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * M * w.N * K, wd_kernel_name(big, single_pass, (M, w.N, K)), (M, w.N, K, 1, "none", 0)))
    return out


def reblock_planes(hi: torch.Tensor, lo: torch.Tensor, transposed: bool = False):
    """fragment-blocked copies (pfpp_pw.fhi / flo layout) of row-major planes [N, K] — of the matrix itself, or (transposed) of its
    transpose [K, N] (pfpp_reblock_planes; what pfpp_tlayers_fwd / _bwd do for a training layer's weights) -> (fhi, flo) flat fp16"""
    from ._lib import PlanesC, ReblockJob

    _chk(hi, torch.float16, "hi"); _chk(lo, torch.float16, "lo")
    N, K = hi.shape
    fhi, flo = torch.empty(N * K, dtype=torch.float16, device=hi.device), torch.empty(N * K, dtype=torch.float16, device=hi.device)
    job = ReblockJob(PlanesC(hi.data_ptr(), lo.data_ptr(), 1.0), N, K, hi.stride(0), fhi.data_ptr(), flo.data_ptr(), int(transposed))
    check(_lib.load().pfpp_reblock_planes(C.byref(job), 1, _stream()), "pfpp_reblock_planes")
    return fhi, flo


def layernorm_linear_small(x: torch.Tensor, w, *, mod: Optional[torch.Tensor] = None, group_batch: Optional[torch.Tensor] = None,
                           group_rows: int = 1, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
                           bias: Optional[torch.Tensor] = None, geglu: bool = False, eps: float = 1e-5):
    """LayerNorm(x) . w^T (+ bias) in one launch for few rows (pfpp_layernorm_linear_small): w = packing.PW [N, 512].
    geglu: w / bias are the packed GEGLU pair -> SplitAct planes of u [M, N / 2]; else fp32 [M, N]"""
    from ._lib import PlanesC, PwC

    _chk(x, torch.float32, "x")
    M, Cc = x.shape
    N = w.N
    fh, fl = w.frag()
    pw = PwC(w.f32.data_ptr(), w.hi.data_ptr(), w.lo.data_ptr(), w.scale, w.hi.shape[-1], fh.data_ptr(), fl.data_ptr())
    for t_, nm in ((mod, "mod"), (gamma, "gamma"), (beta, "beta"), (bias, "bias")):
        if t_ is not None:
            _chk(t_, torch.float32, nm)
    if group_batch is not None:
        _chk(group_batch, torch.int32, "group_batch")
    if geglu:
        out = SplitAct.empty(M, N // 2, x.device)
        up = PlanesC(out.hi.data_ptr(), out.lo.data_ptr(), 1.0)
        check(_lib.load().pfpp_layernorm_linear_small(_ptr(x), _ptr(mod), 0 if mod is None else mod.stride(0), _ptr(gamma), _ptr(beta),
                                                      _ptr(group_batch), group_rows, C.byref(pw), _ptr(bias), None, 0, C.byref(up), N // 2,
                                                      M, N, Cc, eps, _stream()), "pfpp_layernorm_linear_small")
        return out
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_lib.load().pfpp_layernorm_linear_small(_ptr(x), _ptr(mod), 0 if mod is None else mod.stride(0), _ptr(gamma), _ptr(beta),
                                                  _ptr(group_batch), group_rows, C.byref(pw), _ptr(bias), _ptr(out), N, None, 0, M, N, Cc,
                                                  eps, _stream()), "pfpp_layernorm_linear_small")
    return out


def attn_blockdiag(qkv: torch.Tensor, n_frag: int, L: int, H: int, dh: int, scale: float, out=None):
    _chk(qkv, torch.float32, "qkv")
    _sync_attention_mode()
    if qkv.shape != (n_frag * L, 3 * H * dh):
        raise ValueError("attn_blockdiag: qkv must be [n_frag*L, 3*H*dh]")
    if isinstance(out, SplitAct):
        check(_lib.load().pfpp_attn_blockdiag_split(_ptr(qkv), _ptr(out.hi), _ptr(out.lo), n_frag, L, H, dh, scale,
                                                    _stream()), "pfpp_attn_blockdiag_split")
        return out
    if out is None:
        out = torch.empty((n_frag * L, H * dh), dtype=torch.float32, device=qkv.device)
    check(_lib.load().pfpp_attn_blockdiag(_ptr(qkv), _ptr(out), n_frag, L, H, dh, scale, _stream()),
          "pfpp_attn_blockdiag")
    return out


def attn_dense(qkv: torch.Tensor, seq_off: torch.Tensor, seq_len: torch.Tensor, max_len: int, H: int, dh: int,
               scale: float, key_valid_u8: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused softmax(QK^T*scale + key mask)V from a packed [rows, 3*H*dh] projection; sequences are row
    ranges [seq_off[s], seq_off[s]+seq_len[s]) (int32); key_valid [n_seq, max_len] uint8 or None"""
    _sync_attention_mode()
    _chk(qkv, torch.float32, "qkv"); _chk(seq_off, torch.int32, "seq_off"); _chk(seq_len, torch.int32, "seq_len")
    rows, w = qkv.shape
    if w != 3 * H * dh:
        raise ValueError("attn_dense: qkv must be [rows, 3*H*dh]")
    kv_stride = 0
    if key_valid_u8 is not None:
        _chk(key_valid_u8, torch.uint8, "key_valid")
        kv_stride = key_valid_u8.shape[-1]
    if isinstance(out, SplitAct):
        check(_lib.load().pfpp_attn_dense_split(_ptr(qkv), _ptr(out.hi), _ptr(out.lo), _ptr(seq_off), _ptr(seq_len),
                                                _ptr(key_valid_u8), kv_stride, seq_off.numel(), max_len, H, dh, scale,
                                                _stream()), "pfpp_attn_dense_split")
        return out
    if out is None:
        out = torch.empty((rows, H * dh), dtype=torch.float32, device=qkv.device)
    else:
        _chk(out, torch.float32, "out")
    check(_lib.load().pfpp_attn_dense(_ptr(qkv), _ptr(out), _ptr(seq_off), _ptr(seq_len), _ptr(key_valid_u8), kv_stride,
                                      seq_off.numel(), max_len, H, dh, scale, _stream()), "pfpp_attn_dense")
    return out


def softmax_rows(S: torch.Tensor, key_valid_u8: Optional[torch.Tensor], rows_per_batch: int, T: int,
                 scale: float) -> torch.Tensor:
    """in-place masked softmax over the first T columns of S [..., ld]"""
    _chk(S, torch.float32, "S")
    ld = S.shape[-1]
    rows = S.numel() // ld
    if key_valid_u8 is not None:
        _chk(key_valid_u8, torch.uint8, "key_valid")
    check(_lib.load().pfpp_softmax_rows(_ptr(S), _ptr(key_valid_u8), rows, rows_per_batch, T, ld, scale, _stream()),
          "pfpp_softmax_rows")
    return S


def mean_pool(x: torch.Tensor, n: int, L: int) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    Cc = x.shape[-1]
    out = torch.empty((n, Cc), dtype=torch.float32, device=x.device)
    check(_lib.load().pfpp_mean_pool(_ptr(x), _ptr(out), n, L, Cc, _stream()), "pfpp_mean_pool")
    return out


def ddpm_step(x, eps, noise, ref_part_u8, reference, coef) -> torch.Tensor:
    """coef = (c_eps, c_div, c_x0, c_x, c_noise) python floats (already fp32-rounded)"""
    _chk(x, torch.float32, "x"); _chk(eps, torch.float32, "eps")
    if noise is not None:
        _chk(noise, torch.float32, "noise")
    if ref_part_u8 is not None:
        _chk(ref_part_u8, torch.uint8, "ref_part"); _chk(reference, torch.float32, "reference")
    n = x.numel() // 7
    out = torch.empty_like(x)
    check(_lib.load().pfpp_ddpm_step(_ptr(x), _ptr(eps), _ptr(noise), _ptr(ref_part_u8), _ptr(reference), _ptr(out),
                                     n, *[float(c) for c in coef], _stream()), "pfpp_ddpm_step")
    return out


def add_noise(x0, noise, sqrt_ab, sqrt_1mab) -> torch.Tensor:
    for t, nm in ((x0, "x0"), (noise, "noise"), (sqrt_ab, "sqrt_ab"), (sqrt_1mab, "sqrt_1mab")):
        _chk(t, torch.float32, nm)
    B = x0.shape[0]
    out = torch.empty_like(x0)
    check(_lib.load().pfpp_add_noise(_ptr(x0), _ptr(noise), _ptr(sqrt_ab), _ptr(sqrt_1mab), _ptr(out), B,
                                     x0.numel() // B, _stream()), "pfpp_add_noise")
    return out


def verifier_embed(feat_emb, edge_idx, pe) -> torch.Tensor:
    _chk(feat_emb, torch.float32, "feat_emb"); _chk(edge_idx, torch.int64, "edge_indices"); _chk(pe, torch.float32, "pe")
    n, Cc = feat_emb.shape
    tok = torch.empty_like(feat_emb)
    check(_lib.load().pfpp_verifier_embed(_ptr(feat_emb), _ptr(edge_idx), _ptr(pe), _ptr(tok), n, Cc, pe.shape[0],
                                          _stream()), "pfpp_verifier_embed")
    return tok


def pose_compose(pose, pivot, init_pose=None, has_init=None) -> torch.Tensor:
    _chk(pose, torch.float32, "pose"); _chk(pivot, torch.int32, "pivot")
    n = pivot.numel()
    if init_pose is not None:
        _chk(init_pose, torch.float32, "init_pose"); _chk(has_init, torch.uint8, "has_init")
    out = torch.empty((n, 7), dtype=torch.float32, device=pose.device)
    check(_lib.load().pfpp_pose_compose(_ptr(pose), _ptr(pivot), _ptr(init_pose), _ptr(has_init), _ptr(out), n,
                                        _stream()), "pfpp_pose_compose")
    return out


def pose_apply_points(pts: torch.Tensor, pose_idx: torch.Tensor, pose: torch.Tensor, normalise: bool = False) -> torch.Tensor:
    """pts [n,3], pose_idx int32 [n], pose [P,7] -> R(q[pose_idx]) p + t[pose_idx]  (node_merge_utils.py:16-41)"""
    _chk(pts, torch.float32, "pts"); _chk(pose_idx, torch.int32, "pose_idx"); _chk(pose, torch.float32, "pose")
    out = torch.empty_like(pts)
    check(_lib.load().pfpp_pose_apply_points(_ptr(pts), _ptr(pose_idx), _ptr(pose), _ptr(out), pts.shape[0], int(normalise),
                                             _stream()), "pfpp_pose_apply_points")
    return out


def edge_histogram(pts: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor, edge_off: torch.Tensor, max_m: int) -> torch.Tensor:
    """per-edge 6-bin histogram of the bidirectional nearest-neighbour distances of the matched points"""
    _chk(pts, torch.float32, "pts")
    for t, nm in ((idx_a, "idx_a"), (idx_b, "idx_b"), (edge_off, "edge_off")):
        _chk(t, torch.int32, nm)
    n_edges = edge_off.numel() - 1
    hist = torch.empty((n_edges, 6), dtype=torch.int32, device=pts.device)
    check(_lib.load().pfpp_edge_histogram(_ptr(pts), _ptr(idx_a), _ptr(idx_b), _ptr(edge_off), _ptr(hist), n_edges, max_m,
                                          _stream()), "pfpp_edge_histogram")
    return hist


# --------------------------------------------------------------------------- evaluation metrics (8f-3)
def nn_dist(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """src [B,n,3], dst [B,m,3] -> [B,n] squared distance to the nearest neighbour (chamferdist's KNN-1 term)"""
    _chk(src, torch.float32, "src"); _chk(dst, torch.float32, "dst")
    if src.dim() != 3 or dst.dim() != 3 or src.shape[0] != dst.shape[0] or src.shape[2] != 3 or dst.shape[2] != 3:
        raise ValueError("nn_dist: src [B,n,3] and dst [B,m,3] expected")
    B, n, _ = src.shape
    out = torch.empty((B, n), dtype=torch.float32, device=src.device)
    lib = _lib.load()
    for b0 in range(0, B, 65535):
        b1 = min(B, b0 + 65535)
        check(lib.pfpp_nn_dist(_ptr(src[b0:b1]), _ptr(dst[b0:b1]), _ptr(out[b0:b1]), b1 - b0, n, dst.shape[1], _stream()),
              "pfpp_nn_dist")
    return out


def quat_to_euler_xyz(quat: torch.Tensor, to_degree: bool = True) -> torch.Tensor:
    """[..., 4] (w first) -> [..., 3] Euler angles, convention XYZ (transform.quaternion_to_euler)"""
    _chk(quat, torch.float32, "quat")
    if quat.shape[-1] != 4:
        raise ValueError("quat_to_euler_xyz: last dimension must be 4")
    out = torch.empty(quat.shape[:-1] + (3,), dtype=torch.float32, device=quat.device)
    check(_lib.load().pfpp_quat_to_euler_xyz(_ptr(quat), _ptr(out), quat.numel() // 4, int(to_degree), _stream()),
          "pfpp_quat_to_euler_xyz")
    return out


# --------------------------------------------------------------------------- merge step (8f-2)
def estimate_normals(pts: torch.Tensor, k: int = 20) -> torch.Tensor:
    """pts [P,N,3] -> unit normals [P,N,3] (pytorch3d.ops.estimate_pointcloud_normals, neighborhood_size=k)"""
    _chk(pts, torch.float32, "pts")
    if pts.dim() != 3 or pts.shape[2] != 3:
        raise ValueError("estimate_normals: pts must be [P,N,3]")
    P, N, _ = pts.shape
    out = torch.empty_like(pts)
    check(_lib.load().pfpp_estimate_normals(_ptr(pts), _ptr(out), P, N, k, _stream()), "pfpp_estimate_normals")
    return out


def merge_keep_mask(d: torch.Tensor, normals: torch.Tensor, threshold: float = 1e-3) -> torch.Tensor:
    """d [P,P,N] nearest-neighbour distances part i -> part j, normals [P,N,3] -> keep bool [P,N]
    (node_merge_utils.py:176-205)"""
    _chk(d, torch.float32, "d"); _chk(normals, torch.float32, "normals")
    P, _, N = d.shape
    keep = torch.empty((P, N), dtype=torch.uint8, device=d.device)
    check(_lib.load().pfpp_merge_keep_mask(_ptr(d), _ptr(normals), _ptr(keep), P, N, threshold, _stream()),
          "pfpp_merge_keep_mask")
    return keep.bool()
