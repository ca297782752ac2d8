"""Host-side weight packing for the HIP kernels.

The modules keep the reference's parameter names and shapes (so published state_dicts
load unchanged); the kernels read re-laid-out copies built here.  Copies are cached by the
owning module and rebuilt when a parameter changes (see PackCache).
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import math

import torch


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pad_k(w: torch.Tensor, mult: int = 4) -> torch.Tensor:
    """[N,K] -> [N, round_up(K)] zero padded (16-byte aligned rows for the GEMM loader)"""
    n, k = w.shape
    kp = round_up(k, mult)
    if kp == k:
        return w.contiguous()
    out = w.new_zeros((n, kp))
    out[:, :k] = w
    return out


def split_f16(w: torch.Tensor):
    """fp32 [..., N, K] -> (hi, lo) fp16 planes [..., N, round_up(K, 8)] for the split-f16 GEMM:
    hi = f16(w), lo = f16(w - hi) (unscaled); zero padded (csrc/gemm.hip, PFPP_GEMM_F16X3)"""
    k = w.shape[-1]
    kp = round_up(k, 8)
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    if kp != k:
        pad = (0, kp - k)
        hi = torch.nn.functional.pad(hi, pad)
        lo = torch.nn.functional.pad(lo, pad)
    return hi.contiguous(), lo.contiguous()


def plane_scale(w: torch.Tensor) -> float:
    """power of two that lifts max |w| into [2^12, 2^13): every element down to max |w| * 2^-16 then keeps the split's full 22 bits
    (the fp16 `lo` plane bottoms out at 2^-24 absolute: unscaled, elements below 2^-3 lose relative precision, and tensors of
    1e-6-sized values lose everything), and nothing can reach the fp16 overflow at 65504 whatever the checkpoint's magnitudes.
    Exact: the GEMM multiplies its accumulator by 1 / scale (ops.gemm)."""
    amax = float(w.detach().abs().max()) if w.numel() else 0.0
    if not math.isfinite(amax) or amax == 0.0:
        return 1.0
    # exponent clamped to 2^±100: a (near-)denormal tensor would otherwise ask for a scale beyond the fp32 range (w * scale = inf,
    # alpha = 1 / scale = 0 -> NaN); such a tensor contributes nothing at fp32 precision either way
    return 2.0 ** max(-100, min(100, 12 - math.floor(math.log2(amax))))


class PW:
    """a GEMM weight in kernel layout: fp32 [N, K4] (K padded to 4) for the exact path and the
    pre-split fp16 planes [N, K8] of `scale` * w for the split-f16 path (`scale`: a per-tensor power of two chosen at pack time,
    packing.plane_scale; 1.0 with prescale=False); `K` is the true reduction length"""

    __slots__ = ("f32", "hi", "lo", "N", "K", "scale", "_frag")

    def __init__(self, w: torch.Tensor, K: int = None, prescale: bool = True):
        w2 = w.reshape(-1, w.shape[-1])
        self.K = int(K if K is not None else w.shape[-1])
        self.N = int(w.shape[-2])
        self.f32 = pad_k(w2).reshape(*w.shape[:-1], -1).contiguous()
        wk = w[..., : self.K] if self.K != w.shape[-1] else w
        self.scale = plane_scale(wk) if prescale else 1.0
        self.hi, self.lo = split_f16(wk * self.scale if self.scale != 1.0 else wk)
        self._frag = None

    def frag(self):
        """(fhi, flo): the planes fragment-blocked (include/pfpp.h pfpp_pw.fhi / flo) for the few-token kernels, built on first use:
        block (row tile n // 32, k-step k // 16) = the 64 lanes' B operands of one 32x32x16 MFMA, 1 KB contiguous"""
        if self._frag is None:
            if self.hi.dim() != 2 or self.N % 32 or self.K % 16 or self.hi.shape[-1] != self.K:
                raise ValueError(f"PW.frag: [N % 32 == 0, K % 16 == 0] weights only, got N {self.N} K {self.K}")
            def blocked(pl):
                return pl.view(self.N // 32, 32, self.K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
            self._frag = (blocked(self.hi), blocked(self.lo))
        return self._frag

    @property
    def device(self):
        return self.f32.device


def fold_conv_bn(conv_w, conv_b, bn_w, bn_b, mean, var, eps: float = 1e-5):
    """eval-mode BatchNorm folded into a per-channel scale/shift applied to the raw GEMM
    accumulator:  bn(conv(x)) = acc*s + ((b - mean)*s + beta),  s = gamma / sqrt(var + eps)
    (utils/pn2_utils.py:210-212)."""
    w = conv_w.reshape(conv_w.shape[0], -1)
    s = bn_w / torch.sqrt(var + eps)
    t = (conv_b - mean) * s + bn_b
    return w, s.contiguous(), t.contiguous()


def pack_sa_first(w: torch.Tensor, d_feat: int) -> torch.Tensor:
    """first conv of a set-abstraction level: input channels are [xyz(3) | feats(D)] in the
    reference (utils/pn2_utils.py:146); the grouping kernel emits [feats(D) | xyz(3) | 0]."""
    o, k = w.shape
    assert k == d_feat + 3
    out = w.new_zeros((o, d_feat + 4))
    out[:, :d_feat] = w[:, 3:]
    out[:, d_feat:d_feat + 3] = w[:, :3]
    return out


def pack_geglu(w: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """GEGLU projection [2I, C] (value rows then gate rows, diffusers `chunk(2)`) re-ordered in
    blocks of 64 rows = 32 value rows followed by their 32 gate rows, so that one wave's two
    MFMA column tiles hold a value and its gate in the same lane (csrc/gemm.hip)."""
    two_i, c = w.shape
    inner = two_i // 2
    assert inner % 32 == 0
    nb = inner // 32
    wv = w[:inner].reshape(nb, 32, c)
    wg = w[inner:].reshape(nb, 32, c)
    wp = torch.stack((wv, wg), dim=1).reshape(two_i, c).contiguous()
    bp = torch.stack((b[:inner].reshape(nb, 32), b[inner:].reshape(nb, 32)), dim=1).reshape(two_i).contiguous()
    return wp, bp


class PackCache:
    """rebuilds packed weights when any source tensor was modified in place or replaced"""

    def __init__(self):
        self._key = None
        self.packed: Dict[str, torch.Tensor] = {}

    @staticmethod
    def _fingerprint(tensors: Iterable[torch.Tensor]):
        return tuple((t.data_ptr(), t._version, t.device) for t in tensors)

    def get(self, tensors, builder):
        key = self._fingerprint(tensors)
        if key != self._key:
            with torch.no_grad():
                self.packed = builder()
            self._key = key
        return self.packed
