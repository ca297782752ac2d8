"""Tensor-level wrappers over the training entry points of the C ABI (include/pfpp.h, section a17).

Same rules as ops.py: validation on the host, outputs from torch's allocator, kernels on the current HIP
stream, no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT, GemmGradArgs, check
from . import ops
from .ops import _chk, _ptr, _stream

_f32 = torch.float32
import os as _os

SPLIT_DX = _os.environ.get("PFPP_SPLIT_DX", "1") == "1"


def gemm_grad(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldw: int,
              ldc: int, a_kmajor: bool = False, w_kmajor: bool = False, accumulate: bool = False,
              split_k: int = 0, batch: int = 1, sA: int = 0, sW: int = 0, sC: int = 0, a_scale: float = 1.0,
              w_scale: float = 1.0, alpha: float = 1.0, a_off: int = 0, w_off: int = 0, c_off: int = 0,
              colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """raw pfpp_gemm_grad: out[M,N] (+)= alpha * sum_k A(m,k) W(n,k) with per-operand k-major layouts;
    colsum [M] += sum_k A(m,k) (weight-gradient form only: the bias gradient from the same pass over dY)"""
    _chk(A, _f32, "A"); _chk(W, _f32, "W"); _chk(out, _f32, "out")
    a = GemmGradArgs()
    if colsum is not None:
        _chk(colsum, _f32, "colsum")
        if colsum.numel() < M or not colsum.is_contiguous():
            raise ValueError("gemm_grad: colsum must be a contiguous [M] tensor")
        a.colsum = colsum.data_ptr()
    a.A = A.data_ptr() + 4 * a_off
    a.W = W.data_ptr() + 4 * w_off
    a.C = out.data_ptr() + 4 * c_off
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = lda, ldw, ldc
    a.a_kmajor, a.w_kmajor = int(a_kmajor), int(w_kmajor)
    a.accumulate, a.split_k, a.batch = int(accumulate), split_k, batch
    a.sA, a.sW, a.sC = sA, sW, sC
    a.a_scale, a.w_scale, a.alpha = a_scale, w_scale, alpha
    if ops.GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_gemm_grad(C.byref(a), _stream()), "pfpp_gemm_grad")
        e1.record()
        ops.GEMM_TRACE.append((e0, e1, 2.0 * M * N * K * batch, grad_kernel_name(M, N, batch, a_kmajor, w_kmajor, split_k, accumulate),
                               (M, N, K, batch, "grad", 0)))
        return out
    check(_lib.load().pfpp_gemm_grad(C.byref(a), _stream()), "pfpp_gemm_grad")
    return out


def grad_kernel_name(M: int, N: int, batch: int, a_kmajor: bool, w_kmajor: bool, split_k: int, accumulate: bool) -> str:
    """the gemm_grad_kernel instantiation csrc/gemm_grad.hip dispatches to (mirror of its tile choice)"""
    def tiles(bm, bn):
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * batch
    bm, bn = 128, 64
    if tiles(256, 128) >= 384:
        bm, bn = 256, 128
    elif tiles(128, 128) >= 256 or N > 64:
        bm, bn = 128, 128
    if (bm, bn) == (128, 128) and tiles(128, 128) < 192 and split_k == 1 and N <= 2048:
        bn = 64
    if (bm, bn) == (128, 128) and accumulate and tiles(128, 128) < 40:
        bn = 64
    cfg = {(256, 128): "2, 2, 4, 2", (128, 128): "2, 2, 2, 2", (128, 64): "2, 1, 2, 2"}[(bm, bn)]
    return f"gemm_grad_kernel<{cfg}, {'true' if a_kmajor else 'false'}, {'true' if w_kmajor else 'false'}>"


def dx_splits(M: int, K_in: int, N_out: int) -> bool:
    """does grad_input cut this contraction over several workgroups (atomics into a ZEROED output)?  Few output tiles and a
    long contraction only."""
    tiles = ((M + 127) // 128) * ((K_in + 127) // 128)
    return SPLIT_DX and tiles < 256 and N_out >= 1024


def grad_input(dY: torch.Tensor, W: torch.Tensor, *, g_scale: float = 1.0, out: Optional[torch.Tensor] = None,
               zeroed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX [M, in] = dY [M, out] . W [out, in]   (W in torch Linear layout).  `zeroed`: a zero-filled [M, in] buffer the caller
    prepared ahead (used instead of a fresh torch.zeros when the contraction is split, see dx_splits)"""
    M, N_out = dY.shape
    K_in = W.shape[1]
    if W.shape[0] != N_out:
        raise ValueError("grad_input: dY [M,out] and W [out,in] expected")
    split = out is None and dx_splits(M, K_in, N_out)
    if out is None:
        if split and zeroed is not None:
            if zeroed.shape != (M, K_in):
                raise ValueError("grad_input: zeroed buffer has the wrong shape")
            out = zeroed
        else:
            out = (torch.zeros if split else torch.empty)((M, K_in), dtype=_f32, device=dY.device)
    return gemm_grad(dY, W, out, M=M, N=K_in, K=N_out, lda=dY.stride(0), ldw=W.stride(0), ldc=out.stride(0),
                     w_kmajor=True, a_scale=g_scale, accumulate=split, split_k=0 if split else 1)


def grad_weight(dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, *, g_scale: float = 1.0, k_cols: Optional[int] = None,
                db: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [out, in] += dY [M, out]^T . X [M, in]   (accumulates with atomics; dW must be initialised);
    db [out] += column sums of dY in the same launch (the bias gradient)"""
    M, N_out = dY.shape
    K_in = X.shape[1] if k_cols is None else k_cols
    if X.shape[0] != M or dW.shape[0] != N_out:
        raise ValueError("grad_weight: shape mismatch")
    return gemm_grad(dY, X, dW, M=N_out, N=K_in, K=M, lda=dY.stride(0), ldw=X.stride(0), ldc=dW.stride(0),
                     a_kmajor=True, w_kmajor=True, accumulate=True, a_scale=g_scale, colsum=db)


def grad_weight_group(problems, *, g_scale: float = 1.0) -> None:
    """[(dY [M,out], X [M,in], dW [out,in]), ...] (at most 8): every dW += dY^T . X in ONE launch (pfpp_gemm_grad_group)"""
    n = len(problems)
    if not 1 <= n <= 8:
        raise ValueError("grad_weight_group: 1..8 problems")
    arr = (GemmGradArgs * n)()
    flops = 0.0
    for a, (dY, X, dW) in zip(arr, problems):
        _chk(dY, _f32, "dY"); _chk(X, _f32, "X"); _chk(dW, _f32, "dW")
        M, N_out = dY.shape
        if X.shape[0] != M or dW.shape[0] != N_out or dW.shape[1] > X.shape[1]:
            raise ValueError("grad_weight_group: shape mismatch")
        a.A, a.W, a.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
        a.M, a.N, a.K = N_out, dW.shape[1], M
        a.lda, a.ldw, a.ldc = dY.stride(0), X.stride(0), dW.stride(0)
        a.a_kmajor = a.w_kmajor = 1
        a.accumulate, a.split_k, a.batch = 1, 0, 1
        a.sA = a.sW = a.sC = 0
        a.a_scale, a.w_scale, a.alpha = g_scale, 1.0, 1.0
        flops += 2.0 * N_out * dW.shape[1] * M
    if ops.GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().pfpp_gemm_grad_group(arr, n, _stream()), "pfpp_gemm_grad_group")
        e1.record()
        ops.GEMM_TRACE.append((e0, e1, flops, "gemm_grad_group_kernel<2, 2, 2, 2, true, true>", (n, int(flops // 1e6), 0, 1, "grad_group", 0)))
        return
    check(_lib.load().pfpp_gemm_grad_group(arr, n, _stream()), "pfpp_gemm_grad_group")


def colsum(x: torch.Tensor, out: torch.Tensor, *, rows: Optional[int] = None, cols: Optional[int] = None,
           ld: Optional[int] = None, batch: int = 1, sx: int = 0, so: int = 0, accumulate: bool = True,
           x_off: int = 0, o_off: int = 0) -> torch.Tensor:
    """out[c] (+)= sum_r x[r, c]"""
    _chk(x, _f32, "x"); _chk(out, _f32, "out")
    if rows is None:
        rows, cols = x.shape
        ld = x.stride(0)
    check(_lib.load().pfpp_colsum(C.c_void_p(x.data_ptr() + 4 * x_off), C.c_void_p(out.data_ptr() + 4 * o_off), rows, cols,
                                  ld, batch, sx, so, int(accumulate), _stream()), "pfpp_colsum")
    return out


def dropout(x: torch.Tensor, p: float, seed: int, site: int, *, res: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = (res or 0) + x * keep / (1 - p) with the counter-based mask of (seed, site)"""
    _chk(x, _f32, "x")
    if res is not None:
        _chk(res, _f32, "res")
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().pfpp_dropout(_ptr(x), _ptr(res), _ptr(out), x.numel(), p, seed, site, _stream()), "pfpp_dropout")
    return out


def dropout_mask(n: int, p: float, seed: int, site: int, device) -> torch.Tensor:
    keep = torch.empty((n,), dtype=torch.uint8, device=device)
    check(_lib.load().pfpp_dropout_mask(_ptr(keep), n, p, seed, site, _stream()), "pfpp_dropout_mask")
    return keep


def geglu(z: torch.Tensor, p: float = 0.0, seed: int = 0, site: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(z, _f32, "z")
    rows, two_inner = z.shape
    inner = two_inner // 2
    if out is None:
        out = torch.empty((rows, inner), dtype=_f32, device=z.device)
    check(_lib.load().pfpp_geglu(_ptr(z), _ptr(out), rows, inner, p, seed, site, _stream()), "pfpp_geglu")
    return out


def geglu_bwd(z: torch.Tensor, du: torch.Tensor, p: float = 0.0, seed: int = 0, site: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(z, _f32, "z"); _chk(du, _f32, "du")
    rows, two_inner = z.shape
    if out is None:
        out = torch.empty_like(z)
    check(_lib.load().pfpp_geglu_bwd(_ptr(z), _ptr(du), _ptr(out), rows, two_inner // 2, p, seed, site, _stream()),
          "pfpp_geglu_bwd")
    return out


def act(pre: torch.Tensor, kind: str) -> torch.Tensor:
    _chk(pre, _f32, "pre")
    out = torch.empty_like(pre)
    check(_lib.load().pfpp_act(_ptr(pre), _ptr(out), pre.numel(), ACT[kind], _stream()), "pfpp_act")
    return out


def act_bwd(pre: torch.Tensor, dy: torch.Tensor, kind: str) -> torch.Tensor:
    _chk(pre, _f32, "pre"); _chk(dy, _f32, "dy")
    out = torch.empty_like(pre)
    check(_lib.load().pfpp_act_bwd(_ptr(pre), _ptr(dy), _ptr(out), pre.numel(), ACT[kind], _stream()), "pfpp_act_bwd")
    return out


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, *, mod: Optional[torch.Tensor] = None,
                  gamma: Optional[torch.Tensor] = None, group_batch: Optional[torch.Tensor] = None,
                  group_rows: int = 32, rows_per_batch: int = 1, dmult: Optional[torch.Tensor] = None,
                  dadd: Optional[torch.Tensor] = None, ld_d: int = 0, eps: float = 1e-5,
                  drop: Optional[tuple] = None) -> torch.Tensor:
    """dx += LayerNorm backward; dmult/dadd (+)= the (scale, shift) / (gamma, beta) gradients (see pfpp.h).
    drop = (p, seed, site): also returns dropout(dx) of that site as a new tensor (pfpp_layernorm_bwd_dropout) instead of dx"""
    _chk(x, _f32, "x"); _chk(dy, _f32, "dy"); _chk(dx, _f32, "dx")
    rows, Cc = x.shape
    ld_mod = 0
    if mod is not None:
        _chk(mod, _f32, "mod")
        ld_mod = mod.stride(0)
    if group_batch is not None:
        _chk(group_batch, torch.int32, "group_batch")
        if group_batch.numel() * group_rows < rows:
            raise ValueError("layernorm_bwd: group_batch too short")
    if drop is not None:
        p, seed, site = drop
        out = torch.empty_like(dx)
        check(_lib.load().pfpp_layernorm_bwd_dropout(_ptr(x), _ptr(dy), _ptr(mod), ld_mod, _ptr(gamma), _ptr(group_batch), group_rows,
                                                     rows_per_batch, _ptr(dx), _ptr(dmult), _ptr(dadd), ld_d, rows, Cc, eps, _ptr(out),
                                                     p, seed, site, _stream()), "pfpp_layernorm_bwd_dropout")
        return out
    check(_lib.load().pfpp_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(mod), ld_mod, _ptr(gamma), _ptr(group_batch), group_rows,
                                         rows_per_batch, _ptr(dx), _ptr(dmult), _ptr(dadd), ld_d, rows, Cc, eps, _stream()),
          "pfpp_layernorm_bwd")
    return dx


def dropout_layernorm(y: torch.Tensor, res: Optional[torch.Tensor], p: float, seed: int, site: int, *,
                      mod: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                      beta: Optional[torch.Tensor] = None, group_batch: Optional[torch.Tensor] = None, group_rows: int = 1,
                      rows_per_batch: int = 1, eps: float = 1e-5):
    """-> (h, n): h = (res or 0) + dropout(y) written over y, n = LayerNorm(h) (AdaLN `mod` [B, 2C] or gamma/beta) — one launch
    for pfpp_dropout + pfpp_layernorm (pfpp_dropout_layernorm)"""
    _chk(y, _f32, "y")
    rows, Cc = y.shape
    ld_mod = 0
    if res is not None:
        _chk(res, _f32, "res")
    if mod is not None:
        _chk(mod, _f32, "mod")
        ld_mod = mod.stride(0)
        if mod.shape[-1] != 2 * Cc:
            raise ValueError("dropout_layernorm: mod must be [B, 2C]")
    if group_batch is not None:
        _chk(group_batch, torch.int32, "group_batch")
        if group_batch.numel() * group_rows < rows:
            raise ValueError("dropout_layernorm: group_batch too short")
    n = torch.empty_like(y)
    check(_lib.load().pfpp_dropout_layernorm(_ptr(y), _ptr(res), _ptr(y), _ptr(n), _ptr(mod), ld_mod, _ptr(gamma), _ptr(beta),
                                             _ptr(group_batch), group_rows, rows_per_batch, rows, Cc, eps, p, seed, site, _stream()),
          "pfpp_dropout_layernorm")
    return y, n


def attn_blockdiag_bwd(qkv: torch.Tensor, dout: torch.Tensor, n_frag: int, L: int, H: int, dh: int, scale: float,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(qkv, _f32, "qkv"); _chk(dout, _f32, "dout")
    if out is None:
        out = torch.empty_like(qkv)
    check(_lib.load().pfpp_attn_blockdiag_bwd(_ptr(qkv), _ptr(dout), _ptr(out), n_frag, L, H, dh, scale, _stream()),
          "pfpp_attn_blockdiag_bwd")
    return out


def attn_dense_train(qkv: torch.Tensor, seq_off: torch.Tensor, seq_len: torch.Tensor, max_len: int, H: int, dh: int,
                     scale: float, key_valid: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """pfpp_attn_dense that also returns lse [rows, H] for the backward"""
    _chk(qkv, _f32, "qkv"); _chk(seq_off, torch.int32, "seq_off"); _chk(seq_len, torch.int32, "seq_len")
    rows = qkv.shape[0]
    if out is None:
        out = torch.empty((rows, H * dh), dtype=_f32, device=qkv.device)
    lse = torch.empty((rows, H), dtype=_f32, device=qkv.device)
    kvs = 0
    if key_valid is not None:
        _chk(key_valid, torch.uint8, "key_valid")
        kvs = key_valid.stride(0)
    check(_lib.load().pfpp_attn_dense_train(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(seq_off), _ptr(seq_len), _ptr(key_valid),
                                            kvs, seq_off.numel(), max_len, H, dh, scale, _stream()), "pfpp_attn_dense_train")
    return out, lse


def attn_dense_bwd(qkv: torch.Tensor, out_fwd: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, seq_off: torch.Tensor,
                   seq_len: torch.Tensor, max_len: int, H: int, dh: int, scale: float,
                   key_valid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                   aux_stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """dqkv of the dense (ragged, key-masked) attention.  With `aux_stream` the dk/dv pass runs there while dq runs on the
    current stream (both after the small D = rowsum(dout . out) kernel); the current stream waits for both before returning."""
    _chk(qkv, _f32, "qkv"); _chk(out_fwd, _f32, "out_fwd"); _chk(dout, _f32, "dout"); _chk(lse, _f32, "lse")
    if out is None:
        out = torch.empty_like(qkv)
    dvec = torch.empty_like(lse)
    kvs = 0
    if key_valid is not None:
        _chk(key_valid, torch.uint8, "key_valid")
        kvs = key_valid.stride(0)
    if aux_stream is not None:
        def part(bits):
            check(_lib.load().pfpp_attn_dense_bwd_parts(_ptr(qkv), _ptr(out_fwd), _ptr(dout), _ptr(lse), _ptr(dvec), _ptr(out),
                                                        _ptr(seq_off), _ptr(seq_len), _ptr(key_valid), kvs, seq_off.numel(), max_len,
                                                        H, dh, scale, bits, _stream()), "pfpp_attn_dense_bwd_parts")
        main = torch.cuda.current_stream()
        part(1)
        aux_stream.wait_stream(main)
        with torch.cuda.stream(aux_stream):
            part(4)
        part(2)
        main.wait_stream(aux_stream)
        for t in (qkv, dout, lse, dvec, out):
            t.record_stream(aux_stream)
        return out
    check(_lib.load().pfpp_attn_dense_bwd(_ptr(qkv), _ptr(out_fwd), _ptr(dout), _ptr(lse), _ptr(dvec), _ptr(out), _ptr(seq_off),
                                          _ptr(seq_len), _ptr(key_valid), kvs, seq_off.numel(), max_len, H, dh, scale,
                                          _stream()), "pfpp_attn_dense_bwd")
    return out


def mean_pool_bwd(dpooled: torch.Tensor, L: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(dpooled, _f32, "dpooled")
    n, Cc = dpooled.shape
    if out is None:
        out = torch.empty((n * L, Cc), dtype=_f32, device=dpooled.device)
    check(_lib.load().pfpp_mean_pool_bwd(_ptr(dpooled), _ptr(out), n, L, Cc, _stream()), "pfpp_mean_pool_bwd")
    return out


def token_combine_bwd(dtok: torch.Tensor, ref_u8: torch.Tensor, dref_emb: torch.Tensor, L: int,
                      slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dx_emb [n, C]; dref_emb [2, C] accumulates.  slot: ref_u8 is the padded [n_slots] array, fragment f reads ref_u8[slot[f]]"""
    _chk(dtok, _f32, "dtok"); _chk(ref_u8, torch.uint8, "ref_part"); _chk(dref_emb, _f32, "dref_emb")
    n = ref_u8.numel() if slot is None else slot.numel()
    Cc = dtok.shape[1]
    dx_emb = torch.empty((n, Cc), dtype=_f32, device=dtok.device)
    if slot is not None:
        _chk(slot, torch.int32, "slot")
        check(_lib.load().pfpp_token_combine_bwd_slots(_ptr(dtok), _ptr(ref_u8), _ptr(slot), _ptr(dx_emb), _ptr(dref_emb), n, L, Cc,
                                                       _stream()), "pfpp_token_combine_bwd_slots")
        return dx_emb
    check(_lib.load().pfpp_token_combine_bwd(_ptr(dtok), _ptr(ref_u8), _ptr(dx_emb), _ptr(dref_emb), n, L, Cc, _stream()),
          "pfpp_token_combine_bwd")
    return dx_emb


def silu_embed_bwd(tables: torch.Tensor, t: torch.Tensor, dse: torch.Tensor, dtables: torch.Tensor,
                   active: Optional[torch.Tensor] = None) -> torch.Tensor:
    """active (int32 [ceil(rows / 32)], optional): the rows t[.] are also marked in this bitmap (adamw_rows_active)"""
    _chk(tables, _f32, "tables"); _chk(t, torch.int64, "t"); _chk(dse, _f32, "dse"); _chk(dtables, _f32, "dtables")
    n_tab, n_emb, Cc = tables.shape
    if active is not None:
        _chk(active, torch.int32, "active")
        if active.numel() * 32 < n_emb:
            raise ValueError("silu_embed_bwd: the bitmap has fewer bits than the tables have rows")
        check(_lib.load().pfpp_silu_embed_bwd_mark(_ptr(tables), _ptr(t), _ptr(dse), _ptr(dtables), n_tab, n_emb, t.numel(), Cc,
                                                   _ptr(active), _stream()), "pfpp_silu_embed_bwd_mark")
        return dtables
    check(_lib.load().pfpp_silu_embed_bwd(_ptr(tables), _ptr(t), _ptr(dse), _ptr(dtables), n_tab, n_emb, t.numel(), Cc,
                                          _stream()), "pfpp_silu_embed_bwd")
    return dtables


def embed_pack_weights(w_shape: torch.Tensor, w_param: torch.Tensor, b_shape: torch.Tensor, b_param: torch.Tensor, fhi: torch.Tensor,
                       flo: torch.Tensor, bias: torch.Tensor) -> None:
    """[W_shape | W_param | 0] [C, 320] -> fragment-blocked split-f16 planes fhi / flo (C * 320 halfs each), bias = b_shape + b_param:
    the weight operand of the one-launch token embedding, from the step's fp32 parameters (csrc/embed_train.hip)"""
    for t_, nm in ((w_shape, "w_shape"), (w_param, "w_param"), (b_shape, "b_shape"), (b_param, "b_param"), (bias, "bias")):
        _chk(t_, _f32, nm)
    _chk(fhi, torch.float16, "fhi"); _chk(flo, torch.float16, "flo")
    Cc = b_shape.numel()
    if tuple(w_shape.shape) != (Cc, 148) or tuple(w_param.shape) != (Cc, 147) or fhi.numel() != Cc * 320 or flo.numel() != Cc * 320 or bias.numel() != Cc:
        raise ValueError("embed_pack_weights: shapes")
    check(_lib.load().pfpp_embed_pack_weights(_ptr(w_shape), _ptr(w_param), _ptr(b_shape), _ptr(b_param), _ptr(fhi), _ptr(flo), _ptr(bias), Cc,
                                              _stream()), "pfpp_embed_pack_weights")


def embed_tokens_packed(latent: torch.Tensor, xyz: torch.Tensor, scale: torch.Tensor, x: torch.Tensor, slot: Optional[torch.Tensor],
                        fhi: torch.Tensor, flo: torch.Tensor, bias: torch.Tensor, ref_emb: torch.Tensor, ref_u8: torch.Tensor, pe: torch.Tensor,
                        frag_pos: torch.Tensor, n: int, L: int) -> torch.Tensor:
    """the token embedding in one launch (pfpp_embed_tokens_small) on the planes of embed_pack_weights -> tok [n L, C]"""
    from ._lib import PwC

    for t_, nm in ((latent, "latent"), (xyz, "xyz"), (scale, "scale"), (x, "x"), (bias, "bias"), (ref_emb, "ref_emb"), (pe, "pe")):
        _chk(t_, _f32, nm)
    _chk(ref_u8, torch.uint8, "ref_part"); _chk(frag_pos, torch.int32, "frag_pos")
    _chk(fhi, torch.float16, "fhi"); _chk(flo, torch.float16, "flo")
    if slot is not None:
        _chk(slot, torch.int32, "slot")
    Cc = bias.numel()
    if fhi.numel() != Cc * 320 or flo.numel() != Cc * 320:
        raise ValueError("embed_tokens_packed: the planes do not belong to a [C, 320] weight")
    pw = PwC(0, 0, 0, 1.0, 320, fhi.data_ptr(), flo.data_ptr())
    tok = torch.empty((n * L, Cc), dtype=torch.float32, device=latent.device)
    check(_lib.load().pfpp_embed_tokens_small(_ptr(latent), _ptr(xyz), _ptr(scale), _ptr(x), _ptr(slot), C.byref(pw), _ptr(bias), _ptr(ref_emb),
                                              _ptr(ref_u8), _ptr(pe), _ptr(frag_pos), _ptr(tok), n, L, Cc, _stream()),
          "pfpp_embed_tokens_small")
    return tok


def token_features_t(latent: torch.Tensor, xyz: torch.Tensor, scale: torch.Tensor, x: torch.Tensor, slot: Optional[torch.Tensor],
                     ref_u8: torch.Tensor, n: int, L: int):
    """-> (ft_hi, ft_lo) [320, Mp] fp16: the extended token features of the n listed fragments, transposed (csrc/embed_train.hip);
    latent [slots, L, 64], xyz [slots, L, 3], scale [slots], x [slots, 7], ref_u8 [slots], slot [n] int32 or None"""
    for t_, nm in ((latent, "latent"), (xyz, "xyz"), (scale, "scale"), (x, "x")):
        _chk(t_, _f32, nm)
    _chk(ref_u8, torch.uint8, "ref_part")
    if slot is not None:
        _chk(slot, torch.int32, "slot")
    Mp = _lib.load().pfpp_token_features_t_cols(n, L)
    ft = torch.empty((2, 320, Mp), dtype=torch.float16, device=latent.device)
    check(_lib.load().pfpp_token_features_t(_ptr(latent), _ptr(xyz), _ptr(scale), _ptr(x), _ptr(slot),
                                            _ptr(ref_u8), _ptr(ft[0]), _ptr(ft[1]), n, L, _stream()), "pfpp_token_features_t")
    return ft[0], ft[1]


def token_embed_bwd(dtok: torch.Tensor, ft_hi: torch.Tensor, ft_lo: torch.Tensor, g_shape_w: torch.Tensor, g_shape_b: torch.Tensor,
                    g_param_w: torch.Tensor, g_param_b: torch.Tensor, g_ref_emb: torch.Tensor, n: int, L: int, *, g_scale: float = 1.0) -> None:
    """gradients of shape_embedding / param_fc / ref_part_emb accumulated from dtok [n L, C] in one launch (csrc/embed_train.hip)"""
    _chk(dtok, _f32, "dtok"); _chk(ft_hi, torch.float16, "ft_hi"); _chk(ft_lo, torch.float16, "ft_lo")
    Cc = dtok.shape[1]
    for t_, nm, shape in ((g_shape_w, "g_shape_w", (Cc, 148)), (g_shape_b, "g_shape_b", (Cc,)), (g_param_w, "g_param_w", (Cc, 147)),
                          (g_param_b, "g_param_b", (Cc,)), (g_ref_emb, "g_ref_emb", (2, Cc))):
        _chk(t_, _f32, nm)
        if tuple(t_.shape) != shape:
            raise ValueError(f"token_embed_bwd: {nm} has shape {tuple(t_.shape)}, expected {shape}")
    if dtok.shape[0] != n * L or ft_hi.shape != ft_lo.shape or ft_hi.shape[0] != 320 or ft_hi.shape[1] != _lib.load().pfpp_token_features_t_cols(n, L):
        raise ValueError("token_embed_bwd: dtok / feature planes do not belong to n fragments of L tokens")
    check(_lib.load().pfpp_token_embed_bwd(_ptr(dtok), _ptr(ft_hi), _ptr(ft_lo), _ptr(g_shape_w), _ptr(g_shape_b), _ptr(g_param_w),
                                           _ptr(g_param_b), _ptr(g_ref_emb), n, L, Cc, g_scale, _stream()), "pfpp_token_embed_bwd")


def ada_linear_bwd(dmods: torch.Tensor, se: torch.Tensor, w: torch.Tensor, g_w: torch.Tensor, g_b: torch.Tensor,
                   dse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """backward of the AdaLN modulation linears (csrc/ada_bwd.hip): g_w[j] += dmods[j]^T se[j], g_b[j] += column sums of dmods[j],
    -> dse[j] = dmods[j] w[j];  dmods [n, B, N2], se [n, B, C], w / g_w [n, N2, C], g_b [n, N2]"""
    for t_, nm in ((dmods, "dmods"), (se, "se"), (w, "w"), (g_w, "g_w"), (g_b, "g_b")):
        _chk(t_, _f32, nm)
    n, B, N2 = dmods.shape
    Cc = se.shape[2]
    if se.shape != (n, B, Cc) or w.shape != (n, N2, Cc) or g_w.shape != (n, N2, Cc) or g_b.numel() != n * N2:
        raise ValueError("ada_linear_bwd: shapes")
    if dse is None:
        dse = torch.empty_like(se)
    _chk(dse, _f32, "dse")
    scratch = torch.empty(_lib.load().pfpp_ada_linear_bwd_scratch_floats(n, N2), dtype=torch.float32, device=dmods.device)
    check(_lib.load().pfpp_ada_linear_bwd(_ptr(dmods), _ptr(se), _ptr(w), _ptr(g_w), _ptr(g_b), _ptr(dse), _ptr(scratch), n, B, Cc, N2,
                                          _stream()), "pfpp_ada_linear_bwd")
    return dse


def mse_loss_masked(pred: torch.Tensor, target: torch.Tensor, valid: torch.Tensor, ref_u8: torch.Tensor, grad_out: float = 1.0,
                    amax: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mse_loss over the rows with valid != 0 and ref == 0, the selection never materialised (pfpp_mse_loss_masked); amax [1]
    receives max |dpred|"""
    _chk(pred, _f32, "pred"); _chk(target, _f32, "target"); _chk(valid, _f32, "valid"); _chk(ref_u8, torch.uint8, "ref")
    n, width = pred.shape
    if valid.numel() != n or ref_u8.numel() != n:
        raise ValueError("mse_loss_masked: valid / ref must have one entry per row")
    loss = torch.empty((1,), dtype=_f32, device=pred.device)
    dpred = torch.empty_like(pred)
    check(_lib.load().pfpp_mse_loss_masked(_ptr(pred), _ptr(target), _ptr(valid), _ptr(ref_u8), _ptr(loss), _ptr(dpred), _ptr(amax), n,
                                           width, grad_out, _stream()), "pfpp_mse_loss_masked")
    return loss, dpred


def mse_loss(pred: torch.Tensor, target: torch.Tensor, sel_u8: torch.Tensor, grad_out: float = 1.0,
             need_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """-> (loss [1], dpred) over the rows with sel != 0 (Denoiser._loss)"""
    _chk(pred, _f32, "pred"); _chk(target, _f32, "target"); _chk(sel_u8, torch.uint8, "sel")
    n, width = pred.shape
    loss = torch.empty((1,), dtype=_f32, device=pred.device)
    dpred = torch.empty_like(pred) if need_grad else None
    check(_lib.load().pfpp_mse_loss(_ptr(pred), _ptr(target), _ptr(sel_u8), _ptr(loss), _ptr(dpred), n, width, grad_out,
                                    _stream()), "pfpp_mse_loss")
    return loss, dpred


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, beta1: float, beta2: float,
          eps: float, weight_decay: float, step: int, hi: Optional[torch.Tensor] = None,
          lo: Optional[torch.Tensor] = None, g_scale: float = 1.0, zero_grad: bool = False,
          overflow: Optional[torch.Tensor] = None) -> None:
    """one fused AdamW step over flat buffers (torch.optim.AdamW arithmetic); refreshes the split planes; zero_grad: clears g in
    the same pass; overflow (int32 [2] on the device: flag, count): elements with a non-finite gradient are skipped (per element, from
    that element's gradient alone: deterministic, identical on data-parallel replicas) and flagged (pfpp_adamw_guarded)"""
    if overflow is not None:
        _chk(overflow, torch.int32, "overflow")
        if overflow.numel() < 2:
            raise ValueError("adamw: overflow must hold [flag, count]")
    for t_, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t_, _f32, nm)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(_lib.load().pfpp_adamw_guarded(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(hi), _ptr(lo), p.numel(), lr, beta1, beta2, eps,
                                         weight_decay, bc1, bc2, g_scale, int(zero_grad), _ptr(overflow), _stream()), "pfpp_adamw_guarded")


def adamw_rows(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: torch.Tensor, *, mode: int, lr: float, beta1: float,
               beta2: float, eps: float, weight_decay: float, step: int, hi: Optional[torch.Tensor] = None,
               lo: Optional[torch.Tensor] = None, g_scale: float = 1.0, zero_grad: bool = False,
               overflow: Optional[torch.Tensor] = None) -> None:
    """adamw() over a stack of embedding tables p [n_tables, rows, C] restricted by rows (pfpp_adamw_rows): mode 0 = every row but
    the ones listed in t (int64 [n]), mode 1 = only the listed rows (each once); the two together are one adamw() over the stack"""
    for t_, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t_, _f32, nm)
    _chk(t, torch.int64, "t")
    n_tab, rows, Cc = p.shape
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(_lib.load().pfpp_adamw_rows(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(hi), _ptr(lo), n_tab, rows, Cc, _ptr(t), t.numel(), mode,
                                      lr, beta1, beta2, eps, weight_decay, bc1, bc2, g_scale, int(zero_grad), _ptr(overflow), _stream()),
          "pfpp_adamw_rows")


def adamw_rows_active(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, active: torch.Tensor, *, lr: float, beta1: float,
                      beta2: float, eps: float, weight_decay: float, step: int, hi: Optional[torch.Tensor] = None,
                      lo: Optional[torch.Tensor] = None, g_scale: float = 1.0, zero_grad: bool = False,
                      overflow: Optional[torch.Tensor] = None) -> None:
    """adamw() over a stack of embedding tables p [n_tables, rows, C] restricted to the rows marked in `active` (int32 bitmap kept by
    silu_embed_bwd(active=...)): rows that never received a gradient have zero moments and, with fl32(1 - lr * weight_decay) = 1, an
    update that is exactly the identity (pfpp_adamw_rows_active; every row is taken when the decay does not round to 1)"""
    for t_, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t_, _f32, nm)
    _chk(active, torch.int32, "active")
    n_tab, rows, Cc = p.shape
    if active.numel() * 32 < rows:
        raise ValueError("adamw_rows_active: the bitmap has fewer bits than the tables have rows")
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(_lib.load().pfpp_adamw_rows_active(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(hi), _ptr(lo), n_tab, rows, Cc, _ptr(active),
                                             lr, beta1, beta2, eps, weight_decay, bc1, bc2, g_scale, int(zero_grad), _ptr(overflow), _stream()),
          "pfpp_adamw_rows_active")


def bn_stats(x: torch.Tensor, running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
             momentum: float = 0.1) -> Tuple[torch.Tensor, torch.Tensor]:
    """per-column batch mean / biased variance of x [rows, C]; updates the running statistics in place like
    nn.BatchNorm2d in train mode (utils/pn2_utils.py:211-214)"""
    _chk(x, _f32, "x")
    rows, Cc = x.shape
    mean = torch.empty((Cc,), dtype=_f32, device=x.device)
    var = torch.empty((Cc,), dtype=_f32, device=x.device)
    lib = _lib.load()
    ws = torch.empty((int(lib.pfpp_bn_stats_workspace(rows, Cc)),), dtype=torch.uint8, device=x.device)
    if running_mean is not None:
        _chk(running_mean, _f32, "running_mean"); _chk(running_var, _f32, "running_var")
    check(lib.pfpp_bn_stats(_ptr(x), rows, Cc, x.stride(0), _ptr(mean), _ptr(var), _ptr(running_mean), _ptr(running_var),
                            momentum, _ptr(ws), _stream()), "pfpp_bn_stats")
    return mean, var


def bn_apply(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
             eps: float = 1e-5, pool: int = 0) -> torch.Tensor:
    """relu(batch-norm(x)) with the given statistics, optional max over groups of `pool` rows"""
    _chk(x, _f32, "x")
    for t_, nm in ((mean, "mean"), (var, "var"), (gamma, "gamma"), (beta, "beta")):
        _chk(t_, _f32, nm)
    rows, Cc = x.shape
    out = torch.empty((rows // pool if pool else rows, Cc), dtype=_f32, device=x.device)
    check(_lib.load().pfpp_bn_apply(_ptr(x), rows, Cc, x.stride(0), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), eps,
                                    _ptr(out), Cc, pool, _stream()), "pfpp_bn_apply")
    return out


BN_STAT_COPIES = 64      # the epilogues spread their fp64 atomics over this many accumulator copies


def bn_stats_buffer(C: int, device) -> torch.Tensor:
    return torch.zeros((BN_STAT_COPIES, 2, C), dtype=torch.float64, device=device)


def bn_finalize(stats: torch.Tensor, rows: int, gamma: torch.Tensor, beta: torch.Tensor,
                running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
                momentum: float = 0.1, eps: float = 1e-5, want_stats: bool = False):
    """stats accumulated by a GEMM epilogue -> (a_mul, a_add) for the consumer [+ (mean, var)]; updates the running
    buffers like nn.BatchNorm2d in train mode and clears `stats`"""
    _chk(stats, torch.float64, "stats"); _chk(gamma, _f32, "gamma"); _chk(beta, _f32, "beta")
    Cc = stats.shape[2]
    a_mul = torch.empty((Cc,), dtype=_f32, device=stats.device)
    a_add = torch.empty((Cc,), dtype=_f32, device=stats.device)
    mean = torch.empty((Cc,), dtype=_f32, device=stats.device) if want_stats else None
    var = torch.empty((Cc,), dtype=_f32, device=stats.device) if want_stats else None
    check(_lib.load().pfpp_bn_finalize(_ptr(stats), stats.shape[0], rows, Cc, _ptr(gamma), _ptr(beta), eps, momentum,
                                       _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(var), _ptr(a_mul), _ptr(a_add),
                                       _stream()), "pfpp_bn_finalize")
    return (a_mul, a_add, mean, var) if want_stats else (a_mul, a_add)


def bn_minmax_apply(mx: torch.Tensor, mn: torch.Tensor, a_mul: torch.Tensor, a_add: torch.Tensor) -> torch.Tensor:
    """max over a pool group of relu(a*x + b), from the group's max and min of x"""
    _chk(mx, _f32, "mx"); _chk(mn, _f32, "mn"); _chk(a_mul, _f32, "a_mul"); _chk(a_add, _f32, "a_add")
    rows, Cc = mx.shape
    out = torch.empty_like(mx)
    check(_lib.load().pfpp_bn_minmax_apply(_ptr(mx), _ptr(mn), _ptr(a_mul), _ptr(a_add), _ptr(out), rows, Cc, _stream()),
          "pfpp_bn_minmax_apply")
    return out


# ------------------------------------------------------------------------------------------------------------------
# plane-producing forms (include/pfpp.h): the result goes out as split-f16 planes for the plane GEMM (pfpp_hip.planes)
# ------------------------------------------------------------------------------------------------------------------
def geglu_planes(z: torch.Tensor, p: float, seed: int, site: int):
    """-> Planes of u = value * gelu(gate) (+ dropout) [rows, inner]; no fp32 copy"""
    from .planes import Planes, _pl

    _chk(z, _f32, "z")
    rows, two_inner = z.shape
    out = Planes.empty(rows, two_inner // 2, z.device)
    check(_lib.load().pfpp_geglu_p(_ptr(z), None, rows, two_inner // 2, p, seed, site, _pl(out), _stream()), "pfpp_geglu_p")
    return out


def geglu_bwd_planes(z: torch.Tensor, du: torch.Tensor, p: float, seed: int, site: int, scale: float):
    """-> Planes of scale * dz [rows, 2 inner]; no fp32 copy"""
    from .planes import Planes, _pl

    _chk(z, _f32, "z"); _chk(du, _f32, "du")
    rows, two_inner = z.shape
    out = Planes.empty(rows, two_inner, z.device, scale)
    check(_lib.load().pfpp_geglu_bwd_p(_ptr(z), _ptr(du), None, rows, two_inner // 2, p, seed, site, _pl(out), _stream()),
          "pfpp_geglu_bwd_p")
    return out


def dropout_layernorm_planes(y: torch.Tensor, res: Optional[torch.Tensor], p: float, seed: int, site: int, *,
                             mod: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                             beta: Optional[torch.Tensor] = None, group_batch: Optional[torch.Tensor] = None, group_rows: int = 1,
                             rows_per_batch: int = 1, eps: float = 1e-5):
    """dropout_layernorm with the normalised rows as Planes: -> (h written over y, Planes n)"""
    from .planes import Planes, _pl

    _chk(y, _f32, "y")
    rows, Cc = y.shape
    ld_mod = 0
    if res is not None:
        _chk(res, _f32, "res")
    if mod is not None:
        _chk(mod, _f32, "mod")
        ld_mod = mod.stride(0)
        if mod.shape[-1] != 2 * Cc:
            raise ValueError("dropout_layernorm: mod must be [B, 2C]")
    if group_batch is not None:
        _chk(group_batch, torch.int32, "group_batch")
        if group_batch.numel() * group_rows < rows:
            raise ValueError("dropout_layernorm: group_batch too short")
    n = Planes.empty(rows, Cc, y.device)
    check(_lib.load().pfpp_dropout_layernorm_p(_ptr(y), _ptr(res), _ptr(y), None, _ptr(mod), ld_mod, _ptr(gamma), _ptr(beta),
                                               _ptr(group_batch), group_rows, rows_per_batch, rows, Cc, eps, p, seed, site,
                                               _pl(n), _stream()), "pfpp_dropout_layernorm_p")
    return y, n


def layernorm_bwd_planes(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, scale: float, *, mod: Optional[torch.Tensor] = None,
                         gamma: Optional[torch.Tensor] = None, group_batch: Optional[torch.Tensor] = None,
                         group_rows: int = 32, rows_per_batch: int = 1, dmult: Optional[torch.Tensor] = None,
                         dadd: Optional[torch.Tensor] = None, ld_d: int = 0, eps: float = 1e-5,
                         drop: Optional[tuple] = None, want_ret: bool = True, want_dx: bool = False,
                         ret_fp32: bool = False):
    """layernorm_bwd (dx += ..., in place) that also emits Planes (scale * value) of what the backward chain continues with
    (`ret`: dropout(dx) for drop = (p, seed, site), else dx) and / or of the updated dx.  -> (ret_planes, dx_planes, ret_fp32_tensor)"""
    from .planes import Planes, _pl

    _chk(x, _f32, "x"); _chk(dy, _f32, "dy"); _chk(dx, _f32, "dx")
    rows, Cc = x.shape
    ld_mod = 0
    if mod is not None:
        _chk(mod, _f32, "mod")
        ld_mod = mod.stride(0)
    if group_batch is not None:
        _chk(group_batch, torch.int32, "group_batch")
        if group_batch.numel() * group_rows < rows:
            raise ValueError("layernorm_bwd: group_batch too short")
    p, seed, site = drop if drop is not None else (0.0, 0, 0)
    ret = Planes.empty(rows, Cc, x.device, scale) if want_ret else None
    dxp = Planes.empty(rows, Cc, x.device, scale) if want_dx else None
    out32 = torch.empty_like(dx) if (ret_fp32 and drop is not None) else None
    check(_lib.load().pfpp_layernorm_bwd_p(_ptr(x), _ptr(dy), _ptr(mod), ld_mod, _ptr(gamma), _ptr(group_batch), group_rows,
                                           rows_per_batch, _ptr(dx), _ptr(dmult), _ptr(dadd), ld_d, rows, Cc, eps, _ptr(out32),
                                           p, seed, site, int(drop is not None), _pl(ret), _pl(dxp), _stream()),
          "pfpp_layernorm_bwd_p")
    return ret, dxp, (out32 if drop is not None else dx) if ret_fp32 else None


def attn_dense_train_planes(qkv: torch.Tensor, seq_off: torch.Tensor, seq_len: torch.Tensor, max_len: int, H: int, dh: int,
                            scale: float):
    """attn_dense_train with the output both as fp32 (the backward's D = rowsum(dO . O)) and as Planes -> (out, Planes, lse)"""
    from .planes import Planes, _pl

    _chk(qkv, _f32, "qkv"); _chk(seq_off, torch.int32, "seq_off"); _chk(seq_len, torch.int32, "seq_len")
    rows = qkv.shape[0]
    out = torch.empty((rows, H * dh), dtype=_f32, device=qkv.device)
    outp = Planes.empty(rows, H * dh, qkv.device)
    lse = torch.empty((rows, H), dtype=_f32, device=qkv.device)
    check(_lib.load().pfpp_attn_dense_train_p(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(seq_off), _ptr(seq_len), None, 0,
                                              seq_off.numel(), max_len, H, dh, scale, _pl(outp), _stream()), "pfpp_attn_dense_train_p")
    return out, outp, lse


def attn_dense_bwd_planes(qkv, out_fwd, dout, lse, seq_off, seq_len, max_len: int, H: int, dh: int, scale: float, g_scale: float):
    """-> Planes of g_scale * dqkv (no fp32 copy)"""
    from .planes import Planes, _pl

    _chk(qkv, _f32, "qkv"); _chk(out_fwd, _f32, "out_fwd"); _chk(dout, _f32, "dout"); _chk(lse, _f32, "lse")
    dq = Planes.empty(qkv.shape[0], qkv.shape[1], qkv.device, g_scale)
    dvec = torch.empty_like(lse)
    check(_lib.load().pfpp_attn_dense_bwd_p(_ptr(qkv), _ptr(out_fwd), _ptr(dout), _ptr(lse), _ptr(dvec), None, _ptr(seq_off),
                                            _ptr(seq_len), None, 0, seq_off.numel(), max_len, H, dh, scale, _pl(dq), _stream()),
          "pfpp_attn_dense_bwd_p")
    return dq


def attn_blockdiag_bwd_planes(qkv, dout, n_frag: int, L: int, H: int, dh: int, scale: float, g_scale: float):
    from .planes import Planes, _pl

    _chk(qkv, _f32, "qkv"); _chk(dout, _f32, "dout")
    dq = Planes.empty(qkv.shape[0], qkv.shape[1], qkv.device, g_scale)
    check(_lib.load().pfpp_attn_blockdiag_bwd_p(_ptr(qkv), _ptr(dout), None, n_frag, L, H, dh, scale, _pl(dq), _stream()),
          "pfpp_attn_blockdiag_bwd_p")
    return dq


# ------------------------------------------------------------------------------------------------------------------
# the two output heads as one launch each way (csrc/heads.hip; denoiser_transformer.py:138-147)
# ------------------------------------------------------------------------------------------------------------------
def head_params(w0, w2, w4: torch.Tensor, b0: torch.Tensor, b2: torch.Tensor, b4: torch.Tensor, static: bool = False):
    """pfpp_head_params of one head: w0 / w2 = packing.PW (planes of scale * W), the last layer and the biases in fp32.
    static (eval: the weights do not change under the struct): also their fragment-blocked planes (PW.frag)"""
    from ._lib import HeadParams, PlanesC

    for t_, nm in ((w4, "w4"), (b0, "b0"), (b2, "b2"), (b4, "b4")):
        _chk(t_, _f32, nm)
    f0 = f2 = PlanesC(None, None, 0.0)
    if static:
        (h0, l0), (h2, l2) = w0.frag(), w2.frag()
        f0, f2 = PlanesC(h0.data_ptr(), l0.data_ptr(), w0.scale), PlanesC(h2.data_ptr(), l2.data_ptr(), w2.scale)
    return HeadParams(PlanesC(w0.hi.data_ptr(), w0.lo.data_ptr(), w0.scale), PlanesC(w2.hi.data_ptr(), w2.lo.data_ptr(), w2.scale),
                      w4.data_ptr(), b0.data_ptr(), b2.data_ptr(), b4.data_ptr(), f0, f2)


def heads_fwd(pooled: torch.Tensor, trans, rot, out: torch.Tensor, slot: Optional[torch.Tensor] = None, save: bool = False):
    """out[slot[r] or r, 0:3 | 3:7] = mlp_out_trans(pooled[r]) | mlp_out_rot(pooled[r]); trans / rot = head_params(...).
    save: also returns (a0, v0 [2, R, C], a1, v1 [2, R, C/2]) for heads_bwd"""
    _chk(pooled, _f32, "pooled"); _chk(out, _f32, "out")
    R, Cc = pooled.shape
    saved = (None, None, None, None)
    if save:
        saved = tuple(torch.empty((2, R, n), dtype=_f32, device=pooled.device) for n in (Cc, Cc, Cc // 2, Cc // 2))
    if slot is not None:
        _chk(slot, torch.int32, "slot")
    check(_lib.load().pfpp_heads_fwd(_ptr(pooled), C.byref(trans), C.byref(rot), R, Cc, *(_ptr(t_) for t_ in saved), _ptr(out),
                                     _ptr(slot), out.shape[-1], _stream()), "pfpp_heads_fwd")
    return saved if save else None


def heads_bwd(dout: torch.Tensor, trans, rot, saved, g_trans, g_rot, g_scale: float, L: int, want_dx: bool = True,
              slot: Optional[torch.Tensor] = None):
    """backward of heads_fwd from dout [R, 7] (or, with slot, from the rows slot[r] of dout [n_slots, 7]): -> (da0 [2, R, C],
    da1 [2, R, C/2], dx [R * L, C] or None); dW4 / db4 / db2 / db0 of both heads are accumulated into g_trans / g_rot (HeadGrads),
    the four wide weight gradients are the caller's (da0, da1 are their dY operands)"""
    _chk(dout, _f32, "dout")
    if slot is not None:
        _chk(slot, torch.int32, "slot")
    a0, v0, a1, v1 = saved
    _, R, Cc = a0.shape
    dev = dout.device
    da0 = torch.empty((2, R, Cc), dtype=_f32, device=dev)
    da1 = torch.empty((2, R, Cc // 2), dtype=_f32, device=dev)
    dp = torch.empty((2, R, Cc), dtype=_f32, device=dev)
    dx = torch.empty((R * L, Cc), dtype=_f32, device=dev) if want_dx else None
    check(_lib.load().pfpp_heads_bwd(_ptr(dout), _ptr(slot), C.byref(trans), C.byref(rot), R, Cc, _ptr(a0), _ptr(v0), _ptr(a1), _ptr(v1), _ptr(da0),
                                     _ptr(da1), _ptr(dp), C.byref(g_trans), C.byref(g_rot), g_scale, _ptr(dx), L, _stream()),
          "pfpp_heads_bwd")
    return da0, da1, (dx if want_dx else dp)
