"""TEST INFRASTRUCTURE — deterministic closed-form weights for the parity tests.

w[i] = f(name, i): a 32-bit integer hash of (crc32(name), flat index) mapped to [-1, 1) and scaled
like a fan-in initialiser, so the 57.6 M denoiser parameters never have to be stored; the same
filler is applied to the imported reference modules (tools/make_goldens.py), to the oracle and to
the HIP-backed modules.  Shapes/keys: SURVEY.md §8b.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np
import torch


def _hash_uniform(name: str, n: int) -> np.ndarray:
    """n values in [-1, 1): murmur3 finaliser over (index + seed)"""
    seed = np.uint32(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x85EBCA6B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x.astype(np.uint64) * np.uint64(0xC2B2AE35) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return (x.astype(np.float64) / 2147483648.0 - 1.0).astype(np.float32)


def fill(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(_hash_uniform(name, n) * np.float32(scale) + np.float32(offset)).reshape(tuple(shape))


def _linear(sd, name, out_f, in_f, bias=True, gain=1.0, extra_shape=()):
    s = gain * math.sqrt(3.0 / in_f)
    sd[f"{name}.weight"] = fill(f"{name}.weight", (out_f, in_f) + tuple(extra_shape), s)
    if bias:
        sd[f"{name}.bias"] = fill(f"{name}.bias", (out_f,), 0.1)


def vqvae_state_dict(num_point_dim: int = 64, n_embeddings: int = 1024, embedding_dim: int = 16,
                     local_decode_pts: int = 40) -> Dict[str, torch.Tensor]:
    """keys of puzzlefusion_plusplus.denoiser.model.modules.encoder.VQVAE (72 entries)"""
    sd: Dict[str, torch.Tensor] = {}
    specs = (("sa1", 3, (64, 64, 128)), ("sa2", 131, (128, 128, 256)), ("sa3", 259, (256, 256, 512)))
    for name, cin, mlp in specs:
        last = cin
        for i, co in enumerate(mlp):
            p = f"pn2.{name}"
            _linear(sd, f"{p}.mlp_convs.{i}", co, last, gain=1.6, extra_shape=(1, 1))
            sd[f"{p}.mlp_bns.{i}.weight"] = fill(f"{p}.mlp_bns.{i}.weight", (co,), 0.2, 1.0)
            sd[f"{p}.mlp_bns.{i}.bias"] = fill(f"{p}.mlp_bns.{i}.bias", (co,), 0.1)
            sd[f"{p}.mlp_bns.{i}.running_mean"] = fill(f"{p}.mlp_bns.{i}.running_mean", (co,), 0.1)
            sd[f"{p}.mlp_bns.{i}.running_var"] = fill(f"{p}.mlp_bns.{i}.running_var", (co,), 0.4, 1.0)
            sd[f"{p}.mlp_bns.{i}.num_batches_tracked"] = torch.tensor(100, dtype=torch.long)
            last = co
    _linear(sd, "pn2.conv6", num_point_dim, 512, extra_shape=(1,))
    _linear(sd, "pn2.fc1", 256, num_point_dim)
    _linear(sd, "pn2.fc2", 512, 256)
    _linear(sd, "pn2.fc3", local_decode_pts * 3, 512)
    # a codebook on the scale of the latents (a trained one is; the default init U(-1/K, 1/K) makes
    # every argmin a rounding coin-flip and would pin nothing)
    sd["vector_quantization.embedding.weight"] = fill("vector_quantization.embedding.weight",
                                                      (n_embeddings, embedding_dim), 1.0)
    return sd


def denoiser_state_dict(embed_dim: int = 512, num_layers: int = 6, max_len: int = 20) -> Dict[str, torch.Tensor]:
    """keys of DenoiserTransformer (57,618,183 parameters at the reference size)"""
    from .pfpp_oracle import positional_table

    C = embed_dim
    sd: Dict[str, torch.Tensor] = {}
    sd["ref_part_emb.weight"] = fill("ref_part_emb.weight", (2, C), 0.5)
    for i in range(num_layers):
        p = f"transformer_layers.{i}"
        for n in ("norm1", "norm2"):
            sd[f"{p}.{n}.emb.weight"] = fill(f"{p}.{n}.emb.weight", (6 * C, C), 1.0)
            _linear(sd, f"{p}.{n}.linear", 2 * C, C, gain=0.5)
        for a in ("self_attn", "global_attn"):
            for proj in ("to_q", "to_k", "to_v"):
                _linear(sd, f"{p}.{a}.{proj}", C, C, bias=False, gain=1.5)
            _linear(sd, f"{p}.{a}.to_out.0", C, C)
        sd[f"{p}.norm3.weight"] = fill(f"{p}.norm3.weight", (C,), 0.2, 1.0)
        sd[f"{p}.norm3.bias"] = fill(f"{p}.norm3.bias", (C,), 0.1)
        _linear(sd, f"{p}.ff.net.0.proj", 8 * C, C)
        _linear(sd, f"{p}.ff.net.2", C, 4 * C)
    _linear(sd, "shape_embedding", C, 148)
    _linear(sd, "param_fc", C, 147)
    sd["pos_encoding.pe"] = positional_table(C, max_len)
    for h, o in (("mlp_out_trans", 3), ("mlp_out_rot", 4)):
        _linear(sd, f"{h}.0", C, C)
        _linear(sd, f"{h}.2", C // 2, C)
        _linear(sd, f"{h}.4", o, C // 2)
    return sd


def verifier_state_dict(embed_dim: int = 256, num_layers: int = 6, ff: int = 2048, max_len: int = 20):
    """keys of VerifierTransformer (7,892,737 parameters at the reference size)"""
    from .pfpp_oracle import positional_table

    C = embed_dim
    sd: Dict[str, torch.Tensor] = {}
    for i in range(num_layers):
        p = f"transformer_encoder.layers.{i}"
        sd[f"{p}.self_attn.in_proj_weight"] = fill(f"{p}.self_attn.in_proj_weight", (3 * C, C), 1.5 * math.sqrt(3.0 / C))
        sd[f"{p}.self_attn.in_proj_bias"] = fill(f"{p}.self_attn.in_proj_bias", (3 * C,), 0.1)
        _linear(sd, f"{p}.self_attn.out_proj", C, C)
        _linear(sd, f"{p}.linear1", ff, C)
        _linear(sd, f"{p}.linear2", C, ff)
        for n in ("norm1", "norm2"):
            sd[f"{p}.{n}.weight"] = fill(f"{p}.{n}.weight", (C,), 0.2, 1.0)
            sd[f"{p}.{n}.bias"] = fill(f"{p}.{n}.bias", (C,), 0.1)
    sd["edge_indices_pe.pe"] = positional_table(C // 2, max_len)
    _linear(sd, "edge_feature_emb", C, 7)
    _linear(sd, "mlp_out", 1, C)
    return sd
