"""Parameter containers of one denoiser layer (drop-in for denoiser/model/modules/attention.py).

The reference composes diffusers-0.21.4 `Attention` / `FeedForward`; diffusers is not a dependency
here, so `Attention` and `FeedForward` below are parameter holders with the same state_dict keys
(to_q/to_k/to_v without bias, to_out.0 with bias; net.0.proj = GEGLU projection, net.2 = output
linear).  The arithmetic of a layer lives in pfpp_hip.denoiser (fused AdaLN / attention / GEGLU
kernels); these classes only own the weights.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class MyAdaLayerNorm(nn.Module):
    """emb: Embedding(num_embeddings, dim) -> SiLU -> linear: Linear(dim, 2*dim) = (scale, shift);
    y = LN(x) * (1 + scale) + shift   (attention.py:5-25)"""

    def __init__(self, embedding_dim: int, num_embeddings: int):
        super().__init__()
        self.emb = nn.Embedding(num_embeddings, embedding_dim)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, embedding_dim * 2)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False)


class Attention(nn.Module):
    def __init__(self, query_dim: int, heads: int = 8, dim_head: int = 64, dropout: float = 0.0, bias: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])


class _GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim: int, dropout: float = 0.0, activation_fn: str = "geglu", final_dropout: bool = False,
                 mult: int = 4):
        super().__init__()
        if activation_fn != "geglu":
            raise ValueError("only the GEGLU feed-forward of the reference is implemented")
        inner = dim * mult
        self.net = nn.ModuleList([_GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim)])


class EncoderLayer(nn.Module):
    """norm1 -> self_attn (block-diagonal) -> +res; norm2 -> global_attn (key-padding) -> +res;
    norm3 -> ff (GEGLU) -> +res   (attention.py:29-91)"""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout: float = 0.0,
                 activation_fn: str = "geglu", num_embeds_ada_norm: int = None, attention_bias: bool = False,
                 norm_elementwise_affine: bool = True, final_dropout: bool = False):
        super().__init__()
        if attention_bias:
            raise ValueError("attention_bias=True is not used by the reference (denoiser_transformer.py:34)")
        self.norm1 = MyAdaLayerNorm(dim, num_embeds_ada_norm)
        self.self_attn = Attention(dim, num_attention_heads, attention_head_dim, dropout, attention_bias)
        self.norm2 = MyAdaLayerNorm(dim, num_embeds_ada_norm)
        self.global_attn = Attention(dim, num_attention_heads, attention_head_dim, dropout, attention_bias)
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)
