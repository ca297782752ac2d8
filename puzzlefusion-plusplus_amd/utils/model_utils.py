"""Drop-in for the reference's utils/model_utils.py (state_dict-compatible).

PositionalEncoding keeps the `pe` buffer [1, max_len, d_model] that published checkpoints carry
(utils/model_utils.py:5-21); on the HIP path the table is added inside pfpp_token_combine /
pfpp_verifier_embed, so `forward` here only serves callers that use the module directly.
EmbedderNerf describes the NeRF feature layout (utils/model_utils.py:39-69); the features
themselves are produced by pfpp_token_features.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 20):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        table = torch.zeros(max_len, d_model)
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        inv = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        table[:, 0::2] = torch.sin(pos * inv)
        table[:, 1::2] = torch.cos(pos * inv)
        self.register_buffer("pe", table[None])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, P, L, C]: adds pe over the fragment axis (P must equal max_len, as in the reference)"""
        return self.dropout(x + self.pe.unsqueeze(2))


class EmbedderNerf:
    """layout descriptor: [x | sin(2^0 x) | cos(2^0 x) | ... | sin(2^(m-1) x) | cos(2^(m-1) x)]"""

    def __init__(self, input_dims: int, num_freqs: int = 10, max_freq_log2: int = 9, include_input: bool = True,
                 log_sampling: bool = True, periodic_fns=(torch.sin, torch.cos)):
        if not (include_input and log_sampling and max_freq_log2 == num_freqs - 1 and len(periodic_fns) == 2):
            raise ValueError("only the reference's embedding (include_input, log-sampled 2^0..2^(m-1), sin+cos) "
                             "is implemented by the HIP token kernel")
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.out_dim = input_dims * (1 + 2 * num_freqs)

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        """host/torch evaluation for callers outside the fused path (any device)"""
        parts = [inputs]
        for k in range(self.num_freqs):
            parts += [torch.sin(inputs * float(2 ** k)), torch.cos(inputs * float(2 ** k))]
        return torch.cat(parts, dim=-1)
