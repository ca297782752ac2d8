"""Codebook lookup (drop-in for vqvae/model/modules/quantizer.py), HIP-backed."""
from __future__ import annotations

import torch
import torch.nn as nn

from pfpp_hip import ops


class VectorQuantizer(nn.Module):
    """n_e codes of width e_dim; forward maps every e_dim-wide sub-vector of z to its nearest code
    (first minimum of |z|^2 + |e|^2 - 2 z.e) and returns the straight-through value z + (e - z)
    (quantizer.py:45-63).  Returns the reference's 5-tuple; the pre-training-only entries
    (loss, perplexity, one-hot encodings) are None on this inference path."""

    def __init__(self, n_e: int, e_dim: int, beta: float):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z: torch.Tensor):
        zc = z.contiguous()
        rows = zc.shape[0]
        slot = torch.arange(rows, dtype=torch.int32, device=z.device)
        z_q, codes = ops.vq_encode(zc.view(rows, -1, zc.shape[-1]), self.embedding.weight.detach().contiguous(),
                                   slot, rows, return_codes=True)
        return None, z_q.view(z.shape), None, None, codes.long().reshape(-1, 1)
