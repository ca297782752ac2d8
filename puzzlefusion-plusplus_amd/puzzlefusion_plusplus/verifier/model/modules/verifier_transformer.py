"""VerifierTransformer (drop-in for verifier/model/modules/verifier_transformer.py), HIP-backed.

Owns a torch nn.TransformerEncoder purely as the parameter container (identical state_dict keys:
transformer_encoder.layers.{i}.self_attn.in_proj_weight, ...); forward runs pfpp_hip.verifier.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import TransformerEncoder, TransformerEncoderLayer

from pfpp_hip import verifier as hip_verifier
from pfpp_hip.packing import PackCache
from utils.model_utils import PositionalEncoding


class VerifierTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.model_channels = cfg.model.embed_dim
        self.num_layers = cfg.model.num_layers
        self.num_heads = cfg.model.num_heads
        C = self.model_channels
        layer = TransformerEncoderLayer(d_model=C, nhead=self.num_heads, dim_feedforward=2048, dropout=0.1,
                                        batch_first=True, activation="gelu")
        self.transformer_encoder = TransformerEncoder(layer, num_layers=self.num_layers, enable_nested_tensor=False)
        self.edge_indices_pe = PositionalEncoding(C // 2, max_len=20)
        self.edge_feature_emb = nn.Linear(7, C)
        self.mlp_out = nn.Linear(C, 1)
        self._cache = PackCache()

    def packed(self):
        live = dict(self.named_parameters())
        live.update(dict(self.named_buffers()))
        return self._cache.get(list(live.values()),
                               lambda: hip_verifier.pack_verifier({k: v.detach() for k, v in live.items()},
                                                                  self.num_layers))

    def forward(self, edge_features, edge_indices, mask):
        """edge_features [B,E,7], edge_indices i64 [B,E,2], mask [B,E] -> logits [B,E,1]"""
        if self.training:
            raise RuntimeError("VerifierTransformer (HIP): inference forward only; call .eval()")
        return hip_verifier.verifier_forward(self.packed(), edge_features, edge_indices, mask,
                                             num_layers=self.num_layers, num_heads=self.num_heads)
