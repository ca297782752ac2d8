"""PointNet++ fragment encoder (drop-in for vqvae/model/modules/pn2.py), HIP-backed.

Keeps the reference's parameter tree: sa1/sa2/sa3 (PointNetSetAbstraction), conv6 and the decoder
linears fc1-fc3 (present so VQ-VAE checkpoints load with strict=True; the decoder and the Chamfer
loss belong to stage-1 pre-training and are not part of this path).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from pfpp_hip import ops
from utils.pn2_utils import PointNetSetAbstraction


class PN2(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_point = cfg.ae.num_point
        self.num_dim = cfg.ae.num_dim
        self.local_decode_pts = cfg.ae.local_decode_pts
        # pn2.py:16-18
        self.sa1 = PointNetSetAbstraction(256, 0.2, 32, 3, [64, 64, 128], False)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = PointNetSetAbstraction(self.num_point, 0.8, 64, 256 + 3, [256, 256, 512], False)
        self.conv6 = nn.Conv1d(512, self.num_dim, kernel_size=1)
        self.fc1 = nn.Linear(self.num_dim, 256)
        self.fc2 = nn.Linear(256, 512)
        self.fc3 = nn.Linear(512, self.local_decode_pts * 3)

    def encode_channels_last(self, pts: torch.Tensor):
        """pts [F,N,3] -> z_e [F,L,num_dim], xyz [F,L,3]"""
        xyz, feats = pts.contiguous(), None
        for sa in (self.sa1, self.sa2, self.sa3):
            xyz, feats = sa.forward_channels_last(xyz, feats)
        F, L, C = feats.shape
        w = self.conv6.weight.detach().reshape(self.num_dim, C)
        z = ops.linear(feats.view(F * L, C), w.contiguous(), self.conv6.bias.detach().contiguous())
        return z.view(F, L, self.num_dim), xyz

    def encode(self, xyz: torch.Tensor):
        """xyz [F,3,N] (channel-first, as the reference passes it) -> (z_e [F,L,C], xyz [F,L,3]) (pn2.py:57-68)"""
        return self.encode_channels_last(xyz.permute(0, 2, 1))

    def decode(self, global_feat):
        raise NotImplementedError("PN2.decode belongs to VQ-VAE pre-training (out of scope of the HIP path)")

    def forward(self, data_dict):
        raise NotImplementedError("PN2.forward (reconstruction) belongs to VQ-VAE pre-training; use encode()")
