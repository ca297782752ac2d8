"""DenoiserTransformer (drop-in for denoiser/model/modules/denoiser_transformer.py), HIP-backed.

Same constructor (cfg.model.{embed_dim,out_channels,num_layers,num_heads,num_dim,...}), same
forward signature and the same state_dict keys/shapes; the forward pass is pfpp_hip.denoiser.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from pfpp_hip import denoiser as hip_denoiser
from pfpp_hip.packing import PackCache
from puzzlefusion_plusplus.denoiser.model.modules.attention import EncoderLayer
from utils.model_utils import EmbedderNerf, PositionalEncoding


class _TrainFn(torch.autograd.Function):
    """DenoiserTransformer.forward in train mode as one autograd node: backward() runs the HIP backward and
    accumulates straight into the parameters' .grad (views of the engine's flat gradient buffer)"""

    @staticmethod
    def forward(ctx, eng, seed, x, timesteps, latent, xyz, part_valids, scale, ref_part, grad_anchor):
        # grad_anchor: any parameter that requires grad, so that autograd records this node
        pred, saved = eng.forward(x, timesteps, latent, xyz, part_valids, scale, ref_part, seed=seed, train=True)
        ctx.eng, ctx.saved = eng, saved
        return pred

    @staticmethod
    def backward(ctx, dpred):
        ctx.eng.backward(ctx.saved, dpred.contiguous())
        ctx.saved = None
        return (None,) * 10


class DenoiserTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        m = cfg.model
        self.model_channels = m.embed_dim
        self.out_channels = m.out_channels
        self.num_layers = m.num_layers
        self.num_heads = m.num_heads
        C = self.model_channels
        self.ref_part_emb = nn.Embedding(2, C)
        self.activation = nn.SiLU()
        self.transformer_layers = nn.ModuleList([
            EncoderLayer(dim=C, num_attention_heads=self.num_heads, attention_head_dim=C // self.num_heads,
                         dropout=0.2, activation_fn="geglu", num_embeds_ada_norm=6 * C, attention_bias=False,
                         norm_elementwise_affine=True, final_dropout=False)
            for _ in range(self.num_layers)
        ])
        pose_pe, pos_pe, scale_pe = EmbedderNerf(7), EmbedderNerf(3), EmbedderNerf(1)   # multires fixed at 10
        self.param_embedding, self.pos_embedding, self.scale_embedding = pose_pe.embed, pos_pe.embed, scale_pe.embed
        self.shape_embedding = nn.Linear(m.num_dim + scale_pe.out_dim + pos_pe.out_dim, C)
        self.param_fc = nn.Linear(pose_pe.out_dim, C)
        self.pos_encoding = PositionalEncoding(C, max_len=getattr(m, "max_len", 20))
        self.mlp_out_trans = nn.Sequential(nn.Linear(C, C), nn.SiLU(), nn.Linear(C, C // 2), nn.SiLU(),
                                           nn.Linear(C // 2, 3))
        self.mlp_out_rot = nn.Sequential(nn.Linear(C, C), nn.SiLU(), nn.Linear(C, C // 2), nn.SiLU(),
                                         nn.Linear(C // 2, 4))
        self._cache = PackCache()
        # False (default): every one of the P slots is evaluated, exactly like the reference.
        # True: padded fragment slots are dropped (their don't-care outputs become 0; the outputs of
        # valid fragments are unchanged) — see pfpp_hip.denoiser.denoiser_forward_compact.
        self.compact_padded = False

    def packed(self):
        live = dict(self.named_parameters())
        live.update(dict(self.named_buffers()))
        return self._cache.get(list(live.values()),
                               lambda: hip_denoiser.pack_denoiser({k: v.detach() for k, v in live.items()},
                                                                  self.num_layers))

    def train_engine(self):
        """the training engine (pfpp_hip.train.DenoiserTrainEngine); created on first use: from then on the
        parameters are views of one flat buffer (same names, shapes and values)"""
        if getattr(self, "_engine", None) is None:
            from pfpp_hip.train import DenoiserTrainEngine

            object.__setattr__(self, "_engine", DenoiserTrainEngine(self))
        return self._engine

    def forward(self, x, timesteps, latent, xyz, part_valids, scale, ref_part, layout=None):
        """x [B,P,7], timesteps i64 [B], latent [B,P,L,64], xyz [B,P,L,3], part_valids [B,P],
        scale [B,P,1], ref_part bool [B,P] -> predicted noise [B,P,7] (trans 3 | rot 4)"""
        if self.training and torch.is_grad_enabled():
            # train mode (dropouts active) with gradients: the fused training engine behind autograd
            eng = self.train_engine()
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # torch.manual_seed governs the dropout masks
            return _TrainFn.apply(eng, seed, x, timesteps, latent, xyz, part_valids, scale, ref_part, self.ref_part_emb.weight)
        if self.training:
            pred, _ = self.train_engine().forward(x, timesteps, latent, xyz, part_valids, scale, ref_part,
                                                  seed=int(torch.randint(0, 2 ** 62, (1,)).item()), train=True)
            return pred
        if self.compact_padded or layout is not None:
            # `layout` (pfpp_hip.denoiser.CompactLayout, built once while part_valids is unchanged) makes the call
            # free of device->host reads: required inside HIP-graph capture
            return hip_denoiser.denoiser_forward_compact(self.packed(), x.float(), timesteps, latent, xyz, part_valids, scale,
                                                         ref_part, num_layers=self.num_layers, num_heads=self.num_heads,
                                                         layout=layout)
        return hip_denoiser.denoiser_forward(self.packed(), x.float(), timesteps, latent, xyz, part_valids, scale, ref_part,
                                             num_layers=self.num_layers, num_heads=self.num_heads)
