"""VQ-VAE fragment encoder (drop-in for vqvae/model/modules/vq_vae.py and its duplicate
denoiser/model/modules/encoder.py), HIP-backed.  state_dict keys: pn2.*, vector_quantization.*"""
from __future__ import annotations

import torch
import torch.nn as nn

from pfpp_hip import encoder as hip_encoder
from pfpp_hip.packing import PackCache
from puzzlefusion_plusplus.vqvae.model.modules.pn2 import PN2
from puzzlefusion_plusplus.vqvae.model.modules.quantizer import VectorQuantizer


class VQVAE(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pn2 = PN2(cfg)
        self.encoder = self.pn2.encode
        self.vector_quantization = VectorQuantizer(cfg.ae.n_embeddings, cfg.ae.embedding_dim, cfg.ae.beta)
        self._cache = PackCache()

    # ------------------------------------------------------------------ packed weights for the fused path
    def packed(self):
        live = dict(self.named_parameters())
        live.update(dict(self.named_buffers()))
        srcs = [v for k, v in live.items() if not k.endswith("num_batches_tracked") and ".fc" not in k]
        return self._cache.get(srcs, lambda: hip_encoder.pack_encoder({k: v.detach() for k, v in live.items()}))

    def encode(self, part_pcs: torch.Tensor):
        """part_pcs [F,N,3] -> {"z_q": [F,L,num_dim], "xyz": [F,L,3]}  (vq_vae.py:52-68)"""
        if self.training:
            raise RuntimeError("VQVAE.encode (HIP) runs the frozen encoder in eval mode; call .eval()")
        return hip_encoder.encode_valid(self.packed(), part_pcs.contiguous(), self.cfg.ae.num_point)

    def extract_features(self, part_pcs, part_valids, pose):
        """fused Denoiser._extract_features (denoiser.py:66-77): rotate by the current noisy
        quaternions, encode the valid fragments, scatter into zero-padded [B,P,L,*] tensors"""
        if self.training:
            raise RuntimeError("VQVAE.extract_features (HIP) needs eval mode")
        slot = torch.nonzero(part_valids.reshape(-1).bool()).flatten().to(torch.int32)
        return hip_encoder.extract_features(self.packed(), part_pcs.contiguous(), pose.contiguous(), slot,
                                            self.cfg.ae.num_point)

    def decode(self, z_q):
        raise NotImplementedError("decoder = VQ-VAE pre-training, out of scope of the HIP path")
