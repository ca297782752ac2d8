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
        self._cache_train = PackCache()

    # ------------------------------------------------------------------ packed weights for the fused path
    def packed(self):
        live = dict(self.named_parameters())
        live.update(dict(self.named_buffers()))
        srcs = [v for k, v in live.items() if not k.endswith("num_batches_tracked") and ".fc" not in k]
        return self._cache.get(srcs, lambda: hip_encoder.pack_encoder({k: v.detach() for k, v in live.items()}))

    def packed_train(self):
        """train-mode operands: BatchNorm on batch statistics, the module's running-statistics buffers are
        updated in place by the kernels (the reference's frozen-but-.train() encoder, train_denoiser.py:33-35)"""
        live = dict(self.named_parameters())
        live.update(dict(self.named_buffers()))
        srcs = [v for k, v in live.items() if "running_" not in k and "num_batches" not in k and ".fc" not in k]
        pk = self._cache_train.get(srcs, lambda: hip_encoder.pack_encoder_train({k: v.detach() for k, v in live.items()}))
        self._cache._key = None          # the eval-mode packing folds the running statistics: stale after this call
        return pk

    def _encoder_grad_guard(self):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.pn2.parameters()):
            raise RuntimeError("VQVAE (HIP): the encoder is a frozen feature extractor on this path (gradients do not flow "
                               "into it); freeze its parameters (train_denoiser.py:33-35) or call under torch.no_grad()")

    def encode(self, part_pcs: torch.Tensor):
        """part_pcs [F,N,3] -> {"z_q": [F,L,num_dim], "xyz": [F,L,3]}  (vq_vae.py:52-68)"""
        if self.training:
            self._encoder_grad_guard()
            if part_pcs.shape[0] > 2048:
                raise ValueError("train-mode encode: batch statistics need all fragments in one pass (F <= 2048)")
            return hip_encoder.encode_valid(self.packed_train(), part_pcs.contiguous(), self.cfg.ae.num_point)
        return hip_encoder.encode_valid(self.packed(), part_pcs.contiguous(), self.cfg.ae.num_point)

    def extract_features(self, part_pcs, part_valids, pose, slot=None):
        """fused Denoiser._extract_features (denoiser.py:66-77): rotate by the current noisy
        quaternions, encode the valid fragments, scatter into zero-padded [B,P,L,*] tensors"""
        if slot is None:       # callers that keep part_valids fixed pass the precomputed list (no device->host read)
            from pfpp_hip.denoiser import layout_of

            slot = layout_of(part_valids, self.cfg.ae.num_point).slot32
        if self.training:
            self._encoder_grad_guard()
            if slot.numel() > 2048:
                raise ValueError("train-mode encode: batch statistics need all fragments in one pass (F <= 2048)")
            pk = self.packed_train()
        else:
            pk = self.packed()
        return hip_encoder.extract_features(pk, part_pcs.contiguous(), pose.contiguous(), slot, self.cfg.ae.num_point)

    def decode(self, z_q):
        raise NotImplementedError("decoder = VQ-VAE pre-training, out of scope of the HIP path")
