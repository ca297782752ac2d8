"""Verifier shell (drop-in for puzzlefusion_plusplus/verifier/model/verifier.py): weighted BCE on the
edge logits of the HIP-backed VerifierTransformer (negatives weighted 0.2, verifier.py:27)."""
from __future__ import annotations

import torch
from torch.nn import functional as F

from pfpp_hip.lightning_compat import LightningModule
from puzzlefusion_plusplus.verifier.model.modules.verifier_transformer import VerifierTransformer


class Verifier(LightningModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.verifier = VerifierTransformer(cfg)
        self.save_hyperparameters()
        self.neg_weight = 0.2

    def forward(self, data_dict):
        logits = self.verifier(data_dict["edge_features"], data_dict["edge_indices"], data_dict["edge_valids"])
        return {"logits": logits}

    def _loss(self, data_dict, output_dict):
        mask = data_dict["edge_valids"].bool()
        logits = output_dict["logits"].squeeze(-1)[mask]
        target = data_dict["cls_gt"].float()[mask]
        weight = torch.where(target > 0.5, torch.ones_like(target), torch.full_like(target, self.neg_weight))
        return {"bce_loss": F.binary_cross_entropy_with_logits(logits, target, weight=weight)}

    def training_step(self, data_dict, idx):
        loss = self._loss(data_dict, self(data_dict))["bce_loss"]
        self.log("train_loss/bce_loss", loss, on_step=True, on_epoch=False)
        return loss

    def configure_optimizers(self):
        return torch.optim.AdamW(self.parameters(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-08)
