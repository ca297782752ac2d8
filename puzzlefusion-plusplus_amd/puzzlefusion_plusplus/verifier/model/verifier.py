"""Verifier shell (drop-in for puzzlefusion_plusplus/verifier/model/verifier.py), inference surface.

`forward(data_dict) -> {"logits"}` and `_loss` (the weighted BCE of verifier.py:20-47: negatives weighted 0.2) run on the
HIP-backed VerifierTransformer.  Training the verifier (verifier.py:49-69, train_verifier.py) is outside the hot path
(SURVEY.md §8a lists a18 = VerifierTransformer.forward only): the HIP VerifierTransformer has no backward, so this shell
does not pretend to be trainable — `training_step` / `configure_optimizers` raise with that explanation instead of
returning a loss that carries no graph.  Published verifier checkpoints load unchanged (same state_dict keys)."""
from __future__ import annotations

import torch
from torch.nn import functional as F

from pfpp_hip.lightning_compat import LightningModule
from puzzlefusion_plusplus.verifier.model.modules.verifier_transformer import VerifierTransformer

_NO_TRAINING = ("Verifier: training is not part of the MI355X hot path (the HIP VerifierTransformer is forward-only); train the "
                "verifier with the reference implementation and load the checkpoint here")


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
        """evaluation-time loss value (no graph): weighted BCE over the valid edges, verifier.py:20-47"""
        mask = data_dict["edge_valids"].bool()
        logits = output_dict["logits"].squeeze(-1)[mask]
        target = data_dict["cls_gt"].float()[mask]
        weight = torch.where(target > 0.5, torch.ones_like(target), torch.full_like(target, self.neg_weight))
        return {"bce_loss": F.binary_cross_entropy_with_logits(logits, target, weight=weight)}

    def validation_step(self, data_dict, idx):
        with torch.no_grad():
            loss = self._loss(data_dict, self(data_dict))["bce_loss"]
        self.log("val_loss/bce_loss", loss, on_step=False, on_epoch=True)
        return loss

    def training_step(self, data_dict, idx):
        raise NotImplementedError(_NO_TRAINING)

    def configure_optimizers(self):
        raise NotImplementedError(_NO_TRAINING)
