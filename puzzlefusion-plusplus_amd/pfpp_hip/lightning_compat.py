"""LightningModule if lightning is installed, else a minimal stand-in with the hooks the shells use
(self.log, self.device, save_hyperparameters) so the public surface does not change when the
optional host dependency is absent (it is absent in the build image; SURVEY.md §7 hard parts)."""
from __future__ import annotations

import torch
import torch.nn as nn

try:  # pragma: no cover - depends on the host environment
    import lightning.pytorch as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}

        @property
        def device(self) -> torch.device:
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def save_hyperparameters(self, *a, **k):
            return None

        def log(self, name, value, **kwargs):
            self.logged[name] = value


def instantiate(node, *args):
    """hydra.utils.instantiate for the `_target_:` nodes the reference configs use; falls back to
    an importlib lookup when hydra is absent."""
    try:  # pragma: no cover
        import hydra

        return hydra.utils.instantiate(node, *args)
    except Exception:  # noqa: BLE001
        import importlib

        target = getattr(node, "_target_", None) or node["_target_"]
        mod, _, name = target.rpartition(".")
        return getattr(importlib.import_module(mod), name)(*args)
