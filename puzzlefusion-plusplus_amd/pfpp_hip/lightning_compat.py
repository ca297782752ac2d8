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

        def log(self, name, value, sync_dist: bool = False, **kwargs):
            if sync_dist and torch.is_tensor(value):
                import torch.distributed as dist

                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    value = value.detach().clone().float()
                    dist.all_reduce(value)                       # Lightning's sync_dist: the mean over ranks
                    value /= dist.get_world_size()
            self.logged[name] = value


def instantiate(node, *args):
    """hydra.utils.instantiate for the `_target_:` nodes the reference configs use; falls back to
    an importlib lookup when hydra is absent."""
    try:  # pragma: no cover
        import hydra
    except ImportError:
        hydra = None
    if hydra is not None:  # pragma: no cover
        return hydra.utils.instantiate(node, *args)         # constructor / config errors propagate as they are
    import importlib

    get = node.get if hasattr(node, "get") else (lambda k, d=None: getattr(node, k, d))
    target = get("_target_")
    if target is None:
        raise ValueError("instantiate: the config node has no _target_")
    keys = list(node.keys()) if hasattr(node, "keys") else [k for k in vars(node)]
    kwargs = {k: get(k) for k in keys if not str(k).startswith("_")}
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(*args, **kwargs)
