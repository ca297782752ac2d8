"""torch.optim front end of the fused AdamW kernel (Denoiser.configure_optimizers, denoiser.py:230-241).

The optimizer state lives in the flat buffers of pfpp_hip.train.FlatParams; `state[p]` exposes per-parameter
views of them with torch.optim.AdamW's keys (step / exp_avg / exp_avg_sq), so optimizer state_dicts are
interchangeable with the reference's checkpoints."""
from __future__ import annotations

import torch


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, engine, lr: float = 2e-4, betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6):
        self.engine = engine
        flat = engine.flat
        params = [flat.named[n] for n in flat.order]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._bind_state()

    def _bind_state(self) -> None:
        flat = self.engine.flat
        for n in flat.order:
            p = flat.named[n]
            st = self.state[p]
            st["step"] = torch.tensor(float(self.engine.step_count))
            st["exp_avg"] = flat.view(flat.exp_avg, n)
            st["exp_avg_sq"] = flat.view(flat.exp_avg_sq, n)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        self.engine.optimizer_step(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"])
        for st in self.state.values():
            st["step"] += 1
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        """gradients stay views of the flat buffer (set_to_none would detach them from the kernels)"""
        self.engine.flat.zero_grad()
        self.engine.flat.attach_grads()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        flat = self.engine.flat
        steps = []
        with torch.no_grad():
            for n in flat.order:
                st = self.state[flat.named[n]]
                for key, buf in (("exp_avg", flat.exp_avg), ("exp_avg_sq", flat.exp_avg_sq)):
                    view = flat.view(buf, n)
                    if st[key].data_ptr() != view.data_ptr():
                        view.copy_(st[key])
                        st[key] = view
                steps.append(int(st["step"]))
        self.engine.step_count = max(steps) if steps else 0
