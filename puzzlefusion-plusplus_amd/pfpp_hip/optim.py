"""torch.optim front end of the fused AdamW kernel (Denoiser.configure_optimizers, denoiser.py:230-241).

The optimizer state lives in the flat buffers of pfpp_hip.train.FlatParams; `state[p]` exposes per-parameter
views of them with torch.optim.AdamW's keys (step / exp_avg / exp_avg_sq).

Checkpoint compatibility.  The reference builds `torch.optim.AdamW(self.parameters())` over the WHOLE Denoiser module
(denoiser.py:230-237): its param group lists the DenoiserTransformer's parameters in registration order followed by the
frozen encoder's (which never get state), and torch's Optimizer.load_state_dict maps state BY POSITION.  Pass the same
list as `params` (Denoiser.configure_optimizers does) and the group has the reference's order and length, so the
optimizer state of a reference Lightning checkpoint loads — and lands on the right parameters — through
`trainer.fit(ckpt_path=...)`; the kernel-side order of the flat buffer (`flat.order`) is independent of it.  Without
`params` only the DenoiserTransformer's parameters are registered (in module order) and only self round trips work."""
from __future__ import annotations

from typing import Iterable, Optional

import torch


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, engine, lr: float = 2e-4, betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6,
                 params: Optional[Iterable[torch.nn.Parameter]] = None):
        self.engine = engine
        flat = engine.flat
        mine = {id(flat.named[n]) for n in flat.order}
        if params is None:
            params = list(engine.module.parameters())          # registration order of the DenoiserTransformer, not flat.order
        params = list(params)
        got = {id(p) for p in params}
        if not mine <= got:
            raise ValueError("FusedAdamW: `params` must contain every parameter of the DenoiserTransformer the engine trains")
        for p in params:
            if id(p) not in mine and p.requires_grad:
                raise ValueError("FusedAdamW: a trainable parameter outside the DenoiserTransformer was passed; the fused kernel "
                                 "only updates the engine's flat buffer (freeze it, as train_denoiser.py:33-35 does, or give it "
                                 "its own optimizer)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # in_backward: arm() before a backward lets every layer take its update (and gradient clear) as soon as its gradients are final
        # (DenoiserTrainEngine.arm_optimizer); step() then closes the step with the same hyper-parameters.  One backward per step only.
        self.in_backward = False
        self._armed = False
        self._bind_state()

    def _bind_state(self) -> None:
        flat = self.engine.flat
        self._step_t = torch.tensor(float(self.engine.step_count))
        for n in flat.order:
            p = flat.named[n]
            st = self.state[p]
            st["step"] = self._step_t
            st["exp_avg"] = flat.view(flat.exp_avg, n)
            st["exp_avg_sq"] = flat.view(flat.exp_avg_sq, n)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        armed, self._armed = self._armed, False
        self.engine.optimizer_step(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], zero_grad=armed)
        self._step_t += 1          # ONE tensor shared by every parameter's state["step"] (150 separate increments cost 0.3 ms of host time)
        return loss

    def arm(self) -> None:
        """optimizer-in-backward for the next backward pass (ignored by the engine when gradients are exchanged between ranks first)"""
        g = self.param_groups[0]
        self.engine.arm_optimizer(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], zero_grad=True)
        self._armed = True

    def zero_grad(self, set_to_none: bool = True) -> None:
        """gradients stay views of the flat buffer (set_to_none would detach them from the kernels)"""
        self.engine.flat.zero_grad()
        self.engine.flat.attach_grads()

    def load_state_dict(self, state_dict) -> None:
        """positional load like every torch optimizer; the moments are then copied into the flat buffers and re-bound as views.
        Shapes are checked per parameter NAME (the AdaLN tables of norm1 / norm2 have identical shapes: a permuted group
        would otherwise load silently onto the wrong parameters)."""
        n_saved = sum(len(g["params"]) for g in state_dict["param_groups"])
        n_here = sum(len(g["params"]) for g in self.param_groups)
        if n_saved != n_here:
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint's optimizer covers {n_saved} parameters, this one {n_here}: "
                             "construct it with params=list(lightning_module.parameters()) to match a reference checkpoint")
        super().load_state_dict(state_dict)
        flat = self.engine.flat
        steps = []
        with torch.no_grad():
            for n in flat.order:
                st = self.state[flat.named[n]]
                if "exp_avg" not in st:               # a checkpoint saved before this parameter took its first step
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = flat.view(flat.exp_avg, n).zero_()
                    st["exp_avg_sq"] = flat.view(flat.exp_avg_sq, n).zero_()
                for key, buf in (("exp_avg", flat.exp_avg), ("exp_avg_sq", flat.exp_avg_sq)):
                    view = flat.view(buf, n)
                    if tuple(st[key].shape) != tuple(view.shape):
                        raise ValueError(f"FusedAdamW.load_state_dict: {key} of {n} has shape {tuple(st[key].shape)}, expected {tuple(view.shape)}")
                    if st[key].data_ptr() != view.data_ptr():
                        view.copy_(st[key])
                        st[key] = view
                steps.append(int(st["step"]))
        self.engine.step_count = max(steps) if steps else 0
        self.engine.tables_state_changed()           # the moments of the timestep tables came from the checkpoint
        self._step_t = torch.tensor(float(self.engine.step_count))       # one shared counter again (see step())
        for n in flat.order:
            self.state[flat.named[n]]["step"] = self._step_t
