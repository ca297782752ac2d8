"""Denoiser shell (drop-in for puzzlefusion_plusplus/denoiser/model/denoiser.py).

Keeps the LightningModule surface of the reference (forward(data_dict) -> {pred_noise, gt_noise},
_loss, training_step, validation_step, configure_optimizers; state_dict prefixes `denoiser.` and
`encoder.`) and runs the hot path — rotate, encode, denoise, DDPM step — on the HIP kernels.
In train mode DenoiserTransformer.forward is one autograd node backed by pfpp_hip.train (dropouts, backward
kernels, gradients accumulated into the parameters' .grad) and configure_optimizers returns the fused AdamW.
validation_step computes the reference's four metrics with denoiser/evaluation/evaluator.py (HIP nearest-neighbour kernel).
"""
from __future__ import annotations

import os

import torch

from pfpp_hip.lightning_compat import LightningModule, instantiate
from pfpp_hip.scheduler import PiecewiseScheduler
from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer
from puzzlefusion_plusplus.denoiser.model.modules.encoder import VQVAE


class _MaskedMSE(torch.autograd.Function):
    """mean over the selected rows of (pred - gt)^2 with its gradient from the same kernel (pfpp_hip.train_ops.mse_loss)"""

    @staticmethod
    def forward(ctx, pred, gt, valids, ref):
        from pfpp_hip import train_ops as T
        from pfpp_hip.train import _f32c, _u8

        n = valids.numel()
        # valid & ~reference is evaluated inside the kernel (pfpp_mse_loss_masked): no mask tensors, no extra launches
        loss, dpred = T.mse_loss_masked(pred.detach().reshape(n, -1).contiguous(), _f32c(gt).reshape(n, -1), _f32c(valids).reshape(n),
                                        _u8(ref).reshape(n))
        ctx.save_for_backward(dpred)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        (dpred,) = ctx.saved_tensors
        return (dpred * grad_out).view(ctx.shape), None, None, None


class Denoiser(LightningModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.denoiser = DenoiserTransformer(cfg)
        self.save_hyperparameters()
        m = cfg.model
        self.noise_scheduler = PiecewiseScheduler(
            num_train_timesteps=m.DDPM_TRAIN_STEPS, beta_schedule=m.DDPM_BETA_SCHEDULE, prediction_type=m.PREDICT_TYPE,
            beta_start=m.BETA_START, beta_end=m.BETA_END, clip_sample=False, timestep_spacing=m.timestep_spacing)
        ae_name = getattr(cfg.ae, "ae_name", None)
        self.encoder = instantiate(ae_name, cfg) if ae_name is not None else VQVAE(cfg)
        self.num_points = m.num_point
        self.num_channels = m.num_dim
        self.noise_scheduler.set_timesteps(num_inference_steps=m.num_inference_steps)
        self.rmse_r_list, self.rmse_t_list, self.acc_list, self.cd_list = [], [], [], []

    # ------------------------------------------------------------------ batch hooks (Lightning calls them around the H2D copy)
    def on_before_batch_transfer(self, batch, dataloader_idx=0):
        """while the collated batch is still on the host: derive the valid-fragment layout there, so the forward never
        reads part_valids back from the GPU (a device->host read would drain the streams once per iteration)"""
        pv = batch.get("part_valids") if isinstance(batch, dict) else None
        if torch.is_tensor(pv) and pv.device.type == "cpu" and pv.dim() == 2:
            batch["_pfpp_valids_host"] = pv.numpy().copy()        # numpy: the transfer leaves it on the host
        return batch

    def on_after_batch_transfer(self, batch, dataloader_idx=0):
        host = batch.pop("_pfpp_valids_host", None) if isinstance(batch, dict) else None
        if host is not None and batch["part_valids"].device.type == "cuda":
            from pfpp_hip.denoiser import CompactLayout, attach_layout

            attach_layout(batch["part_valids"], self.num_points,
                          CompactLayout.from_host(torch.from_numpy(host), self.num_points, batch["part_valids"].device))
        return batch

    # ------------------------------------------------------------------ hot path
    def _extract_features(self, part_pcs, part_valids, noisy_trans_and_rots):
        """rotate every fragment by its current noisy quaternion, encode the valid ones, scatter
        into zero-padded [B,P,L,*] (denoiser.py:55-77) — one fused pipeline on the GPU"""
        return self.encoder.extract_features(part_pcs, part_valids, noisy_trans_and_rots)

    def forward(self, data_dict, noise=None, timesteps=None):
        """training-style forward (denoiser.py:80-115); `noise` / `timesteps` may be injected"""
        ref_part = data_dict["ref_part"]
        if self.training and noise is None and timesteps is None and "_pfpp_features" in data_dict:
            # the batch came through training_schedule(): its noise draw, add_noise and encoder pass were issued one batch ahead
            # on the encoder stream (pfpp_hip.train.TrainingSchedule)
            from pfpp_hip.train import take_features

            f = take_features(data_dict)
            pred = self.denoiser(f["noisy"], f["t"], f["latent"], f["xyz"], data_dict["part_valids"], data_dict["part_scale"], ref_part)
            return {"pred_noise": pred, "gt_noise": f["noise"]}
        gt = torch.cat([data_dict["part_trans"], data_dict["part_rots"]], dim=-1).float().contiguous()
        B = gt.shape[0]
        if noise is None:
            noise = torch.randn(gt.shape, device=gt.device)
        if timesteps is None:
            timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (B,), device=gt.device).long()
        noisy = self.noise_scheduler.add_noise(gt, noise, timesteps)
        noisy = torch.where(ref_part.bool().unsqueeze(-1), gt, noisy)    # == noisy[ref_part] = gt[ref_part] (denoiser.py:95), no host sync
        latent, xyz = self._extract_features(data_dict["part_pcs"], data_dict["part_valids"], noisy)
        pred = self.denoiser(noisy, timesteps, latent, xyz, data_dict["part_valids"], data_dict["part_scale"], ref_part)
        return {"pred_noise": pred, "gt_noise": noise}

    def _loss(self, data_dict, output_dict):
        # F.mse_loss(pred[valids & ~ref], gt[valids & ~ref]) (denoiser.py:118-126) written as a masked mean: boolean-mask
        # indexing reads the selection size back to the host, which would stall the enqueue of every iteration
        pred, gt = output_dict["pred_noise"], output_dict["gt_noise"]
        if pred.is_cuda and pred.dtype == torch.float32 and pred.requires_grad and gt.dtype == torch.float32:
            # training: the loss and d loss / d pred in ONE launch (pfpp_mse_loss, what the training engine uses) instead of ten
            # dependent elementwise launches forward and ten backward at the turn of every iteration
            return {"mse_loss": _MaskedMSE.apply(pred, gt, data_dict["part_valids"], data_dict["ref_part"])}
        sel = data_dict["part_valids"].bool() & ~data_dict["ref_part"].bool()
        d = (pred - gt) * sel.unsqueeze(-1)
        return {"mse_loss": (d * d).sum() / (sel.sum() * d.shape[-1])}

    def training_schedule(self, batches, device=None):
        """iterate `batches` (a DataLoader or any iterable of batch dicts) in the benchmarked schedule: loop body on a high-priority
        stream, next batch's frozen-encoder pass one iteration ahead on the CU-masked stream (pfpp_hip.train.TrainingSchedule)"""
        from pfpp_hip.train import TrainingSchedule

        return TrainingSchedule(self, batches, device)

    def training_step(self, data_dict, idx):
        opt = getattr(self, "_fused_opt", None)
        if opt is not None and opt.in_backward and torch.is_grad_enabled():
            opt.arm()                      # AdamW per layer under the backward (behind the layer's all-reduce when N > 1)
        acc = getattr(self, "_pfpp_accumulate", 1)
        if acc > 1 and opt is not None:
            # gradient accumulation under a foreign trainer (Lightning wraps DDP.no_sync around the micro-batches; there is no DDP
            # wrapper here): only the last backward of an optimizer step starts the gradient exchange.  pfpp_hip.launch.Trainer
            # holds engine.no_sync() itself.
            trainer = self._trainer()
            if trainer is not None and type(trainer).__module__ != "pfpp_hip.launch":
                last = bool(getattr(getattr(getattr(trainer, "fit_loop", None), "epoch_loop", None), "batch_progress", None)
                            and trainer.fit_loop.epoch_loop.batch_progress.is_last_batch)
                opt.engine._sync = (idx + 1) % acc == 0 or last
        out = self(data_dict)
        total = 0
        for name, value in self._loss(data_dict, out).items():
            total = total + value
            self.log(f"train_loss/{name}", value, on_step=True, on_epoch=False)
        self.log("train_loss/total_loss", total, on_step=True, on_epoch=False)
        return total

    @torch.no_grad()
    def sample(self, data_dict, x_init=None, noises=None, record=None):
        """the 20-step ancestral sampler of validation_step (denoiser.py:153-185); returns the final
        [B,P,7] poses.  x_init / noises inject the random draws (parity tests).
        Range guard of the split-f16 GEMMs: weights are pre-scaled per tensor at pack time (packing.plane_scale), activations
        are split as they are — if one reaches the fp16 range (|v| >= 65504) the poses come out non-finite, and the whole
        sampler is re-run with the exact fp32 GEMMs from the same random draws."""
        from pfpp_hip import ops

        dev = data_dict["part_trans"].device
        rng = torch.cuda.get_rng_state(dev) if (x_init is None or noises is None) and dev.type == "cuda" else None
        rec = [] if record is not None else None
        x = self._sample(data_dict, x_init, noises, rec)
        if ops.f16x3_range_fallback(x):
            import warnings

            warnings.warn("split-f16 GEMM operand out of the fp16 range: re-running the sampler with exact fp32 GEMMs")
            if rng is not None:
                torch.cuda.set_rng_state(rng, dev)
            rec = [] if record is not None else None
            with ops.exact_fp32():
                x = self._sample(data_dict, x_init, noises, rec)
        if record is not None:
            record.extend(rec)
        return x

    def _sample(self, data_dict, x_init, noises, record):
        gt = torch.cat([data_dict["part_trans"], data_dict["part_rots"]], dim=-1).float().contiguous()
        ref_part = data_dict["ref_part"]
        x = torch.randn(gt.shape, device=gt.device) if x_init is None else x_init.clone()
        is_ref = ref_part.bool().unsqueeze(-1)
        reference = torch.where(is_ref, gt, torch.zeros_like(gt))      # reference[ref] = gt[ref]; x[ref] = reference[ref]
        x = torch.where(is_ref, reference, x)
        B = x.shape[0]
        for i, t in enumerate(self.noise_scheduler.timesteps.tolist()):
            ts = torch.full((B,), t, dtype=torch.int64, device=x.device)
            ts._pfpp_t = int(t)          # every puzzle at the same timestep: the AdaLN rows of (t, B) are computed once (pfpp_hip.denoiser.ada_mods)
            latent, xyz = self._extract_features(data_dict["part_pcs"], data_dict["part_valids"], x)
            eps = self.denoiser(x, ts, latent, xyz, data_dict["part_valids"], data_dict["part_scale"], ref_part)
            x = self.noise_scheduler.step(eps, t, x, variance_noise=None if noises is None else noises[i],
                                          ref_part=ref_part, reference=reference).prev_sample
            if record is not None:
                record.append(x.clone())
        return x

    def validation_step(self, data_dict, idx):
        """validation loss, the 20-step sampler and the four evaluation metrics (denoiser.py:147-208)"""
        from puzzlefusion_plusplus.denoiser.evaluation.evaluator import (ChamferDistance, calc_part_acc, calc_shape_cd,
                                                                        rot_metrics, trans_metrics)

        with torch.no_grad():
            out = self(data_dict)
            for name, value in self._loss(data_dict, out).items():
                self.log(f"val_loss/{name}", value, on_step=False, on_epoch=True)
            x = self.sample(data_dict)
        gt_trans, gt_rots = data_dict["part_trans"].float(), data_dict["part_rots"].float()
        pred_trans, pred_rots = x[..., :3].contiguous(), x[..., 3:].contiguous()
        pts = data_dict["part_pcs"] * data_dict["part_scale"].unsqueeze(-1)          # denoiser.py:189-190
        metric = getattr(self, "metric", None) or ChamferDistance()
        valids = data_dict["part_valids"]
        acc, _, _ = calc_part_acc(pts, pred_trans, gt_trans, pred_rots, gt_rots, valids, metric)
        shape_cd = calc_shape_cd(pts, pred_trans, gt_trans, pred_rots, gt_rots, valids, metric)
        self.acc_list.append(acc)
        self.rmse_r_list.append(rot_metrics(pred_rots, gt_rots, valids, "rmse"))
        self.rmse_t_list.append(trans_metrics(pred_trans, gt_trans, valids, "rmse"))
        self.cd_list.append(shape_cd)
        return x

    def on_validation_epoch_end(self):
        """denoiser.py:211-227"""
        total = [torch.mean(torch.cat(v)) for v in (self.acc_list, self.rmse_t_list, self.rmse_r_list, self.cd_list)]
        for name, value in zip(("eval/part_acc", "eval/rmse_t", "eval/rmse_r", "eval/shape_cd"), total):
            self.log(name, value, sync_dist=True)
        self.acc_list, self.rmse_t_list, self.rmse_r_list, self.cd_list = [], [], [], []
        return tuple(total)

    def _trainer(self):
        """the attached trainer or None (a real LightningModule raises RuntimeError, not AttributeError, when unattached)"""
        try:
            return getattr(self, "trainer", None)
        except RuntimeError:
            return None

    def on_fit_start(self):
        """Data parallelism.  The training forward writes parameter gradients through its own kernels (one autograd node,
        pfpp_hip.train): torch's DistributedDataParallel would see no gradient hook fire.  The exchange is the engine's
        (pfpp_hip.parallel.GradExchange: per-layer all-reduce of the flat gradient buffer under the backward, over the process group
        the strategy initialised), so the reference's launch line `+trainer.devices=4 +trainer.strategy=ddp`
        (scripts/train_denoiser.sh:6-7) is served as follows:
        * pfpp_hip.launch.Trainer (Lightning absent): strategy "ddp" means exactly that built-in exchange;
        * a real Lightning DDPStrategy: it has already set the device, initialised the process group and injected the
          DistributedSampler — all of which are kept; only its DistributedDataParallel wrapper is taken off again
          (strategy.model = this module), and gradient accumulation is mapped to engine.no_sync() in training_step;
        * parameter-sharding strategies (FSDP, DeepSpeed) cannot work with a flat parameter buffer and are refused."""
        trainer = self._trainer()
        strategy = getattr(trainer, "strategy", None)
        name = strategy if isinstance(strategy, str) else (type(strategy).__name__.lower() if strategy is not None else "")
        if "fsdp" in name or "deepspeed" in name:
            raise RuntimeError(f"Denoiser: strategy {name} shards parameters through gradient hooks that the HIP training path never "
                               "fires; use strategy='ddp' (served by the built-in per-layer gradient exchange) or 'auto'")
        if "ddp" in name and not isinstance(strategy, str):
            wrapped = getattr(strategy, "model", None)
            if wrapped is not None and wrapped is not self and isinstance(wrapped, torch.nn.parallel.DistributedDataParallel):
                strategy.model = self                 # Lightning then calls training_step on the module itself
        object.__setattr__(self, "_pfpp_accumulate", int(getattr(trainer, "accumulate_grad_batches", 1) or 1))

    def configure_optimizers(self):
        # same hyper-parameters as the reference (denoiser.py:230-237) on the fused kernel; the frozen encoder
        # (train_denoiser.py:33-35) has no gradients and therefore no optimizer state in either implementation
        from pfpp_hip.optim import FusedAdamW

        # params = the whole module's parameter list, like the reference's AdamW(self.parameters()): same group order and
        # length, so the optimizer state of a reference checkpoint loads positionally onto the right parameters
        optimizer = FusedAdamW(self.denoiser.train_engine(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-08,
                               params=list(self.parameters()))
        # optimizer-in-backward (the benchmarked form of the step) when nothing accumulates gradients over several backward passes
        # (per-layer updates and gradient clears inside loss.backward() leave nothing for gradient clipping / accumulation to act on)
        trainer = self._trainer()
        accumulate = getattr(trainer, "accumulate_grad_batches", 1) or 1
        clip = getattr(trainer, "gradient_clip_val", None)
        optimizer.in_backward = os.environ.get("PFPP_OPT_IN_BWD", "1") == "1" and accumulate == 1 and not clip
        object.__setattr__(self, "_fused_opt", optimizer)
        sched_cfg = getattr(self.cfg.model, "lr_scheduler", None)
        if sched_cfg is None:
            return optimizer
        return {"optimizer": optimizer, "lr_scheduler": instantiate(sched_cfg, optimizer)}
