"""Data-parallel training of the Denoiser through its module surface — the reference's launch line

    python train_denoiser.py experiment_name=... data.batch_size=64 +trainer.devices=4 +trainer.strategy=ddp
    (scripts/train_denoiser.sh:1-7, train_denoiser.py:17-60)

served by this file when Lightning / Hydra are not the ones driving it:

    python -m pfpp_hip.launch --config-dir config/denoiser experiment_name=... data.batch_size=64 \
        +trainer.devices=4 +trainer.strategy=ddp

`Trainer` is the subset of `lightning.pytorch.Trainer` that train_denoiser.py:44-60 uses (callbacks / logger objects are accepted
and ignored).  What it does for `devices=N, strategy="ddp"`:

* one process per GPU: like Lightning's subprocess launcher it re-executes the command line N-1 times with LOCAL_RANK /
  WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set (under torchrun it takes the ranks it is given), binds the process to
  `cuda:LOCAL_RANK` and calls `init_process_group("nccl")` (RCCL);
* the train loader gets a `DistributedSampler` (shuffle as the loader had it, `set_epoch` per epoch) keeping the loader's
  batch size / drop_last / workers / collate function — every rank sees `batch_size` puzzles per step, the reference's
  per-device batch (denoiser/dataset/dataset.py:276-294);
* NO DistributedDataParallel wrapper: the HIP training path writes parameter gradients from its own kernels, so a hook-based
  reducer never sees them.  The gradient exchange is the engine's (pfpp_hip.parallel.GradExchange: per-layer all-reduce under the
  backward, AdamW per layer behind it); `accumulate_grad_batches=k` maps to `engine.no_sync()` around the first k-1 backward
  passes of a step (what Lightning does with DDP.no_sync);
* the loop body is the module's own surface in the benchmarked schedule: `for batch in model.training_schedule(loader):
  training_step -> backward -> optimizer.step -> zero_grad`; validation every `check_val_every_n_epoch` epochs on every rank's
  shard with `sync_dist` logging; `last.ckpt` in Lightning's checkpoint layout written by rank 0.
"""
from __future__ import annotations

import contextlib
import os
import socket
import subprocess
import sys
from typing import Any, Dict, List, Optional

import torch


# ------------------------------------------------------------------------------------------------- config tree (Hydra's subset)
def _deep_merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _set_path(tree: Dict[str, Any], dotted: str, value: Any, create: bool) -> None:
    keys = dotted.split(".")
    node = tree
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create and k not in node:
                raise KeyError(f"override {dotted}: no such key (prefix it with + to add it, as Hydra does)")
            node[k] = {} if not isinstance(node.get(k), dict) else node[k]
        node = node[k]
    if not create and keys[-1] not in node:
        raise KeyError(f"override {dotted}: no such key (prefix it with + to add it, as Hydra does)")
    node[keys[-1]] = value


def compose(config_dir: str, config_name: str = "global_config", overrides: Optional[List[str]] = None) -> Dict[str, Any]:
    """the part of Hydra's composition the reference's config/denoiser tree needs: the `defaults:` list of root-level files merged
    in order (`_self_` = this file), `a.b=c` / `+a.b=c` overrides (yaml-typed values), `${a.b}` / `${hydra:runtime.cwd}`
    interpolation.  -> a plain nested dict"""
    import yaml

    def load(name):
        with open(os.path.join(config_dir, name + ".yaml")) as fh:
            return yaml.safe_load(fh) or {}

    root = load(config_name)
    defaults = root.pop("defaults", ["_self_"])
    root.pop("hydra", None)
    tree: Dict[str, Any] = {}
    for d in defaults:
        if isinstance(d, dict):                       # `override hydra/...: disabled` and group entries: not part of the data tree
            continue
        _deep_merge(tree, root if d == "_self_" else load(d))
    for ov in overrides or []:
        key, _, val = ov.partition("=")
        create = key.startswith("+")
        _set_path(tree, key.lstrip("+"), yaml.safe_load(val) if val != "" else None, create)

    def lookup(path):
        node = tree
        for k in path.split("."):
            node = node[k]
        return node

    def resolve(v, depth=0):
        if isinstance(v, dict):
            return {k: resolve(x, depth) for k, x in v.items()}
        if isinstance(v, list):
            return [resolve(x, depth) for x in v]
        if isinstance(v, str) and "${" in v:
            if depth > 8:
                raise ValueError(f"interpolation cycle at {v!r}")
            out, i = "", 0
            while i < len(v):
                j = v.find("${", i)
                if j < 0:
                    out += v[i:]
                    break
                e = v.index("}", j)
                ref = v[j + 2: e]
                sub = os.getcwd() if ref == "hydra:runtime.cwd" else resolve(lookup(ref), depth + 1)
                if j == 0 and e == len(v) - 1:
                    return sub                        # a whole-value reference keeps its type
                out += v[i:j] + str(sub)
                i = e + 1
            return out
        return v

    return resolve(tree)


# ------------------------------------------------------------------------------------------------- trainer
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Trainer:
    """`lightning.pytorch.Trainer(**cfg.trainer)` for the Denoiser's module surface (see the module docstring)"""

    def __init__(self, accelerator: str = "gpu", devices: Any = 1, strategy: str = "auto", max_epochs: int = 1, max_steps: int = -1,
                 accumulate_grad_batches: int = 1, check_val_every_n_epoch: int = 1, num_sanity_val_steps: int = 0, precision: Any = 32,
                 callbacks=None, logger=None, profiler=None, gradient_clip_val: Optional[float] = None,
                 default_root_dir: Optional[str] = None, use_distributed_sampler: bool = True, seed: int = 0, **ignored):
        if str(precision) not in ("32", "32-true"):
            raise ValueError("Trainer: this path trains in fp32 (config/denoiser/global_config.yaml: precision 32)")
        if accelerator not in ("gpu", "auto", "cuda"):
            raise ValueError("Trainer: there is no CPU training path")
        name = str(strategy).lower()
        if any(k in name for k in ("fsdp", "deepspeed")):
            raise ValueError(f"Trainer: strategy {strategy!r} shards parameters through gradient hooks the HIP training path never fires; "
                             "use 'ddp' (the built-in per-layer gradient exchange) or 'auto'")
        self.strategy = name
        self.devices = len(devices) if isinstance(devices, (list, tuple)) else (torch.cuda.device_count() if devices in ("auto", -1) else int(devices))
        self.max_epochs, self.max_steps = int(max_epochs), int(max_steps)
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.check_val_every_n_epoch = int(check_val_every_n_epoch or 0)
        self.num_sanity_val_steps = int(num_sanity_val_steps or 0)
        self.gradient_clip_val = gradient_clip_val
        self.default_root_dir = default_root_dir or os.getcwd()
        self.use_distributed_sampler = bool(use_distributed_sampler)
        self.seed = int(seed)
        self.callbacks, self.logger = callbacks or [], logger
        self.current_epoch = 0
        self.global_step = 0
        self.global_rank = self.local_rank = 0
        self.world_size = 1
        self.logged_metrics: Dict[str, float] = {}
        self._children: List[subprocess.Popen] = []

    # -- processes --------------------------------------------------------------------------------------
    def _launch(self) -> None:
        """one process per device.  Already under a launcher (LOCAL_RANK set: torchrun, or one of our own children): take the rank we
        are given.  Otherwise this process becomes rank 0 and re-executes its command line for ranks 1..N-1 (Lightning's subprocess
        launcher does the same for strategy=ddp)."""
        world = self.devices
        self._launcher = False
        if "LOCAL_RANK" in os.environ:
            self.local_rank = int(os.environ["LOCAL_RANK"])
            self.global_rank = int(os.environ.get("RANK", self.local_rank))
            self.world_size = int(os.environ.get("WORLD_SIZE", world))
            if self.world_size != world and world > 1:
                raise RuntimeError(f"Trainer(devices={world}) under a launcher with WORLD_SIZE={self.world_size}")
        elif world > 1:
            # this process is the launcher: remembered on the object (and the rank variables taken out of its environment again
            # after fit — a second fit() in this process must not mistake itself for a launched rank)
            self._launcher = True
            self._env_before = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_RANK", "RANK")}
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not os.environ.get("MASTER_PORT"):                     # a port the user chose stays; otherwise a free one
                os.environ["MASTER_PORT"] = str(_free_port())
            os.environ.update(WORLD_SIZE=str(world), LOCAL_RANK="0", RANK="0")
            self.world_size = world
            main = sys.modules.get("__main__")
            spec = getattr(main, "__spec__", None)
            cmd = [sys.executable] + (["-m", spec.name] if spec is not None else [os.path.abspath(sys.argv[0])]) + sys.argv[1:]
            for r in range(1, world):
                env = dict(os.environ, LOCAL_RANK=str(r), RANK=str(r))
                self._children.append(subprocess.Popen(cmd, env=env))
        if self.world_size > 1:
            import datetime

            import torch.distributed as dist

            backend = os.environ.get("PFPP_DDP_BACKEND", "nccl")
            n_dev = torch.cuda.device_count()
            if backend == "nccl" and n_dev < self.world_size:
                raise RuntimeError(f"Trainer(devices={self.world_size}): only {n_dev} GPU(s) visible (RCCL needs one device per rank)")
            torch.cuda.set_device(self.local_rank % max(1, n_dev))
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if self._children:
                    # a child that died before the rendezvous would leave rank 0 waiting for the full timeout: look at them first
                    import time

                    time.sleep(0.2)
                    self._check_children()
                if self._children:
                    # and keep looking: a worker that dies later (an exception during its torch import, a raise in its step) drops its
                    # process group without a handshake while this rank may sit inside a collective — nothing in the main thread can
                    # notice that before the collective's timeout, so a daemon thread does and ends the job loudly
                    import threading

                    def _watch(children=tuple(self._children)):
                        import time

                        while True:
                            time.sleep(1.0)
                            if not self._children:               # fit() finished or aborted: nothing left to watch
                                return
                            for p in children:
                                rc = p.poll()
                                if rc is not None and rc != 0:
                                    print(f"Trainer: worker rank (pid {p.pid}) exited with code {rc} while rank 0 was still running — stopping",
                                          file=sys.stderr, flush=True)
                                    self._abort()
                                    os._exit(70)

                    threading.Thread(target=_watch, daemon=True, name="pfpp-rank-watchdog").start()
                dist.init_process_group(backend, rank=self.global_rank, world_size=self.world_size,
                                        timeout=datetime.timedelta(seconds=int(os.environ.get("PFPP_DDP_TIMEOUT_S", "1800"))))

    def _check_children(self) -> None:
        """a worker rank that has exited while rank 0 still trains means every collective from here on would hang: stop now"""
        for p in self._children:
            rc = p.poll()
            if rc is not None and rc != 0:
                raise RuntimeError(f"Trainer: worker rank (pid {p.pid}) exited with code {rc} while rank 0 was still running")

    def _abort(self) -> None:
        """failure path of fit(): no barrier (the other ranks may sit in a collective this rank will never enter) — the children are
        terminated and reaped, the process group is dropped without a handshake"""
        for p in self._children:
            if p.poll() is None:
                p.terminate()
        for p in self._children:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        self._children = []
        try:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                dist.destroy_process_group()
        except Exception:             # noqa: BLE001 — already failing; the original exception is the one to report
            pass

    def _restore_env(self) -> None:
        if getattr(self, "_launcher", False):
            for k, v in self._env_before.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            self._launcher = False

    def _join(self) -> None:
        rc = 0
        for p in self._children:
            rc = rc or p.wait()
        self._children = []
        if rc:
            raise RuntimeError(f"Trainer: a worker rank exited with code {rc}")

    # -- data ---------------------------------------------------------------------------------------------
    def _shard(self, loader, train: bool):
        """the loader with a DistributedSampler in place of its own (what Lightning's use_distributed_sampler does)"""
        from torch.utils.data import DataLoader, DistributedSampler, RandomSampler

        if self.world_size == 1 or not self.use_distributed_sampler or not isinstance(loader, DataLoader):
            return loader, None
        if isinstance(getattr(loader, "sampler", None), DistributedSampler):
            return loader, loader.sampler
        shuffle = isinstance(loader.sampler, RandomSampler)
        sampler = DistributedSampler(loader.dataset, num_replicas=self.world_size, rank=self.global_rank, shuffle=shuffle,
                                     seed=self.seed, drop_last=False)
        kw = dict(batch_size=loader.batch_size, sampler=sampler, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                  pin_memory=loader.pin_memory, drop_last=loader.drop_last, timeout=loader.timeout, worker_init_fn=loader.worker_init_fn)
        if loader.num_workers > 0:
            kw.update(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor)
        return DataLoader(loader.dataset, **kw), sampler

    # -- fit ----------------------------------------------------------------------------------------------
    def fit(self, model, train_dataloaders, val_dataloaders=None, ckpt_path: Optional[str] = None) -> None:
        if self.callbacks or self.logger is not None:
            import warnings

            warnings.warn("pfpp_hip.launch.Trainer: callbacks / logger objects are accepted and IGNORED (no ModelCheckpoint(monitor=...) "
                          "top-k files, no LearningRateMonitor, no wandb): only last.ckpt is written, per epoch, by rank 0")
        try:
            self._launch()
            self._fit(model, train_dataloaders, val_dataloaders, ckpt_path)
        except BaseException:
            self._abort()
            self._restore_env()
            raise
        # success path only: every rank got here, so the barrier cannot hang
        if self.world_size > 1:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()
        try:
            self._join()
        finally:
            self._restore_env()

    def _sync_replicas(self, engine) -> None:
        """what Lightning's DDP does at start-up (broadcast of rank 0's module state): parameters, both Adam moments and the step count
        from rank 0, then the fp16 planes re-split — replicas are equal by construction, not by every rank having seeded and loaded
        identically (the per-element AdamW guard relies on bit-identical replicas)"""
        if self.world_size <= 1:
            return
        import torch.distributed as dist

        f = engine.flat
        for t in (f.params, f.exp_avg, f.exp_avg_sq):
            dist.broadcast(t, src=0)
        step = torch.tensor([engine.step_count], dtype=torch.int64, device=f.params.device)
        dist.broadcast(step, src=0)
        engine.step_count = int(step.item())
        f.refresh_planes()
        f.after_optimizer_step()

    def _fit(self, model, train_loader, val_loader, ckpt_path) -> None:
        dev = torch.device("cuda", torch.cuda.current_device())
        model.to(dev)
        if not _has_trainer_property(model):          # (a real LightningModule's `trainer` is a property Lightning itself sets)
            object.__setattr__(model, "trainer", self)
        model.on_fit_start()
        conf = model.configure_optimizers()
        opt = conf["optimizer"] if isinstance(conf, dict) else conf
        sched = conf.get("lr_scheduler") if isinstance(conf, dict) else None
        if isinstance(sched, dict):
            sched = sched.get("scheduler")
        engine = model.denoiser.train_engine()
        start_epoch = 0
        if ckpt_path is not None:
            ck = torch.load(ckpt_path, map_location=dev, weights_only=False)
            model.load_state_dict(ck["state_dict"])
            if ck.get("optimizer_states"):
                opt.load_state_dict(ck["optimizer_states"][0])
            if sched is not None and ck.get("lr_schedulers"):
                sched.load_state_dict(ck["lr_schedulers"][0])
            start_epoch, self.global_step = int(ck.get("epoch", -1)) + 1, int(ck.get("global_step", 0))
        self._sync_replicas(engine)
        train_loader, sampler = self._shard(train_loader, True)
        if val_loader is not None:
            val_loader, _ = self._shard(val_loader, False)
            if self.num_sanity_val_steps > 0:
                self._validate(model, val_loader, dev, limit=self.num_sanity_val_steps, log=False)
        acc = self.accumulate_grad_batches
        if getattr(opt, "in_backward", False) and (acc > 1 or self.gradient_clip_val):
            opt.in_backward = False               # per-layer updates inside the backward leave nothing to accumulate into / to clip
        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = epoch
            if sampler is not None:
                sampler.set_epoch(epoch)
            model.train()
            n_batches = len(train_loader) if hasattr(train_loader, "__len__") else None
            for i, batch in enumerate(model.training_schedule(train_loader, dev)):
                last = (i + 1) % acc == 0 or (n_batches is not None and i + 1 == n_batches)
                with (contextlib.nullcontext() if last else engine.no_sync()):
                    loss = model.training_step(batch, i)
                    (loss / acc if acc > 1 else loss).backward()
                if last:
                    if self.gradient_clip_val:
                        self._clip(engine, float(self.gradient_clip_val))
                    opt.step()
                    opt.zero_grad()
                    self.global_step += 1
                    if self._children and self.global_step % 16 == 0:
                        self._check_children()
                    if 0 < self.max_steps <= self.global_step:
                        break
            if sched is not None:
                sched.step()
            self._collect_logs(model)
            if val_loader is not None and self.check_val_every_n_epoch and (epoch + 1) % self.check_val_every_n_epoch == 0:
                self._validate(model, val_loader, dev)
            self.save_checkpoint(os.path.join(self.default_root_dir, "last.ckpt"), model, opt, sched)
            if 0 < self.max_steps <= self.global_step:
                break

    @staticmethod
    def _clip(engine, max_norm: float) -> None:
        """clip_grad_norm_ over the flat gradient buffer (Lightning's gradient_clip_val, algorithm "norm"): the exchange is waited
        for first — until then the buffer holds sums in flight, not the mean"""
        mean = engine.finish_grad_exchange()
        g = engine.flat.grads
        coef = (max_norm / (g.norm() * mean + 1e-6)).clamp(max=1.0)
        g.mul_(coef)

    def _validate(self, model, loader, dev, limit: Optional[int] = None, log: bool = True) -> None:
        was_training = model.training
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if limit is not None and i >= limit:
                    break
                batch = model.on_before_batch_transfer(dict(batch))
                batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                batch = model.on_after_batch_transfer(batch)
                model.validation_step(batch, i)
            model.on_validation_epoch_end()
        if log:
            self._collect_logs(model)
        model.train(was_training)

    def _collect_logs(self, model) -> None:
        for k, v in getattr(model, "logged", {}).items():
            self.logged_metrics[k] = float(v.detach().float().mean()) if torch.is_tensor(v) else float(v)

    def save_checkpoint(self, path: str, model, opt, sched=None) -> None:
        """Lightning's checkpoint layout (what `trainer.fit(ckpt_path=...)` and test.py:26-38 read), written by rank 0"""
        if self.global_rank != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        ck = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "2.0.0",
              "state_dict": model.state_dict(), "optimizer_states": [opt.state_dict()],
              "lr_schedulers": [sched.state_dict()] if sched is not None else []}
        tmp = path + ".tmp"
        torch.save(ck, tmp)
        os.replace(tmp, path)


def _has_trainer_property(model) -> bool:
    return isinstance(getattr(type(model), "trainer", None), property)


# ------------------------------------------------------------------------------------------------- train_denoiser.py
def main(argv: Optional[List[str]] = None) -> None:
    """train_denoiser.py:17-60 with the reference's yaml tree and override syntax"""
    import argparse

    from . import config as cfgmod
    from .lightning_compat import instantiate

    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config-dir", default="config/denoiser")
    ap.add_argument("--config-name", default="global_config")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args(argv)
    tree = compose(a.config_dir, a.config_name, a.overrides)
    cfg = cfgmod.to_namespace(tree)
    torch.manual_seed(int(tree.get("train_seed", 0)))                          # pl.seed_everything(cfg.train_seed)
    from puzzlefusion_plusplus.denoiser.dataset.dataset import build_geometry_dataloader

    out_dir = os.path.join(str(tree.get("experiment_output_path", ".")), "training")
    os.makedirs(out_dir, exist_ok=True)
    train_loader, val_loader = build_geometry_dataloader(cfg)
    model = instantiate(cfg.model.model_name, cfg)
    if getattr(cfg.model, "encoder_weights_path", None) is not None:           # train_denoiser.py:30-35
        sd = torch.load(cfg.model.encoder_weights_path, map_location="cpu", weights_only=False)["state_dict"]
        model.encoder.load_state_dict({k.replace("ae.", ""): v for k, v in sd.items()})
    for p in model.encoder.parameters():
        p.requires_grad = False
    tr = dict(tree.get("trainer", {}))
    trainer = Trainer(default_root_dir=out_dir, seed=int(tree.get("train_seed", 0)), **tr)
    ckpt = tree.get("ckpt_path")
    if ckpt is not None and not os.path.exists(ckpt):
        raise FileNotFoundError("Error: Checkpoint path does not exist.")
    trainer.fit(model=model, train_dataloaders=train_loader, val_dataloaders=val_loader, ckpt_path=ckpt)


if __name__ == "__main__":
    main()
