"""bench.py — denoiser DDPM-step throughput on synthetic Breaking-Bad-shaped puzzles.

    python bench.py --gpus N --steps K --warmup W [--mode train|sample]

Default (--mode train) = BASELINE.json configs[1]: one "step" is one DDPM TRAINING iteration of the denoiser
over a batch resident in HBM (Denoiser.training_step, puzzlefusion_plusplus/denoiser/model/denoiser.py:80-145):
draw noise and one timestep per puzzle -> add_noise -> re-pin reference parts -> rotate every fragment by its
noisy pose -> frozen PointNet++/VQ-VAE encode -> DenoiserTransformer forward in train mode (dropouts on) ->
MSE over valid non-reference fragments -> full backward -> (N > 1: gradient all-reduce over RCCL, overlapped
with the backward) -> AdamW.  Nothing is skipped inside the timed region.
--mode sample: one step = one DDPM sampler step (Denoiser.validation_step loop body, denoiser.py:172-185),
reported under "extra" in the default mode.
Workload shape on every GPU: 32 puzzles x 20 fragment slots x 1024 points, valid-fragment counts from
SURVEY.md §8d's distribution.  N > 1: one process per GPU, different puzzles per rank (weak scaling).

--mode stress: BASELINE configs[4] — 100 fragments per puzzle x 2048 points, one joint step = rotate + encode + DenoiserTransformer
+ scheduler step + VerifierTransformer on the 4,950 candidate edges, single-pass fp16 MFMA (PFPP_GEMM_F16) wherever both GEMM
operands travel as split planes; the line carries max |pred_noise(single pass) - pred_noise(f16x3)| next to the throughput.
Beyond the reference's max_len = 20: no reference parity exists for this shape, roofline only.

Prints ONE JSON line (rank 0) with the extra objects:
  roofline     — dominant GEMM kernel of the step: algorithmic FLOPs of its launches / their HIP-event durations against the
                 matrix-core ceiling of its arithmetic (f16 dense peak / 3 for the split-f16 kernels)
  roofline_hbm — the bandwidth / latency regime (reported separately): FPS, ball query, rotate, VQ, LayerNorm,
                 AdamW, scheduler step timed alone at the workload's shapes, algorithmic bytes / time against 8 TB/s, and the
                 time per dependent argmax step of FPS
  cpu_baseline — the CPU oracle (oracle/, a port of the reference path) timed on this host on BASELINE.json configs[0]
                 (1 puzzle, 8 fragments x 512 points) following BASELINE.md §3 within a bounded budget
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime initialises: see pfpp_hip/__init__.py

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

# /opt/skills/guides/MI355X_MICROARCH.md, dense peaks
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 (exact fp32 path)
PEAK_F16_MFMA_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 (split-f16 path: 3 matrix instructions per fp32-grade product)
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel: str, mode: str = "train"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (profiles/*_pmc_FETCH_SIZE.csv and
    *_pmc_WRITE_SIZE.csv of the same bench mode, newest round first; separate --pmc runs of this bench, tools/diag/prof_pmc.sh).  The counters are in
    KB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  -> (bytes or None, description)"""
    import csv
    import glob
    import re

    prof = os.path.join(ROOT, "profiles")
    for fetch in sorted(glob.glob(os.path.join(prof, f"*_bench_{mode}_pmc_FETCH_SIZE.csv")), reverse=True):
        write = fetch.replace("FETCH_SIZE", "WRITE_SIZE")
        if not os.path.exists(write):
            continue
        vals, matched = {}, None
        for path, key in ((fetch, "fetch"), (write, "write")):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    # the traced kernel's name is the row's name up to its argument list (the full template instantiation: a shorter
                    # label must not pick up a sibling instantiation, and a stale one must not fall through to an older file silently)
                    m = re.search(r"(\w+(?:<[^()]*>)?)\(", row["kernel"])        # name<template arguments> in front of the argument list
                    if m and m.group(1) == kernel:
                        vals[key] = float(row["mean_value_per_dispatch"])
                        matched = m.group(1)
        if len(vals) == 2:
            return (int((2.0 * vals["fetch"] + vals["write"]) * 1024),
                    f"{os.path.basename(fetch)} (x2, gfx950) + {os.path.basename(write)}, mean per dispatch of {matched}")
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="puzzles per GPU")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--parts", type=int, default=None, help="fix the number of valid fragments per puzzle")
    ap.add_argument("--mode", choices=("train", "sample", "stress"), default="train")
    ap.add_argument("--stress-batch", type=int, default=8, help="stress mode: puzzles per GPU (100 fragments x 2048 points each)")
    ap.add_argument("--latents-given", action="store_true",
                    help="train mode: skip the encoder (latents of the clean pose precomputed; not a reference mode)")
    ap.add_argument("--serial", action="store_true",
                    help="train mode: one stream only (no side stream for weight gradients, encoder in line) — the execution "
                         "the roofline pass measures per-kernel durations in")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="train mode: run the frozen encoder of each iteration in line instead of one iteration ahead on its own stream")
    ap.add_argument("--compact", action="store_true",
                    help="drop padded fragment slots in the transformer (valid-fragment outputs unchanged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


class SamplerWorkload:
    """device-resident state of the sampler loop for one batch of puzzles"""

    def __init__(self, batch: int, points: int, parts, first_id: int, dev: torch.device, compact: bool = False, ids=None):
        from pfpp_hip import config, synthetic
        from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

        torch.manual_seed(1234)
        self.model = Denoiser(config.denoiser_config()).to(dev).eval()   # random-init weights of the reference architecture
        self.model.denoiser.compact_padded = compact
        with torch.no_grad():   # a codebook on the scale of the latents (a trained one is)
            self.model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
        data = synthetic.make_batch(first_id, batch, num_points=points, num_parts=parts, ids=ids)
        self.data = {k: v.to(dev) for k, v in data.items()}
        self.n_frag = int(self.data["part_valids"].sum().item())
        gt = torch.cat([self.data["part_trans"], self.data["part_rots"]], dim=-1).float().contiguous()
        self.ref = self.data["ref_part"]
        self.reference = torch.zeros_like(gt)
        self.reference[self.ref] = gt[self.ref]
        g = torch.Generator(device=dev).manual_seed(99 + first_id)
        self.x0 = torch.randn(gt.shape, device=dev, generator=g)
        self.x0[self.ref] = self.reference[self.ref]
        self.noise = [torch.randn(gt.shape, device=dev, generator=g) for _ in range(20)]
        self.timesteps = self.model.noise_scheduler.timesteps.tolist()
        self.ts_dev = {t: torch.full((batch,), t, dtype=torch.int64, device=dev) for t in self.timesteps}
        for t_, v_ in self.ts_dev.items():
            v_._pfpp_t = int(t_)          # as Denoiser.sample / AutoAgglomerative tag them: AdaLN rows cached per (t, batch)
        self.x = self.x0.clone()
        self.i = 0

    @torch.no_grad()
    def step(self):
        m, d = self.model, self.data
        k = self.i % len(self.timesteps)
        if k == 0:
            self.x = self.x0.clone()          # a new 20-step trajectory
        t = self.timesteps[k]
        latent, xyz = m._extract_features(d["part_pcs"], d["part_valids"], self.x)
        eps = m.denoiser(self.x, self.ts_dev[t], latent, xyz, d["part_valids"], d["part_scale"], self.ref)
        self.x = m.noise_scheduler.step(eps, t, self.x, variance_noise=self.noise[k], ref_part=self.ref,
                                        reference=self.reference).prev_sample
        self.i += 1


class TrainWorkload:
    """device-resident state of the training loop for one batch of puzzles"""

    def __init__(self, batch: int, points: int, parts, first_id: int, dev: torch.device, latents_given: bool = False,
                 pipeline: bool = True, ids=None):
        from pfpp_hip import config, synthetic
        from pfpp_hip.train import DenoiserTrainEngine
        from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

        torch.manual_seed(1234)
        self.model = Denoiser(config.denoiser_config()).to(dev)     # random-init weights of the reference architecture
        with torch.no_grad():   # a codebook on the scale of the latents (a trained one is)
            self.model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
        for p_ in self.model.encoder.parameters():                  # train_denoiser.py:33-35
            p_.requires_grad = False
        self.model.train()
        self.engine = DenoiserTrainEngine(self.model.denoiser)
        data = synthetic.make_batch(first_id, batch, num_points=points, num_parts=parts, ids=ids)
        self.data = {k: v.to(dev) for k, v in data.items()}
        self.n_frag = int(self.data["part_valids"].sum().item())
        self.gt = torch.cat([self.data["part_trans"], self.data["part_rots"]], dim=-1).float().contiguous()
        self.ref = self.data["ref_part"]
        self.gen = torch.Generator(device=dev).manual_seed(99 + first_id)
        self.batch = batch
        self.dev = dev
        self.latents_given = latents_given
        self.fixed = None
        if latents_given:
            with torch.no_grad():
                self.fixed = self.model._extract_features(self.data["part_pcs"], self.data["part_valids"], self.gt)
        self.i = 0
        self.last_loss = None
        # AdamW per layer under the backward (engine.arm_optimizer): each layer's slice is updated on the weight-gradient stream as soon as
        # its gradients are final, only embeddings / AdaLN / heads are left for the end.  Round 1 measured it slower (8.14 -> 8.24 ms);
        # with the faster weight-gradient kernels of round 2 the side stream has room: 8.28-8.39 -> 8.08 ms on the same box
        self._opt_in_bwd = os.environ.get("PFPP_BENCH_OPT_IN_BWD", "1") == "1"
        from pfpp_hip.train import FeaturePipeline

        self.pipeline = FeaturePipeline(self.model, dev) if (pipeline and not latents_given) else None
        # the transformer's dependency chain sets the length of the iteration; the encoder and the weight gradients fill
        # the chip underneath it from their own streams — so the chain runs on a high-priority stream (10.05 -> 9.9 ms)
        self._hi = (torch.cuda.Stream(device=dev, priority=-1)
                    if (pipeline and os.environ.get("PFPP_MAIN_HIGH", "1") == "1") else None)

    def _draw(self):
        noise = torch.randn(self.gt.shape, device=self.dev, generator=self.gen)
        t = torch.randint(0, self.model.noise_scheduler.config.num_train_timesteps, (self.batch,), device=self.dev, generator=self.gen)
        return noise, t

    def step(self):
        if self._hi is not None:
            with torch.cuda.stream(self._hi):
                self._step()
        else:
            self._step()

    def _step(self):
        m, d = self.model, self.data
        sch = m.noise_scheduler
        between = None
        if self.pipeline is not None:
            # the frozen encoder of the NEXT iteration's (noise, t) draw runs on its own stream under this iteration's
            # transformer work; every iteration still does one full encoder pass + one full transformer step
            if os.environ.get("PFPP_BENCH_ENC_AFTER_FWD", "0") == "1":       # lab: the next encoder under the BACKWARD instead of the forward
                f = self.pipeline.take(d, self.gt, self.ref, self._draw)
                between = lambda: self.pipeline.issue_next(d, self.gt, self.ref, self._draw)
            else:
                f = self.pipeline.next(d, self.gt, self.ref, self._draw)
            noisy, t, latent, xyz, noise = f["noisy"], f["t"], f["latent"], f["xyz"], f["noise"]
        else:
            noise, t = self._draw()
            noisy = sch.add_noise(self.gt, noise, t)
            noisy = torch.where(self.ref.unsqueeze(-1), self.gt, noisy)
            with torch.no_grad():
                latent, xyz = self.fixed if self.latents_given else m._extract_features(d["part_pcs"], d["part_valids"], noisy)
        self.engine.flat.zero_grad()
        if self._opt_in_bwd:
            self.engine.arm_optimizer(lr=2e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, zero_grad=True)
        self.last_loss = self.engine.loss_and_grads(noisy, t, latent, xyz, d["part_valids"], d["part_scale"], self.ref, noise,
                                                    seed=1000 + self.i, train=True, between=between)
        self.engine.optimizer_step(lr=2e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6,
                                   zero_grad=True)      # optimizer.step() + optimizer.zero_grad() in one pass over the buffers
        self.i += 1


class StressWorkload:
    """BASELINE configs[4]: B puzzles x 100 fragments x 2048 points; one step = rotate + encode + denoise + scheduler step +
    edge features of the stepped poses + verifier forward on all 4,950 candidate edges of every puzzle (joint denoiser + verifier)"""

    def __init__(self, batch: int, first_id: int, dev: torch.device, parts: int = 100, points: int = 2048):
        from pfpp_hip import config, synthetic
        from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser
        from puzzlefusion_plusplus.verifier.model.modules.verifier_transformer import VerifierTransformer

        torch.manual_seed(1234)
        self.model = Denoiser(config.denoiser_config(model=dict(max_len=parts))).to(dev).eval()
        self.verifier = VerifierTransformer(config.verifier_config(model=dict(max_len=parts))).to(dev).eval()
        with torch.no_grad():
            self.model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
        data = synthetic.make_batch(first_id, batch, num_points=points, max_parts=parts, num_parts=parts)
        self.data = {k: v.to(dev) for k, v in data.items()}
        self.n_frag = int(self.data["part_valids"].sum().item())
        gt = torch.cat([self.data["part_trans"], self.data["part_rots"]], dim=-1).float().contiguous()
        self.ref = self.data["ref_part"]
        self.reference = torch.zeros_like(gt)
        self.reference[self.ref] = gt[self.ref]
        g = torch.Generator(device=dev).manual_seed(99 + first_id)
        self.x0 = torch.randn(gt.shape, device=dev, generator=g)
        self.x0[self.ref] = self.reference[self.ref]
        self.noise = [torch.randn(gt.shape, device=dev, generator=g) for _ in range(20)]
        self.timesteps = self.model.noise_scheduler.timesteps.tolist()
        self.ts_dev = {t: torch.full((batch,), t, dtype=torch.int64, device=dev) for t in self.timesteps}
        for t_, v_ in self.ts_dev.items():
            v_._pfpp_t = int(t_)          # as Denoiser.sample / AutoAgglomerative tag them: AdaLN rows cached per (t, batch)
        E = parts * (parts - 1) // 2
        self.edge_idx = torch.triu(torch.ones(parts, parts, dtype=torch.bool), diagonal=1).nonzero()[None].expand(batch, E, 2).contiguous().to(dev)
        self.edge_feat = torch.rand(batch, E, 7, device=dev, generator=g)
        self.edge_valid = torch.ones(batch, E, device=dev)
        self.x = self.x0.clone()
        self.i = 0
        self.last_eps = None
        # edge features of the stepped poses (auto_aggl.py:153-201: transform the matched points by the current poses, nearest-
        # neighbour histogram per candidate edge) — the stage between the scheduler step and the verifier in the joint step
        from puzzlefusion_plusplus.auto_aggl import AutoAgglomerative

        self.match = []
        for b in range(batch):
            one = {k: v[b:b + 1] for k, v in self.data.items()}
            md = synthetic.make_matching(one, seed=first_id + b)
            self.match.append((md["part_pcs_by_area"][0].float().contiguous().to(dev), AutoAgglomerative.prepare_matching(md, dev)))
        self.pivot = torch.arange(parts, dtype=torch.int32, device=dev)
        self.pivot_of_point = [self.pivot[mt["point_part"].long()].contiguous() for _, mt in self.match]
        self.E = E

    @torch.no_grad()
    def edge_features(self, x):
        from pfpp_hip import ops
        from puzzlefusion_plusplus.auto_aggl import normalise_edge_hist

        ef = torch.zeros(len(self.match), self.E, 6, dtype=torch.int32, device=x.device)
        for b, (pts, mt) in enumerate(self.match):
            pts_t = ops.pose_apply_points(pts, self.pivot_of_point[b], x[b].contiguous(), normalise=False)
            ef[b, mt["pair_pos"]] = ops.edge_histogram(pts_t, mt["idx_a"], mt["idx_b"], mt["edge_off"], mt["max_m"])
        return normalise_edge_hist(ef)

    @torch.no_grad()
    def step(self):
        m, d = self.model, self.data
        k = self.i % len(self.timesteps)
        if k == 0:
            self.x = self.x0.clone()
        t = self.timesteps[k]
        latent, xyz = m._extract_features(d["part_pcs"], d["part_valids"], self.x)
        eps = m.denoiser(self.x, self.ts_dev[t], latent, xyz, d["part_valids"], d["part_scale"], self.ref)
        self.x = m.noise_scheduler.step(eps, t, self.x, variance_noise=self.noise[k], ref_part=self.ref,
                                        reference=self.reference).prev_sample
        self.edge_feat = self.edge_features(self.x)
        self.last_logits = self.verifier(self.edge_feat, self.edge_idx, self.edge_valid)
        self.last_eps = eps
        self.i += 1

    @torch.no_grad()
    def pred_noise_at_start(self):
        m, d = self.model, self.data
        t = self.timesteps[0]
        latent, xyz = m._extract_features(d["part_pcs"], d["part_valids"], self.x0)
        return m.denoiser(self.x0, self.ts_dev[t], latent, xyz, d["part_valids"], d["part_scale"], self.ref)


def hbm_regime(dev, n_frag: int, points: int, tokens: int, n_params: int = 57_618_183, iters: int = 20):
    """SURVEY.md §8d's second regime: the bandwidth / latency-bound kernels of the step timed alone at the workload's shapes (HIP
    events on the launch stream), algorithmic bytes = what the kernel must read and write once"""
    from pfpp_hip import ops
    from pfpp_hip import train_ops as T

    g = torch.Generator(device=dev).manual_seed(7)
    out = []

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / iters          # us

    def add(name, us, nbytes, **kw):
        out.append(dict(kernel=name, us=round(us, 2), bytes=int(nbytes), GBps=round(nbytes / us * 1e-3, 1),
                        frac=round(nbytes / us * 1e-3 / PEAK_HBM_GBS, 4), **kw))

    pts = (torch.rand(n_frag, points, 3, device=dev, generator=g) * 2 - 1).contiguous()
    pose = torch.randn(n_frag, 7, device=dev, generator=g)
    slot = torch.arange(n_frag, dtype=torch.int32, device=dev)
    add("se3_rotate_gather", timed(lambda: ops.se3_rotate_gather(pts, pose, slot)), n_frag * points * 24 + n_frag * 28)
    levels = ((points, 256, 0.2, 32), (256, 128, 0.4, 64), (128, 25, 0.8, 64))
    cur = pts
    for n_in, s_out, radius, ns in levels:
        us = timed(lambda: ops.fps(cur, s_out))
        add(f"fps N={n_in}->{s_out}", us, n_frag * (n_in * 12 + s_out * 16), us_per_argmax_step=round(us / s_out, 4),
            note="serial chain of dependent argmax steps per fragment: latency-bound by construction")
        _, new_xyz = ops.fps(cur, s_out)
        add(f"ball_query N={n_in} S={s_out} ns={ns}", timed(lambda: ops.ball_query(cur, new_xyz, radius, ns)),
            n_frag * (n_in * 12 + s_out * 12 + s_out * ns * 4))
        cur = new_xyz
    z_e = torch.randn(n_frag, 25, 64, device=dev, generator=g)
    cb = torch.rand(1024, 16, device=dev, generator=g) * 2 - 1
    add("vq_encode", timed(lambda: ops.vq_encode(z_e, cb, slot, n_frag)), n_frag * 25 * 64 * 8 + 1024 * 16 * 4)
    h = torch.randn(tokens, 512, device=dev, generator=g)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    add("layernorm [tokens, 512]", timed(lambda: ops.layernorm(h, gamma=gam, beta=bet)), tokens * 512 * 8)
    xs = [torch.randn(max(1, tokens // 25), 7, device=dev, generator=g) for _ in range(4)]
    coef = (0.5, 0.9, 0.3, 0.6, 0.1)
    add("ddpm_step [fragments, 7]", timed(lambda: ops.ddpm_step(xs[0], xs[1], xs[2], None, None, coef)), xs[0].numel() * 16,
        note="launch-latency bound: a few KB per call")
    n8 = (n_params + 7) // 8 * 8
    p_, g_, m_, v_ = (torch.randn(n8, device=dev, generator=g) * 0.01 for _ in range(4))
    v_.abs_()
    hi, lo = torch.empty(n8, dtype=torch.float16, device=dev), torch.empty(n8, dtype=torch.float16, device=dev)
    add("adamw (57.6 M parameters, planes refreshed)",
        timed(lambda: T.adamw(p_, g_, m_, v_, lr=2e-4, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=1e-6, step=3, hi=hi, lo=lo)),
        n8 * 32)          # reads p, g, m, v; writes p, m, v and the two fp16 planes
    return out


def cpu_protocol(kind: str, budget_s: float = 45.0):
    """BASELINE.md §3 on BASELINE.json configs[0] (1 puzzle, 8 fragments x 512 points, 20-step schedule, fixed injected noise) with the
    CPU oracle (a port of the reference path, validated against the imported reference: tools/make_goldens.py): thread count
    swept upward over {16, 32, physical cores} while it keeps paying off (1 warm-up + 2 timed steps each), then 3 warm-up + up to 10 timed steps at the best
    setting (median and min), a one-step single-thread figure, and the encoder / transformer / scheduler split — all inside
    `budget_s` of CPU time; whatever the budget cut short is said in `sample`."""
    from oracle import pfpp_oracle as O
    from oracle import weights
    from pfpp_hip import synthetic

    t_begin = time.perf_counter()
    enc_sd, den_sd = weights.vqvae_state_dict(), weights.denoiser_state_dict()
    batch = synthetic.make_batch(0, 1, num_points=512, num_parts=8)
    g = torch.Generator().manual_seed(0)
    sched = O.PiecewiseSchedule()
    sched.set_timesteps(20)
    ts_list = sched.timesteps.tolist()
    ref = batch["ref_part"].bool()
    gt = torch.cat([batch["part_trans"], batch["part_rots"]], dim=-1).float()
    x0 = torch.randn(1, 20, 7, generator=g)
    noises = [torch.randn(1, 20, 7, generator=g) for _ in range(20)]
    train = kind == "train"
    if train:
        names = [k for k, v in den_sd.items() if v.dtype.is_floating_point and k != "pos_encoding.pe"]
        sd = {k: v.clone() for k, v in den_sd.items()}
        mom = [torch.zeros_like(sd[n]) for n in names]
        var = [torch.zeros_like(sd[n]) for n in names]
    state = {"x": x0.clone(), "i": 0}
    split = {"encoder": 0.0, "transformer": 0.0, "scheduler": 0.0}

    def one():
        i = state["i"]
        t0 = time.perf_counter()
        if train:
            noise = noises[i % 20]
            t = torch.tensor([ts_list[i % 20]])
            noisy = sched.add_noise(gt, noise, t)
            noisy[ref] = gt[ref]
            with torch.no_grad():
                lat, xyz = O.extract_features(enc_sd, batch["part_pcs"], batch["part_valids"], noisy)
            t1 = time.perf_counter()
            req = {k: (w.clone().requires_grad_(True) if k in names else w) for k, w in sd.items()}
            loss = O.denoiser_loss(O.denoiser_forward(req, noisy, t, lat, xyz, batch["part_valids"], batch["part_scale"], ref),
                                   noise, batch["part_valids"], ref)
            loss.backward()
            t2 = time.perf_counter()
            with torch.no_grad():
                O.adamw_step([sd[n] for n in names], [req[n].grad for n in names], mom, var, i + 1)
            t3 = time.perf_counter()
        else:
            t = ts_list[i % 20]
            x = state["x"]
            lat, xyz = O.extract_features(enc_sd, batch["part_pcs"], batch["part_valids"], x)
            t1 = time.perf_counter()
            eps = O.denoiser_forward(den_sd, x, torch.full((1,), t, dtype=torch.int64), lat, xyz, batch["part_valids"], batch["part_scale"], ref)
            t2 = time.perf_counter()
            state["x"] = sched.step(eps, t, x, noises[i % 20])
            t3 = time.perf_counter()
        split["encoder"] += t1 - t0; split["transformer"] += t2 - t1; split["scheduler"] += t3 - t2
        state["i"] = i + 1
        return t3 - t0

    ncpu = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:  # noqa: BLE001
        phys = ncpu
    # ascending; a larger count is only tried while the previous one still paid off and the budget allows (oversubscribing a
    # 1-puzzle step with hundreds of threads costs tens of seconds per step)
    cands = sorted({c for c in (16, 32, phys) if 1 <= c <= ncpu})
    sweep = {}
    for c in cands:
        if sweep and (time.perf_counter() - t_begin > 0.3 * budget_s or sweep[max(sweep)] > 1.2 * min(sweep.values())):
            break
        torch.set_num_threads(c)
        one()
        sweep[c] = min(one(), one())
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for k in split:
        split[k] = 0.0
    warm = 0
    while warm < 3 and time.perf_counter() - t_begin < 0.45 * budget_s:
        one(); warm += 1
    for k in split:
        split[k] = 0.0
    times = []
    while len(times) < 10 and time.perf_counter() - t_begin < 0.8 * budget_s:
        times.append(one())
    tot = sum(split.values()) or 1.0
    shares = {k: round(v / tot, 3) for k, v in split.items()}
    single = None
    if time.perf_counter() - t_begin < 0.85 * budget_s:
        torch.set_num_threads(1)
        single = one()
    torch.set_num_threads(best)
    times.sort()
    median = times[len(times) // 2] if times else sweep[best]
    tmin = times[0] if times else sweep[best]
    what = "training iterations (forward, autograd backward, AdamW)" if train else "DDPM sampler steps"
    return {
        "value": round(8 / median, 3), "unit": "fragment*steps/s", "cores": best, "kind": "port",
        "median_s_per_step": round(median, 4), "min_s_per_step": round(tmin, 4), "best_value": round(8 / tmin, 3),
        "thread_sweep_s_per_step": {str(c): round(v, 4) for c, v in sweep.items()},
        "single_thread": None if single is None else {"s_per_step": round(single, 3), "value": round(8 / single, 3)},
        "split": shares,
        "sample": f"{warm} warm-up + {len(times)} timed {what} of 1 puzzle, 8 fragments x 512 pts (BASELINE configs[0]) at the best of "
                  f"{sorted(sweep)} torch threads (= {best}; candidates {cands}) on {ncpu} logical CPUs; BASELINE.md §3 asks for 3 + 10, the {budget_s:.0f} s budget "
                  f"allowed {warm} + {len(times)}; median over the timed steps; split = share of "
                  + ("encode / forward+backward / AdamW" if train else "encode / transformer / scheduler step") + " time",
    }


# kernel family -> (arithmetic, index of the template flag that switches the family to single-pass fp16 or None).  The roofline's peak
# follows from this table and nothing else: a renamed or new family raises instead of silently taking another family's ceiling.
KERNEL_FAMILIES = {
    "gemm_pl_kernel": ("f16x3", 9), "gemm_pl_dwgroup_kernel": ("f16x3", None), "gemm_pl64_kernel": ("f16x3", None),
    "gemm_wd_kernel": ("f16x3", 3), "gemm_wd_pf_kernel": ("f16x3", None),
    "gemm_f16x3_kernel": ("f16x3", None), "gemm_f16x3_deep_kernel": ("f16x3", None), "gemm_f16x3_apre_kernel": ("f16x3", None), "gemm_f32_mfma_kernel": ("f32", None),
    "gemm_grad_kernel": ("f16x3", None), "gemm_grad_group_kernel": ("f16x3", None),
    "sa1_train_kernel": ("f16x3", None), "sa2_train_kernel": ("f16x3", None), "sa_rows_train_kernel": ("f16x3", None),
    "sa_rows8_train_kernel": ("f16x3", None), "sa_wide_train_kernel": ("f16x3", None), "sa_mlp3_kernel": ("f16x3", None),
    "sa_mlp2_kernel": ("f16x3", None), "sa_first_stats_kernel": ("f16x3", None),
}


def kernel_arith(name: str) -> str:
    """arithmetic class of a traced matrix kernel from its family name and template flags (KERNEL_FAMILIES)"""
    import re

    m = re.search(r"(\w+?)(?:<([^()]*)>)?(?:\(|\+|$)", name.split("::")[-1].strip())
    fam = m.group(1) if m else name
    if fam not in KERNEL_FAMILIES:
        raise ValueError(f"bench.py: kernel family {fam!r} (from {name!r}) is not in KERNEL_FAMILIES — add it with its arithmetic before quoting a roofline")
    arith, x1 = KERNEL_FAMILIES[fam]
    flags = [f.strip() for f in (m.group(2) or "").split(",")]
    if x1 is not None and len(flags) > x1 and flags[x1] == "true":
        return "f16"
    return arith


def back_to_back_wd(trace, name: str, reps: int = 30):
    """The dominant weight-direct GEMM re-timed WITHOUT the event pair around every launch: for each (M, N, K) the traced pass saw under
    `name`, `reps` launches of pfpp_gemm_wd on random operands of that shape between ONE pair of events, weighted by how often the
    step launches the shape.  L2-warm weights, nothing else on the chip: the optimistic end of the bracket whose pessimistic end
    is the per-launch event average (which contains part of the ~8 us event-pair floor)."""
    from pfpp_hip import ops, packing

    shapes = {}
    for _e0, _e1, flops, nm, shape in trace:
        if nm == name:
            shapes[shape[:3]] = shapes.get(shape[:3], 0) + 1
    if not shapes or not name.startswith("gemm_wd"):
        return None
    tot_ms = tot_n = 0.0
    per = {}
    for (M, N, K), count in shapes.items():
        a = ops.SplitAct(*packing.split_f16(torch.randn(M, K, device="cuda")))
        w = packing.PW(torch.randn(N, K, device="cuda") / K ** 0.5)
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm_wd(a, w, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_wd(a, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        per[f"{M}x{N}x{K}"] = round(ms, 4)
        tot_ms += ms * count
        tot_n += count
    return {"avg_launch_ms": round(tot_ms / tot_n, 4), "per_shape_ms": per}


def roofline_from_trace(trace, steps: int, mode: str, serialised: bool = False):
    """`roofline` object of the JSON line from HIP-event timed MFMA-kernel launches (ops.GEMM_TRACE entries: start / end event,
    algorithmic FLOPs, kernel name, shape): the variant with the largest summed duration against its MFMA ceiling"""
    from pfpp_hip import ops  # noqa: F401

    per = {}
    shapes = {}
    for e0, e1, flops, name, shape in trace:
        ms = e0.elapsed_time(e1)
        sa = shapes.setdefault((name,) + shape, [0.0, 0.0, 0])
        sa[0] += flops; sa[1] += ms; sa[2] += 1
        a = per.setdefault(name, [0.0, 0.0, 0])
        a[0] += flops; a[1] += ms; a[2] += 1
    if os.environ.get("BENCH_GEMM_SHAPES"):
        for key, (fl, ms_, n_) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
            print(f"  {ms_ / steps:8.3f} ms/step  {fl / (ms_ * 1e-3) / 1e12:7.1f} TF/s  x{n_ // steps:3d}  {key}", file=sys.stderr)
    name, (flops, ms, cnt) = max(per.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12          # algorithmic 2*M*N*K of the launches / their duration
    arith = kernel_arith(name)                     # "f16x3" | "f16" | "f32": from the kernel family's own template flags, loud when unknown
    single, split = arith == "f16", arith == "f16x3"
    # the split path spends 3 f16 matrix FLOPs per algorithmic FLOP: its ceiling for algorithmic
    # FLOPs is the f16 dense peak / 3
    peak = PEAK_F16_MFMA_TFLOPS if single else (PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS)
    traffic, traffic_src = pmc_traffic(name.split("+")[0].split("(")[0], mode)
    # what a HIP-event pair reads around a launch that does (almost) nothing: the start marker's completion -> dispatch -> a one-element
    # kernel -> end marker (10 us seen; a 20 us kernel hides all but ~2 us of it).  rocprofv3's kernel durations do not contain it, the averages below do
    floor_us = None
    try:
        one = torch.zeros(1, device="cuda")
        pairs = []
        for _ in range(80):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(); one.fill_(1.0); b_.record()
            pairs.append((a_, b_))
        torch.cuda.synchronize()
        floor_us = round(sorted(a_.elapsed_time(b_) for a_, b_ in pairs[16:])[len(pairs[16:]) // 2] * 1e3, 2)
    except Exception:      # noqa: BLE001 (a diagnostic field only)
        pass
    try:
        b2b = back_to_back_wd(trace, name)
    except Exception as exc:      # noqa: BLE001 (a diagnostic field only)
        b2b = {"error": str(exc)[:200]}
    roofline = {
        "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
        "peak_note": ("f16 dense MFMA peak 2500 TFLOP/s (single-pass fp16)" if single else
                      "f16 dense MFMA peak 2500 TFLOP/s / 3 matrix instructions per fp32-grade product" if split else "fp32 MFMA dense peak"),
        "mfma_tflops_executed": round(achieved * (3 if split else 1), 1),
        "launches_per_step": cnt / steps, "avg_launch_ms": round(ms / cnt, 4),
        "avg_launch_ms_note": "mean of the per-launch HIP-event pairs (contains part of event_pair_floor_us); rocprofv3's kernel durations and "
                              "back_to_back bracket it from below",
        "back_to_back": b2b,
        "frac_back_to_back": (None if not (b2b and "avg_launch_ms" in b2b) else round(flops / cnt / (b2b["avg_launch_ms"] * 1e-3) / 1e12 / peak, 4)),
        "traffic_measured_in_run": False,
        "traffic_box": "the builder's gpurun box, committed counter passes under profiles/ (rocprofv3 is not run inside the driver's bench)",
        "measured": "HIP events around every launch in a second pass over the same steps" + (", streams serialised (python bench.py --serial reproduces it under rocprofv3)" if serialised else ""),
        "event_pair_floor_us": floor_us,
        "variants": {k: {"launches_per_step": round(v[2] / steps, 1), "avg_launch_ms": round(v[1] / v[2], 4),
                         "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1)}
                     for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
        # the other roof of the same kernel: counter traffic per launch / its duration against the 8 TB/s HBM peak (short-K GEMMs on
        # fp32 activations of 1.26 M rows sit between the two roofs)
        "hbm_frac_of_traffic": (round(traffic / (ms / cnt * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if traffic else None),
        "gemm_ms_per_step_all_variants": round(sum(v[1] for v in per.values()) / steps, 3),
        "gemm_tflops_all_variants": round(sum(v[0] for v in per.values()) / (sum(v[1] for v in per.values()) * 1e-3) / 1e12, 2),
    }
    return roofline


def aggl_puzzles_per_s(dev, n_puzzles: int = 3, points: int = 1000, in_flight: int = 1):
    """BASELINE configs[2]-shaped: the full auto-agglomerative loop (denoise -> edge features -> verify -> promote/merge,
    auto_aggl.py:86-318) on single puzzles (batch 1 like the reference's test.py), 20 DDPM steps per outer iteration,
    up to cfg.verifier.max_iters = 6 iterations, synthetic matching data; random-init weights"""
    from pfpp_hip import config, synthetic
    from puzzlefusion_plusplus.auto_aggl import AutoAgglomerative

    torch.manual_seed(4321)
    model = AutoAgglomerative(config.auto_aggl_config()).to(dev).eval()
    with torch.no_grad():
        model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
    puzzles = []
    for i in range(n_puzzles + 1):
        b = {k: v.to(dev) for k, v in synthetic.make_batch(500 + i, 1, num_points=points).items()}
        b.update(synthetic.make_matching(b, seed=i))
        puzzles.append(b)
    model.test_step(puzzles[0])                     # warm-up
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    steps = frags = 0
    if in_flight <= 1:
        for b in puzzles[1:]:
            out = model.test_step(b)
            steps += out["steps"]
            frags += int(b["num_parts"][0]) * out["steps"]
    else:
        for i in range(1, len(puzzles), in_flight):
            group = puzzles[i:i + in_flight]
            for b, out in zip(group, model.test_batch(group)):
                steps += out["steps"]
                frags += int(b["num_parts"][0]) * out["steps"]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"value": round(n_puzzles / dt, 3), "unit": "puzzles/s", "puzzles": n_puzzles, "in_flight": max(1, in_flight),
            "ddpm_steps": steps, "fragment_steps_per_s": round(frags / dt, 1),
            "note": ("one puzzle in flight like the reference's test.py (latency-bound)" if in_flight <= 1 else
                     "independent puzzles batched through the loop (AutoAgglomerative.test_batch); per-puzzle results as in test_step")
                    + "; full loop incl. verifier, promotion, merges and metrics"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, the launch of
    scripts/train_denoiser.sh:6-7 `+trainer.devices=N +trainer.strategy=ddp`) by re-running this file under
    torch.distributed.run on 127.0.0.1; rank 0 of the children prints the one JSON line.  Fails loudly when the box
    has fewer than N devices (PFPP_BENCH_BACKEND=gloo lets ranks share devices: the single-GPU test of this path)."""
    import socket
    import subprocess

    backend = os.environ.get("PFPP_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; one rank per GPU is required "
                         "(RCCL cannot place two ranks on one device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the only implementation of the path)")
    # PFPP_BENCH_BACKEND=gloo: the N > 1 path with every rank on whatever GPUs exist (ranks may share one) — how the
    # multi-rank logic is exercised on a single-GPU box; the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("PFPP_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE {world} rank(s)")

    from pfpp_hip import ops

    train = args.mode == "train"
    stress = args.mode == "stress"
    # N > 1: the job's puzzles (the same 1000 * r + i ids the ranks used to take in order) are dealt to the ranks by valid-fragment
    # count — the encoder's work per puzzle varies ~10x with it and the step ends with the slowest rank
    ids, balance = None, None
    if world > 1 and not stress and args.parts is None and os.environ.get("PFPP_BENCH_BALANCE", "1") == "1":
        from pfpp_hip import synthetic
        from pfpp_hip.parallel import balanced_assignment

        pool = [1000 * r + i for r in range(world) for i in range(args.batch)]
        counts = [synthetic.num_parts_of(i) for i in pool]
        assign = balanced_assignment(counts, world, equal_count=True)
        ids = [pool[j] for j in assign[rank]]
        loads = [sum(counts[j] for j in a) for a in assign]
        naive = [sum(counts[r * args.batch:(r + 1) * args.batch]) for r in range(world)]
        balance = {"fragments_per_rank": loads, "in_order_would_be": naive}
    if stress:
        ops.SINGLE_PASS = os.environ.get("PFPP_STRESS_F16X3", "0") != "1"       # single-pass fp16 on the plane GEMMs (configs[4])
        wl = StressWorkload(args.stress_batch, first_id=1000 * rank, dev=dev)
    elif train:
        wl = TrainWorkload(args.batch, args.points, args.parts, first_id=1000 * rank, dev=dev, latents_given=args.latents_given,
                           pipeline=not (args.no_pipeline or args.serial), ids=ids)
        if args.serial:
            wl.engine.single_stream()
    else:
        wl = SamplerWorkload(args.batch, args.points, args.parts, first_id=1000 * rank, dev=dev, compact=args.compact, ids=ids)
    for _ in range(args.warmup):
        wl.step()

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
            torch.cuda.synchronize(dev)

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    sync_all()
    elapsed = time.perf_counter() - t0
    frag_steps = float(wl.n_frag * args.steps)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        ff = torch.tensor([frag_steps], dtype=torch.float64, device=dev)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
        frag_steps = ff.item()

    # ---- roofline of the dominant kernel: second, instrumented pass over the same steps ----------
    roofline = None
    if not args.no_roofline:
        if train:
            # per-kernel durations are taken with the streams serialised (no side stream, encoder in line): bracketing
            # events on a stream that shares the chip with another stream measure the contention too, not the kernel
            wl.engine.single_stream()
            wl.pipeline = None
            wl.step()
            torch.cuda.synchronize(dev)
        ops.GEMM_TRACE = []
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize(dev)
        trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
        roofline = roofline_from_trace(trace, args.steps, "train" if train else ("stress" if stress else "sampler"), serialised=train)

    extra = {}
    if dist is not None:
        # proof that the collective library saw every rank: sum of ones over the job's ranks and the set of devices they sit on
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)
        devs = [None] * world
        dist.all_gather_object(devs, (torch.cuda.current_device(), torch.cuda.get_device_properties(dev).name))
        extra["rccl_ranks" if backend == "nccl" else f"{backend}_ranks"] = int(ones.item())
        extra["rank_devices"] = [f"cuda:{d} {n}" for d, n in devs]
        extra["backend"] = backend
    if balance is not None:
        extra["rank_balance"] = balance
    if dist is not None and train and world > 1:
        # the step's one exchange, alone: all-reduce of the flat gradient buffer (what GradExchange sends in 6 layer slices + 2)
        gbuf = wl.engine.flat.grads
        for _ in range(2):
            dist.all_reduce(gbuf)
        sync_all()
        t_ar = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(gbuf)
        sync_all()
        t_ar = (time.perf_counter() - t_ar) / 5
        nbytes = gbuf.numel() * 4
        extra["gradient_all_reduce"] = {"bytes": nbytes, "ms": round(t_ar * 1e3, 3),
                                        "algbw_GBps": round(nbytes / t_ar * 1e-9, 1),
                                        "busbw_GBps": round(nbytes / t_ar * 1e-9 * 2 * (world - 1) / world, 1),
                                        "note": "stand-alone, not overlapped; in the step it runs per layer under the backward"}
    roofline_hbm = None
    if rank == 0 and not args.no_roofline:
        if stress:
            roofline_hbm = hbm_regime(dev, wl.n_frag, 2048, wl.n_frag * 25)
        else:
            roofline_hbm = hbm_regime(dev, wl.n_frag, args.points, wl.n_frag * 25 if (train or args.compact) else args.batch * 20 * 25)
    if stress and rank == 0:
        # accuracy of the perf mode: the same first sampler step in the parity arithmetic (split-f16, 3 matrix instructions per product)
        eps_fast = wl.pred_noise_at_start()
        ops.SINGLE_PASS = False
        eps_ref = wl.pred_noise_at_start()
        ops.SINGLE_PASS = os.environ.get("PFPP_STRESS_F16X3", "0") != "1"
        v = wl.data["part_valids"].bool()
        extra["single_pass_vs_f16x3"] = {"max_abs_diff_pred_noise": float((eps_fast - eps_ref)[v].abs().max()),
                                         "max_abs_pred_noise": float(eps_ref[v].abs().max()),
                                         "note": "first sampler step, same inputs and weights; f16x3 is the parity mode (1e-4 vs the CPU oracle), "
                                                 "single-pass fp16 the perf mode of BASELINE configs[4]"}
        extra["verifier_edges_per_puzzle"] = int(wl.edge_feat.shape[1])
    if train and rank == 0:
        extra["final_loss"] = round(float(wl.last_loss), 5)
    if train and rank == 0 and world == 1 and not args.no_roofline and not args.latents_given:
        # BASELINE configs[1] reads "VQ-VAE latents precomputed": the same iteration without the encoder (latents of the clean
        # pose, computed once) — not a mode of the reference (its latents depend on the noisy rotation, denoiser.py:66-71)
        del wl.engine
        lwl = TrainWorkload(args.batch, args.points, args.parts, first_id=1000 * rank, dev=dev, latents_given=True)
        for _ in range(args.warmup):
            lwl.step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            lwl.step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        extra["train_latents_given"] = {"value": round(lwl.n_frag * args.steps / dt, 2), "unit": "fragment*steps/s",
                                        "ms_per_step": round(dt / args.steps * 1e3, 3),
                                        "note": "transformer-only training iteration (forward, loss, backward, AdamW), encoder skipped"}
        del lwl.engine, lwl
    if train and rank == 0 and world == 1 and not args.no_roofline:
        # the inference sampler step at the same shape (Denoiser.validation_step loop body), padded slots
        # evaluated like the reference and dropped
        if hasattr(wl, "engine"):
            del wl.engine
        swl = SamplerWorkload(args.batch, args.points, args.parts, first_id=1000 * rank, dev=dev)
        for name, compact in (("sampler_step", False), ("sampler_step_compact", True)):
            swl.model.denoiser.compact_padded = compact
            for _ in range(args.warmup):
                swl.step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                swl.step()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            extra[name] = {"value": round(swl.n_frag * args.steps / dt, 2), "unit": "fragment*steps/s",
                           "ms_per_step": round(dt / args.steps * 1e3, 3),
                           "padded_slots": "dropped" if compact else "evaluated like the reference"}
        # roofline of the sampler step's dominant MFMA kernel (instrumented pass, all slots evaluated like the reference)
        swl.model.denoiser.compact_padded = False
        ops.GEMM_TRACE = []
        n_r = max(2, min(args.steps, 8))
        for _ in range(n_r):
            swl.step()
        torch.cuda.synchronize(dev)
        trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
        extra["sampler_roofline"] = roofline_from_trace(trace, n_r, "sampler")
        del swl
        # BASELINE configs[4] on this one GPU: the joint denoiser + verifier step at 100 fragments x 2048 points in its fp16-MFMA mode
        # (python bench.py --mode stress is the same workload as the headline line)
        prev_sp = ops.SINGLE_PASS
        try:
            ops.SINGLE_PASS = True
            st = StressWorkload(args.stress_batch, first_id=0, dev=dev)
            for _ in range(2):
                st.step()
            torch.cuda.synchronize(dev)
            n_s = 6
            t1 = time.perf_counter()
            for _ in range(n_s):
                st.step()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            ops.GEMM_TRACE = []
            for _ in range(n_s):
                st.step()
            torch.cuda.synchronize(dev)
            trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
            eps_fast = st.pred_noise_at_start()
            ops.SINGLE_PASS = False
            eps_ref = st.pred_noise_at_start()
            v = st.data["part_valids"].bool()
            extra["stress"] = {
                "workload": f"BASELINE configs[4], 1 GPU: {args.stress_batch} puzzles x 100 fragments x 2048 points, rotate + encode + "
                            "DenoiserTransformer + scheduler step + edge features (pose apply + matched-point histograms) + VerifierTransformer on 4,950 edges "
                            "per puzzle; plane GEMMs and attention forward single-pass fp16",
                "ms_per_step": round(dt / n_s * 1e3, 3), "value": round(st.n_frag * n_s / dt, 2), "unit": "fragment*steps/s",
                "roofline": roofline_from_trace(trace, n_s, "stress"),
                "max_abs_diff_pred_noise_vs_f16x3": float((eps_fast - eps_ref)[v].abs().max()),
                "max_abs_pred_noise": float(eps_ref[v].abs().max()),
            }
            del st
        finally:
            ops.SINGLE_PASS = prev_sp
            ops.GEMM_TRACE = None
        extra["auto_aggl_full_loop"] = aggl_puzzles_per_s(dev)
        extra["auto_aggl_full_loop_batched"] = aggl_puzzles_per_s(dev, n_puzzles=64, in_flight=32)
    if not train and not stress and rank == 0 and world == 1 and not args.compact and not args.no_roofline:
        # the same K steps with the padded fragment slots dropped (outputs of valid fragments unchanged)
        wl.model.denoiser.compact_padded = True
        for _ in range(args.warmup):
            wl.step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        wl.model.denoiser.compact_padded = False
        extra["compact_mode"] = {"value": round(wl.n_frag * args.steps / dt, 2), "unit": "fragment*steps/s",
                                 "ms_per_step": round(dt / args.steps * 1e3, 3),
                                 "note": "padded fragment slots dropped in the transformer; identical predictions for valid fragments"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if train:
            # the line's own workload (training iterations) on the CPU port, and next to it BASELINE.md section 3's protocol as written
            # (the 20-step eval sampler on configs[0]) — the figure the plan names; both bounded, both on this host in this run
            cpu = cpu_protocol("train", budget_s=28.0)
            cpu["sampler_protocol"] = cpu_protocol("sample", budget_s=18.0)
            # BASELINE.md section 3's own figure (20-step eval sampler on configs[0]) where a flat parser finds it
            cpu["sampler_value"] = cpu["sampler_protocol"]["value"]
            cpu["sampler_unit"] = cpu["sampler_protocol"]["unit"]
        else:
            cpu = cpu_protocol("sample")

    if rank == 0:
        line = {
            "metric": "denoiser DDPM-step throughput (fragment*steps/s)",
            "value": round(frag_steps / elapsed, 2), "unit": "fragment*steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 I/O (plane GEMMs and attention forward: single-pass f16, fp32 accumulate; others f16x3)" if (stress and ops.SINGLE_PASS) else
                      "f32 (GEMMs: %s)" % ops.GEMM_MODE), "data": "synthetic",
            "config": {
                "workload": "stress, BASELINE configs[4]: joint step = rotate + PointNet++/VQ encode + DenoiserTransformer + scheduler step + "
                            "edge features + VerifierTransformer on all candidate edges; 100 fragments per puzzle x 2048 points (beyond the reference's "
                            "max_len = 20: no reference parity, roofline only)" if stress else
                            ("DDPM training iteration, BASELINE configs[1]: add_noise + rotate + frozen PointNet++/VQ encode"
                             + (" (skipped: latents given)" if args.latents_given else " (in the loop)") +
                             " + DenoiserTransformer forward (dropouts on) + MSE + full backward + "
                             + ("" if args.no_pipeline or args.latents_given else "[encoder of iteration i+1 issued on its own stream during iteration i] ")
                             + ("RCCL gradient all-reduce + " if world > 1 else "") + "AdamW") if train else
                            ("DDPM sampler step, encoder in the loop (rotate+PointNet++/VQ encode+DenoiserTransformer+"
                             "scheduler step), BASELINE configs[1] shape, inference forward"),
                "padded_slots": ("not evaluated (their gradient contribution is exactly zero)" if train else
                                 "dropped (compact mode)" if args.compact else "evaluated like the reference"),
                "puzzles_per_gpu": args.stress_batch if stress else args.batch, "fragment_slots": 100 if stress else 20,
                "points_per_fragment": 2048 if stress else args.points,
                "valid_fragments_per_gpu": wl.n_frag,
                "puzzle_steps_per_s": round((args.stress_batch if stress else args.batch) * world * args.steps / elapsed, 2),
                "weights": "random init, reference architecture (57.6M denoiser + 0.6M encoder params)",
                "parallelism": (f"data parallel x {world} GPU(s): puzzles sharded, gradients all-reduced (RCCL) per layer "
                                "during the backward" if train else
                                f"independent puzzles x {world} GPU(s), no data-path collective"),
            },
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
