"""AutoAgglomerative (drop-in for puzzlefusion_plusplus/auto_aggl.py): the denoise -> verify loop of
`test_step` on the HIP kernels, device resident.

What runs where (reference line numbers refer to auto_aggl.py):
  * inner loop, :136-151 — 20 DDPM steps per outer iteration: rotate+encode (pfpp_hip.encoder), denoise
    (pfpp_hip.denoiser), ancestral step with the re-pin of the reference fragments fused in, and the
    composed pose record of `get_param` (pfpp_pose_compose) — no host round trip per step (the
    reference does a `.cpu()` every step, :151).
  * edge features, :156-201 — by-area points moved by the predicted poses (pfpp_pose_apply_points),
    per-edge bidirectional nearest-neighbour histogram (pfpp_edge_histogram), normalisation.
  * verifier + threshold, :203-205 — pfpp_hip.verifier.
  * reference-part promotion, :208-222 and the early-exit tests — host logic on a handful of booleans.
  * node merging, :224-286 — NOT implemented in this round (normal estimation + intersect removal +
    random-start FPS): `merge_fn` may be supplied by the caller; without it the
    fragments stay separate (poses of promoted fragments are still frozen as in the reference).
Batch size 1, as in the reference (docs/test.md:8).
"""
from __future__ import annotations

import itertools
from typing import Callable, List, Optional

import numpy as np
import torch

from pfpp_hip import ops
from pfpp_hip.lightning_compat import LightningModule
from pfpp_hip.scheduler import PiecewiseScheduler
from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer
from puzzlefusion_plusplus.verifier.model.modules.verifier_transformer import VerifierTransformer
from puzzlefusion_plusplus.vqvae.model.modules.vq_vae import VQVAE


class AutoAgglomerative(LightningModule):
    def __init__(self, cfg, merge_fn: Optional[Callable] = None):
        super().__init__()
        self.cfg = cfg
        self.denoiser = DenoiserTransformer(cfg.denoiser)
        self.verifier = VerifierTransformer(cfg.verifier)
        self.encoder = VQVAE(cfg.ae)
        self.save_hyperparameters()
        m = cfg.denoiser.model
        self.noise_scheduler = PiecewiseScheduler(
            num_train_timesteps=m.DDPM_TRAIN_STEPS, beta_schedule=m.DDPM_BETA_SCHEDULE, prediction_type=m.PREDICT_TYPE,
            beta_start=m.BETA_START, beta_end=m.BETA_END, clip_sample=False, timestep_spacing=m.timestep_spacing)
        self.num_points = m.num_point
        self.num_channels = m.num_dim
        self.noise_scheduler.set_timesteps(num_inference_steps=m.num_inference_steps)
        self.merge_fn = merge_fn

    # ------------------------------------------------------------------ helpers
    def _extract_features(self, part_pcs, part_valids, x):
        return self.encoder.extract_features(part_pcs, part_valids, x)

    @staticmethod
    def _edge_mask(num_parts: torch.Tensor, P: int) -> torch.Tensor:
        e = torch.tensor(list(itertools.combinations(range(P), 2)), dtype=torch.int64, device=num_parts.device)
        return (e[None, :, 0] < num_parts[:, None]) & (e[None, :, 1] < num_parts[:, None])

    @staticmethod
    def prepare_matching(data_dict, device):
        """flatten the per-edge correspondence lists of the matching data into index arrays once:
        global index of a matched point = start(part) + critical_pcs_idx[start(part) + corr]
        (get_distance_for_matching_pts, node_merge_utils.py:62-89)"""
        n_pcs = data_dict["n_pcs"][0].cpu().long()
        crit = data_dict["critical_pcs_idx"][0].cpu().long()
        edges = data_dict["edges"][0].cpu().long()
        start = torch.cumsum(n_pcs, 0) - n_pcs
        ia, ib, off, pairs = [], [], [0], []
        for e in range(edges.shape[0]):
            idx2, idx1 = int(edges[e, 0]), int(edges[e, 1])
            corr = torch.as_tensor(data_dict["correspondences"][e]).reshape(-1, 2).long().cpu()
            a = start[idx1] + crit[start[idx1] + corr[:, 0]]
            b = start[idx2] + crit[start[idx2] + corr[:, 1]]
            ia.append(a); ib.append(b); off.append(off[-1] + corr.shape[0]); pairs.append((idx1, idx2))
        point_part = torch.repeat_interleave(torch.arange(n_pcs.numel()), n_pcs)
        cat = lambda xs: (torch.cat(xs) if xs else torch.zeros(0, dtype=torch.long)).to(torch.int32).to(device)
        return {
            "idx_a": cat(ia), "idx_b": cat(ib), "edge_off": torch.tensor(off, dtype=torch.int32, device=device),
            "max_m": max([off[i + 1] - off[i] for i in range(len(off) - 1)], default=0), "pairs": pairs,
            "point_part": point_part.to(torch.int32).to(device),
        }

    # ------------------------------------------------------------------ test_step
    @torch.no_grad()
    def test_step(self, data_dict, idx=0, x_init: Optional[torch.Tensor] = None, noises: Optional[List[torch.Tensor]] = None):
        dev = data_dict["part_pcs"].device
        gt = torch.cat([data_dict["part_trans"], data_dict["part_rots"]], dim=-1).float().contiguous()
        B, P, N, _ = data_dict["part_pcs"].shape
        if B != 1:
            raise ValueError("AutoAgglomerative.test_step handles one puzzle per call (docs/test.md:8)")
        x = torch.randn(gt.shape, device=dev) if x_init is None else x_init.clone()
        ref_part = data_dict["ref_part"].clone()
        reference = torch.zeros_like(gt)
        reference[ref_part] = gt[ref_part]
        x[ref_part] = reference[ref_part]
        part_valids = data_dict["part_valids"].clone()
        part_scale = data_dict["part_scale"].clone()
        part_pcs = data_dict["part_pcs"].clone()
        num_parts = data_dict["num_parts"].clone()
        n_nodes = int(num_parts[0])
        nodes = [{"pivot": i, "valids": True, "ref_part": False, "init_pose": None} for i in range(n_nodes)]
        nodes[int(torch.where(ref_part)[1][0])]["ref_part"] = True
        classified = torch.zeros_like(part_valids, dtype=torch.bool)
        have_matching = "edges" in data_dict
        match = self.prepare_matching(data_dict, dev) if have_matching else None
        pivot = torch.arange(n_nodes, dtype=torch.int32, device=dev)
        edge_indices = torch.triu(torch.ones(P, P, dtype=torch.bool, device=dev), diagonal=1).nonzero(as_tuple=False)[None]
        edge_valids = self._edge_mask(num_parts, P)
        max_iters = self.cfg.verifier.max_iters
        traj, step_no, verifier_calls = [], 0, 0
        for it in range(max_iters):
            for t in self.noise_scheduler.timesteps.tolist():
                ts = torch.full((B,), t, dtype=torch.int64, device=dev)
                latent, xyz = self._extract_features(part_pcs, part_valids, x)
                eps = self.denoiser(x, ts, latent, xyz, part_valids, part_scale, ref_part)
                x = self.noise_scheduler.step(eps, t, x, variance_noise=None if noises is None else noises[step_no],
                                              ref_part=ref_part, reference=reference).prev_sample
                traj.append(ops.pose_compose(x[0].contiguous(), pivot))      # get_param (:151), stays on the GPU
                step_no += 1
            if it + 1 == max_iters or not have_matching:
                break
            # ---- edge features (:156-201) ---------------------------------------------------------
            pts_t = ops.pose_apply_points(data_dict["part_pcs_by_area"][0].float().contiguous(),
                                          pivot[match["point_part"].long()].contiguous(), x[0].contiguous(), normalise=False)
            hist = ops.edge_histogram(pts_t, match["idx_a"], match["idx_b"], match["edge_off"], match["max_m"])
            ef = torch.zeros(1, P, P, 6, dtype=torch.int32, device=dev)
            if match["pairs"]:
                i1 = torch.tensor([p[0] for p in match["pairs"]], device=dev)
                i2 = torch.tensor([p[1] for p in match["pairs"]], device=dev)
                ef[0, i1, i2] = hist
            mat_mask = torch.triu(torch.ones(P, P, dtype=torch.bool, device=dev), diagonal=1)
            ef = ef[:, mat_mask]
            cnt = ef.sum(dim=-1, keepdim=True)
            ef = torch.cat((ef / torch.where(cnt == 0, 1, cnt), cnt), dim=-1).float()
            # ---- verifier (:203-205) ----------------------------------------------------------------
            logits = self.verifier(ef.contiguous(), edge_indices.contiguous(), edge_valids)
            verifier_calls += 1
            pred = (torch.sigmoid(logits) > self.cfg.verifier.threshold).squeeze(-1) & edge_valids
            classified_edges = edge_indices[pred].cpu().tolist()
            # ---- reference promotion (:208-222) -----------------------------------------------------
            valid_b = part_valids.bool()
            ref_idx = set(torch.where(ref_part)[1].cpu().tolist())
            classified[0, list(ref_idx)] = True
            larger = valid_b & (part_scale.squeeze(2) > 0.05)
            new_ref = [a if a not in ref_idx else b for a, b in classified_edges if (a in ref_idx) != (b in ref_idx)]
            for j in new_ref:
                ref_part[0, j] = True
            reference = x.clone()
            if bool((classified == larger).all()):
                break
            merges = [(a, b) for a, b in classified_edges
                      if a not in ref_idx and b not in ref_idx and a not in new_ref and b not in new_ref]
            if merges and self.merge_fn is not None:
                self.merge_fn(self, merges, nodes, dict(x=x, part_pcs=part_pcs, part_scale=part_scale,
                                                        part_valids=part_valids, classified=classified))
            if bool((classified == larger).all()):
                break
        final = ops.pose_compose(x[0].contiguous(), pivot)
        valid_nodes = data_dict["part_valids"][0, :n_nodes].bool()
        return {
            "pred_trans": final[:, :3], "pred_rots": final[:, 3:], "x": x,
            "trajectory": torch.stack(traj, 0)[:, valid_nodes],          # [T_total, Pv, 7] like predict_*.npy (:322-337)
            "ref_part": ref_part, "verifier_calls": verifier_calls, "steps": step_no,
        }

    def save_inference_data(self, out, path: str):
        np.save(path, out["trajectory"].cpu().numpy())
