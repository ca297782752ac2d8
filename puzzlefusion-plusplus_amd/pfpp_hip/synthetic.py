"""Breaking-Bad-shaped synthetic puzzles (host-side data utility; numpy on the CPU).

Produces the dict GeometryLatentDataset.__getitem__ returns
(puzzlefusion_plusplus/denoiser/dataset/dataset.py:214-221), batched:
  part_pcs   f32 [B,P,N,3]  each valid fragment recentred, rotated into a random canonical pose and
                            divided by its max-abs (values in [-1,1]); padded fragments all zero
  part_valids f32 [B,P]     first Pv entries 1
  part_scale f32 [B,P,1]    max-abs before normalisation (padded = 1)
  part_trans f32 [B,P,3]    centroid of the fragment in the assembled frame
  part_rots  f32 [B,P,4]    unit quaternion (w,x,y,z) taking the canonical fragment back to the assembly
  ref_part   bool [B,P]     one-hot at the largest fragment
  num_parts  i64 [B]
A closed surface (random ellipsoid, optionally with a flattened band) is cut into Pv shards by
Voronoi cells of random surface seeds and N points are drawn per shard.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


def _rand_unit_quat(rng: np.random.Generator, n: int) -> np.ndarray:
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 0] < 0] *= -1
    return q


def _quat_to_mat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=-1)
    return m.reshape(q.shape[:-1] + (3, 3))


def sample_num_parts(rng: np.random.Generator, max_parts: int = 20) -> int:
    """skewed-small fragment count: 2 + min(max-2, Geometric(0.25))"""
    return int(2 + min(max_parts - 2, rng.geometric(0.25)))


def make_puzzle(puzzle_id: int, num_points: int = 1000, max_parts: int = 20, num_parts: Optional[int] = None,
                quantise_bits: Optional[int] = None) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(1234 + puzzle_id)
    pv = num_parts if num_parts is not None else sample_num_parts(rng, max_parts)
    pv = max(2, min(pv, max_parts))
    axes = rng.uniform(0.35, 1.0, size=3)
    n_surf = max(40000, 4 * pv * num_points)
    d = rng.normal(size=(n_surf, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    surf = d * axes
    if rng.random() < 0.5:  # a flattened band makes it less ball-like (cylinder-ish)
        surf[:, 2] = np.clip(surf[:, 2], -0.6 * axes[2], 0.6 * axes[2])
    surf *= 1.0 + 0.03 * rng.normal(size=(n_surf, 1))          # a little thickness
    seeds = surf[rng.choice(n_surf, size=pv, replace=False)]
    owner = np.argmin(((surf[:, None, :] - seeds[None]) ** 2).sum(-1), axis=1)
    g = _quat_to_mat(_rand_unit_quat(rng, 1))[0]
    surf = surf @ g.T                                             # random global orientation
    out = {
        "part_pcs": np.zeros((max_parts, num_points, 3), np.float32),
        "part_valids": np.zeros((max_parts,), np.float32),
        "part_scale": np.ones((max_parts, 1), np.float32),
        "part_trans": np.zeros((max_parts, 3), np.float32),
        "part_rots": np.zeros((max_parts, 4), np.float32),
        "ref_part": np.zeros((max_parts,), bool),
        "num_parts": np.int64(pv),
    }
    out["part_rots"][:, 0] = 1.0
    q_parts = _rand_unit_quat(rng, pv)
    for p in range(pv):
        cand = np.nonzero(owner == p)[0]
        if cand.size == 0:
            cand = np.array([int(np.argmin(((surf - seeds[p] @ g.T) ** 2).sum(-1)))])
        pts = surf[rng.choice(cand, size=num_points, replace=cand.size < num_points)]
        c = pts.mean(0)
        R = _quat_to_mat(q_parts[p])
        canon = (pts - c) @ R                                     # = R^T (pts - c)
        s = np.abs(canon).max()
        canon = canon / s
        if quantise_bits is not None:
            step = 2.0 ** -quantise_bits
            canon = np.round(canon / step) * step
        out["part_pcs"][p] = canon.astype(np.float32)
        out["part_valids"][p] = 1.0
        out["part_scale"][p, 0] = s
        out["part_trans"][p] = c
        out["part_rots"][p] = q_parts[p]
    out["ref_part"][int(np.argmax(out["part_scale"][:pv, 0]))] = True
    ref_c = out["part_trans"][out["ref_part"]].copy()
    out["part_trans"][:pv] -= ref_c                               # assembly frame centred on the reference part
    return out


def num_parts_of(puzzle_id: int, max_parts: int = 20) -> int:
    """valid-fragment count make_puzzle(puzzle_id) will draw — without building the puzzle (load balancing over ranks)"""
    rng = np.random.default_rng(1234 + puzzle_id)
    return max(2, min(sample_num_parts(rng, max_parts), max_parts))


def make_batch(first_id: int, batch: int, num_points: int = 1000, max_parts: int = 20,
               num_parts: Optional[int] = None, quantise_bits: Optional[int] = None,
               device: str | torch.device = "cpu", ids: Optional[list] = None) -> Dict[str, torch.Tensor]:
    """puzzles first_id .. first_id + batch - 1, or the explicit list `ids`"""
    ids = [first_id + i for i in range(batch)] if ids is None else list(ids)
    items = [make_puzzle(i, num_points, max_parts, num_parts, quantise_bits) for i in ids]
    out = {}
    for k in items[0]:
        arr = np.stack([it[k] for it in items], 0)
        out[k] = torch.from_numpy(arr).to(device)
    return out


def make_edges(batch: int, num_edges: int = 190, num_nodes: int = 20, seed: int = 0,
               device: str | torch.device = "cpu") -> Dict[str, torch.Tensor]:
    """verifier inputs shaped like VerifierDataset (verifier/dataset/dataset.py:88-100): a 6-bin
    normalised histogram + match count per edge, upper-triangular edge list, validity mask."""
    rng = np.random.default_rng(99 + seed)
    iu = np.stack(np.triu_indices(num_nodes, k=1), -1)[:num_edges]
    feats = np.zeros((batch, num_edges, 7), np.float32)
    valid = np.zeros((batch, num_edges), np.float32)
    for b in range(batch):
        pv = sample_num_parts(rng, num_nodes)
        ok = (iu[:, 0] < pv) & (iu[:, 1] < pv)
        cnt = rng.integers(30, 300, size=num_edges).astype(np.float32) * (rng.random(num_edges) < 0.4)
        hist = rng.dirichlet(np.ones(6), size=num_edges).astype(np.float32)
        feats[b, :, :6] = hist * (cnt[:, None] > 0)
        feats[b, :, 6] = cnt
        feats[b, ~ok] = 0
        valid[b, ok] = 1
    idx = np.broadcast_to(iu[None], (batch, num_edges, 2)).astype(np.int64).copy()
    return {
        "edge_features": torch.from_numpy(feats).to(device),
        "edge_indices": torch.from_numpy(idx).to(device),
        "edge_valids": torch.from_numpy(valid).to(device),
    }


def make_matching(batch: Dict[str, torch.Tensor], seed: int = 0, total_points: int = 5000) -> Dict[str, object]:
    """synthetic stand-in for the Jigsaw matching data of one puzzle (B = 1), in the on-disk layout the
    reference reads (Jigsaw_matching/model/modules/matching_base_model.py:630-640): `edges [1,E,2]` =
    (idx2, idx1) with idx1 < idx2, one `[M,2]` correspondence array per edge (indices into the critical
    points of the two parts), `part_pcs_by_area [1,5000,3]` (points per part proportional to its size, in
    the part's scaled local frame), `critical_pcs_idx [1,5000]`, `n_pcs [1,P]`, `n_critical_pcs [1,P]`."""
    rng = np.random.default_rng(777 + seed)
    pv = int(batch["num_parts"][0])
    P = batch["part_valids"].shape[1]
    scale = batch["part_scale"][0, :pv, 0].cpu().numpy().astype(np.float64)
    n_pcs = np.maximum(50, np.floor(total_points * scale / scale.sum())).astype(np.int64)
    n_pcs[0] += total_points - n_pcs.sum() if n_pcs.sum() <= total_points else 0
    pts, crit, n_crit = [], [], []
    N = batch["part_pcs"].shape[2]
    for i in range(pv):
        sel = rng.choice(N, size=int(n_pcs[i]), replace=int(n_pcs[i]) > N)
        pts.append(batch["part_pcs"][0, i].cpu().numpy()[sel] * float(scale[i]))
        nc = max(10, int(n_pcs[i]) // 4)
        n_crit.append(nc)
        c = np.zeros(int(n_pcs[i]), np.int64)
        c[:nc] = rng.choice(int(n_pcs[i]), size=nc, replace=False)
        crit.append(c)
    tot = int(n_pcs.sum())
    by_area = np.zeros((1, max(tot, total_points), 3), np.float32); by_area[0, :tot] = np.concatenate(pts)
    crit_all = np.zeros((1, max(tot, total_points)), np.int64); crit_all[0, :tot] = np.concatenate(crit)
    n_pcs_p = np.zeros((1, P), np.int64); n_pcs_p[0, :pv] = n_pcs
    n_crit_p = np.zeros((1, P), np.int64); n_crit_p[0, :pv] = n_crit
    edges, corr = [], []
    for i in range(pv):
        for j in range(i + 1, pv):
            if rng.random() < min(1.0, 3.0 / pv):
                m = int(rng.integers(30, 300))
                edges.append((j, i))                     # (idx2, idx1)
                corr.append(np.stack([rng.integers(0, n_crit[i], m), rng.integers(0, n_crit[j], m)], 1))
    if not edges:
        edges, corr = [(1, 0)], [np.stack([rng.integers(0, n_crit[0], 40), rng.integers(0, n_crit[1], 40)], 1)]
    dev = batch["part_pcs"].device
    return {
        "edges": torch.tensor(edges, dtype=torch.int64)[None].to(dev), "correspondences": corr,
        "part_pcs_by_area": torch.from_numpy(by_area).to(dev), "critical_pcs_idx": torch.from_numpy(crit_all).to(dev),
        "n_pcs": torch.from_numpy(n_pcs_p).to(dev), "n_critical_pcs": torch.from_numpy(n_crit_p).to(dev),
    }
