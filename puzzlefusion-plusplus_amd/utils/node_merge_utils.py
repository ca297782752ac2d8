"""Node-merge helpers of the auto-agglomerative loop (drop-in for utils/node_merge_utils.py), HIP-backed.

Same function names and argument meaning as the reference for the pieces AutoAgglomerative.test_step calls; the
point-cloud work (pose application, nearest neighbours, normal estimation, intersect filter, farthest point
sampling) runs on the kernels of libpfpp_hip.so, the bookkeeping on a handful of graph nodes stays on the host.
`nodes` is any mapping index -> dict(pivot, valids, ref_part, init_pose) (the reference passes networkx's G.nodes)."""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from pfpp_hip import ops


def get_final_pose_pts(pts: torch.Tensor, trans: torch.Tensor, rots: torch.Tensor) -> torch.Tensor:
    """pts [B,P,N,3], trans [B,P,3], rots [B,P,4] -> R(q/|q|) p + t  (node_merge_utils.py:43-53)"""
    B, P, N, _ = pts.shape
    pose = torch.cat([trans, rots], dim=-1).reshape(B * P, 7).float().contiguous()
    return ops.pose_apply(pts.reshape(B * P, N, 3).float().contiguous(), pose, normalise=True).reshape(B, P, N, 3)


def get_pc_start_end(idx: int, n_pcs: torch.Tensor):
    """[start, end) of part idx inside the flat by-area cloud (node_merge_utils.py:55-59)"""
    c = n_pcs.cumsum(dim=1)
    return (0 if idx == 0 else int(c[0, idx - 1])), int(c[0, idx])


def node_merge_valids_check(edge, ref_part: torch.Tensor, nodes) -> bool:
    """an edge may merge its two nodes unless a reference part is involved (node_merge_utils.py:90-105)"""
    a, b = int(edge[0]), int(edge[1])
    ref_idx = set(torch.where(ref_part)[1].tolist())
    if a in ref_idx or b in ref_idx:
        return False
    return not (bool(ref_part[0][nodes[a]["pivot"]]) or bool(ref_part[0][nodes[b]["pivot"]]))


def merge_node(components: Iterable[int], nodes, pcs: torch.Tensor) -> torch.Tensor:
    """concatenate the (posed) clouds of the still-valid nodes of a component (node_merge_utils.py:125-136)"""
    keep = [i for i in components if nodes[i]["valids"]]
    return pcs[torch.tensor(keep, device=pcs.device)].reshape(-1, 3)


def remove_intersect_points_and_fps_ds(merge_pcs: torch.Tensor, cd_loss=None, num_points: int = 1000,
                                       threshold: float = 0.001, start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """drop the points of each part that lie on a surface shared with another part (close in the bidirectional
    nearest-neighbour sense AND with opposing normals), then farthest-point-sample the union back to `num_points`
    (node_merge_utils.py:159-222).  `start`: first FPS index (the reference draws it at random, torch_cluster
    random_start); cd_loss is accepted for signature compatibility — the distances come from pfpp_nn_dist."""
    parts = merge_pcs.reshape(-1, num_points, 3).float().contiguous()
    P, N, _ = parts.shape
    normals = ops.estimate_normals(parts, 20)
    src = parts.unsqueeze(1).expand(P, P, N, 3).reshape(P * P, N, 3).contiguous()
    dst = parts.unsqueeze(0).expand(P, P, N, 3).reshape(P * P, N, 3).contiguous()
    d = ops.nn_dist(src, dst).view(P, P, N)                     # d[i, j, k]: point k of part i -> part j
    keep = ops.merge_keep_mask(d.contiguous(), normals, threshold)
    final = parts[keep].contiguous()                            # part by part, original order
    n = final.shape[0]
    if n < num_points:
        raise RuntimeError(f"merge: only {n} points survive the intersect filter (< {num_points})")
    if start is None:
        start = torch.randint(0, n, (1,), device=parts.device)
    _, new_xyz = ops.fps(final[None], num_points, start=start.reshape(1).to(torch.int32))
    return new_xyz[0]


def pose_to_affine(trans: torch.Tensor, rots: torch.Tensor) -> torch.Tensor:
    """4x4 [R(q) | t] with pytorch3d's quaternion_to_matrix (node_merge_utils.py:233-238)"""
    r, i, j, k = rots.float().unbind(-1)
    two_s = 2.0 / (rots.float() * rots.float()).sum(-1)
    m = torch.eye(4, device=trans.device, dtype=torch.float32)
    m[0, 0] = 1 - two_s * (j * j + k * k); m[0, 1] = two_s * (i * j - k * r); m[0, 2] = two_s * (i * k + j * r)
    m[1, 0] = two_s * (i * j + k * r); m[1, 1] = 1 - two_s * (i * i + k * k); m[1, 2] = two_s * (j * k - i * r)
    m[2, 0] = two_s * (i * k - j * r); m[2, 1] = two_s * (j * k + i * r); m[2, 2] = 1 - two_s * (i * i + j * j)
    m[:3, 3] = trans.float()
    return m


def assign_init_pose(nodes, trans: torch.Tensor, rots: torch.Tensor, centroid: torch.Tensor, component) -> None:
    """node.init_pose <- [R(q_pivot) | t_pivot - centroid] @ node.init_pose (node_merge_utils.py:225-244)"""
    for idx in component:
        node = nodes[idx]
        piv = node["pivot"]
        a = pose_to_affine(trans[piv] - centroid, rots[piv])
        node["init_pose"] = a if node["init_pose"] is None else a @ node["init_pose"]
