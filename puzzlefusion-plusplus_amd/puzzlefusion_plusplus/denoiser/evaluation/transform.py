"""Point-cloud transforms of the evaluation metrics (drop-in for denoiser/evaluation/transform.py), HIP-backed.

qrot / qtransform / transform_pc: pytorch3d quaternion_apply (no normalisation) + translation, the same kernel as
the pose helpers of a19 (bit-identical operation order).  quaternion_to_euler: quaternion_to_matrix +
matrix_to_euler_angles("XYZ") in one kernel (transform.py:70-86)."""
from __future__ import annotations

import torch

from pfpp_hip import ops


def _broadcast_to_points(a: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """[..., k] -> [..., N, k] like the reference's unsqueeze(-2).repeat_interleave(N, dim=-2) (transform.py:17-18,38-39)"""
    if a.dim() == v.dim() - 1:
        a = a.unsqueeze(-2).expand(*v.shape[:-1], a.shape[-1])
    if a.shape[:-1] != v.shape[:-1]:
        raise AssertionError("quaternion / translation shape does not match the points")
    return a


def qtransform(t: torch.Tensor, q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """rotate v [..., 3] by q [..., 4] (or [..., N, 4]) then translate by t (transform.py:26-47)"""
    if t.shape[-1] != 3:
        raise AssertionError("translation must be [..., 3]")
    if q.dim() == v.dim() - 1 and t.dim() == v.dim() - 1:
        # one pose per cloud: the fused per-fragment kernel
        lead = v.shape[:-2]
        pose = torch.cat([t.reshape(-1, 3), q.reshape(-1, 4)], dim=-1).float().contiguous()
        out = ops.pose_apply(v.reshape(-1, v.shape[-2], 3).float().contiguous(), pose, normalise=False)
        return out.reshape(*lead, v.shape[-2], 3)
    qq = _broadcast_to_points(q, v).reshape(-1, 4).float().contiguous()
    tt = _broadcast_to_points(t, v).reshape(-1, 3).float().contiguous()
    pose = torch.cat([tt, qq], dim=-1).contiguous()
    out = ops.pose_apply(v.reshape(-1, 1, 3).float().contiguous(), pose, normalise=False)
    return out.reshape(v.shape)


def qrot(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """rotate v by q (transform.py:7-23)"""
    return qtransform(torch.zeros(q.shape[:-1] + (3,), dtype=v.dtype, device=v.device), q, v)


def transform_pc(trans: torch.Tensor, rot: torch.Tensor, pc: torch.Tensor, rot_type=None) -> torch.Tensor:
    """rotate and translate the point cloud(s) (transform.py:50-57)"""
    return qtransform(trans, rot, pc)


def quaternion_to_euler(quat: torch.Tensor, to_degree: bool = True) -> torch.Tensor:
    """[..., 4] -> [..., 3] Euler angles "XYZ" (transform.py:60-76)"""
    return ops.quat_to_euler_xyz(quat.float().contiguous(), to_degree)
