"""Batch augmentation on the GPU: the arithmetic of GeometryLatentDataset.__getitem__
(denoiser/dataset/dataset.py:165-215) for a whole batch in one kernel, fed by the dataset in `device_augment` mode."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check
from .ops import _chk, _ptr, _stream


def fragment_prepare(part_pcs_gt: torch.Tensor, num_parts: torch.Tensor, ref_idx: torch.Tensor, q_global: torch.Tensor,
                     q_part: torch.Tensor):
    """raw pfpp_fragment_prepare -> (part_pcs [B,P,N,3], part_trans [B,P,3], part_scale [B,P,1], init_pose_t [B,3])"""
    _chk(part_pcs_gt, torch.float32, "part_pcs_gt"); _chk(num_parts, torch.int32, "num_parts"); _chk(ref_idx, torch.int32, "ref_idx")
    _chk(q_global, torch.float32, "q_global"); _chk(q_part, torch.float32, "q_part")
    B, P, N, _ = part_pcs_gt.shape
    if q_global.shape != (B, 4) or q_part.shape != (B, P, 4) or num_parts.numel() != B or ref_idx.numel() != B:
        raise ValueError("fragment_prepare: q_global [B,4], q_part [B,P,4], num_parts [B], ref_idx [B] expected")
    dev = part_pcs_gt.device
    pcs = torch.empty_like(part_pcs_gt)
    trans = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((B, P, 1), dtype=torch.float32, device=dev)
    init_t = torch.empty((B, 3), dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws = torch.empty((int(lib.pfpp_fragment_prepare_workspace(B, P)),), dtype=torch.uint8, device=dev)
    check(lib.pfpp_fragment_prepare(_ptr(part_pcs_gt), _ptr(num_parts), _ptr(ref_idx), _ptr(q_global), _ptr(q_part), _ptr(pcs),
                                    _ptr(trans), _ptr(scale), _ptr(init_t), B, P, N, _ptr(ws), _stream()), "pfpp_fragment_prepare")
    return pcs, trans, scale, init_t


def random_unit_quaternions(shape, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """uniform on SO(3) (normalised Gaussian 4-vectors), like scipy's Rotation.random()"""
    q = torch.randn(*shape, 4, device=device, generator=generator)
    return q / q.norm(dim=-1, keepdim=True)


def augment_batch(batch: Dict[str, torch.Tensor], device, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """collated `device_augment` samples (part_pcs_gt [B,P,N,3], part_valids, ref_part, num_parts, ...) -> the dict the
    Denoiser consumes, all on `device`"""
    gt = batch["part_pcs_gt"].to(device).float().contiguous()
    B, P = gt.shape[:2]
    num_parts = batch["num_parts"].to(device).to(torch.int32).contiguous()
    ref_part = batch["ref_part"].to(device).bool()
    ref_idx = ref_part.float().argmax(dim=1).to(torch.int32).contiguous()
    q_g = random_unit_quaternions((B,), device, generator).contiguous()
    q_p = random_unit_quaternions((B, P), device, generator).contiguous()
    pcs, trans, scale, init_t = fragment_prepare(gt, num_parts, ref_idx, q_g, q_p)
    valid = (torch.arange(P, device=device)[None, :] < num_parts[:, None])
    out = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out.update(part_pcs=pcs, part_trans=trans, part_scale=scale, part_rots=q_p * valid[..., None], init_pose_r=q_g,
               init_pose_t=init_t, part_valids=batch["part_valids"].to(device).float(), ref_part=ref_part)
    return out
