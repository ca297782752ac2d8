"""Evaluation metrics (drop-in for denoiser/evaluation/evaluator.py): part accuracy, shape Chamfer distance,
rotation / translation errors.  Same function names, arguments and return values; the point-cloud work
(transforms, nearest neighbours, Euler conversion) runs on the HIP kernels, the [B,P]-sized reductions are
plain tensor arithmetic in the reference's order."""
from __future__ import annotations

import torch

from pfpp_hip import ops
from puzzlefusion_plusplus.denoiser.evaluation.transform import quaternion_to_euler, transform_pc


class ChamferDistance(torch.nn.Module):
    """chamferdist.ChamferDistance (third-party in the reference, `from chamferdist import ChamferDistance`,
    denoiser.py:7): KNN-1 squared distances source->target (and target->source), reduced over points then batch."""

    def forward(self, source_cloud, target_cloud, bidirectional=False, reverse=False, batch_reduction="mean",
                point_reduction="sum"):
        if source_cloud.dim() != 3 or target_cloud.dim() != 3 or source_cloud.shape[0] != target_cloud.shape[0]:
            raise ValueError("ChamferDistance: [B,N,3] and [B,M,3] clouds with the same batch size expected")
        if reverse and bidirectional:
            raise ValueError("ChamferDistance: reverse and bidirectional are exclusive")
        if point_reduction not in ("sum", "mean", None) or batch_reduction not in ("sum", "mean", None):
            raise ValueError("ChamferDistance: reductions must be 'sum', 'mean' or None")
        src = source_cloud.float().contiguous()
        dst = target_cloud.float().contiguous()
        fwd = ops.nn_dist(src, dst) if (not reverse) else None
        bwd = ops.nn_dist(dst, src) if (reverse or bidirectional) else None

        def reduce(c):
            if c is None:
                return None
            if point_reduction == "sum":
                c = c.sum(1)
            elif point_reduction == "mean":
                c = c.mean(1)
            if batch_reduction == "sum":
                c = c.sum()
            elif batch_reduction == "mean":
                c = c.mean()
            return c

        fwd, bwd = reduce(fwd), reduce(bwd)
        if bidirectional:
            return fwd + bwd
        return bwd if reverse else fwd


def _valid_mean(loss_per_part, valids):
    """average over the valid parts, NaNs counted as 0 (evaluator.py:8-22)"""
    nan_mask = torch.isnan(loss_per_part)
    loss_per_part[nan_mask] = 0.
    valids = valids.float().detach()
    return (loss_per_part * valids).sum(1) / valids.sum(1)


def trans_metrics(trans1, trans2, valids, metric):
    """translation error per puzzle (evaluator.py:25-50)"""
    assert metric in ['mse', 'rmse', 'mae']
    if metric == 'mse':
        per_part = (trans1 - trans2).pow(2).mean(dim=-1)
    elif metric == 'rmse':
        per_part = (trans1 - trans2).pow(2).mean(dim=-1) ** 0.5
    else:
        per_part = (trans1 - trans2).abs().mean(dim=-1)
    return _valid_mean(per_part, valids)


@torch.no_grad()
def rot_metrics(rot1, rot2, valids, metric):
    """rotation error in Euler-angle (degree) space per puzzle (evaluator.py:53-85)"""
    assert metric in ['mse', 'rmse', 'mae']
    deg1 = quaternion_to_euler(rot1, to_degree=True)
    deg2 = quaternion_to_euler(rot2, to_degree=True)
    diff1 = (deg1 - deg2).abs()
    diff2 = 360. - (deg1 - deg2).abs()
    diff = torch.minimum(diff1, diff2)
    if metric == 'mse':
        per_part = diff.pow(2).mean(dim=-1)
    elif metric == 'rmse':
        per_part = diff.pow(2).mean(dim=-1) ** 0.5
    else:
        per_part = diff.abs().mean(dim=-1)
    return _valid_mean(per_part, valids)


@torch.no_grad()
def calc_part_acc(pts, trans1, trans2, rot1, rot2, valids, chamfer_distance=None):
    """Part Accuracy: per-part Chamfer distance between the part under the predicted and the GT pose below 0.01
    (evaluator.py:88-121) -> (acc [B], acc_per_part [B,P] bool, cd_per_part [B,P])"""
    chamfer_distance = chamfer_distance or ChamferDistance()
    B, P = pts.shape[:2]
    pts1 = transform_pc(trans1, rot1, pts).flatten(0, 1)
    pts2 = transform_pc(trans2, rot2, pts).flatten(0, 1)
    loss_per_data = chamfer_distance(pts1, pts2, bidirectional=True, point_reduction="mean", batch_reduction=None)
    loss_per_data = loss_per_data.view(B, P).type_as(pts)
    thre = 0.01
    acc_per_part = (loss_per_data < thre) & (valids == 1)
    acc = acc_per_part.sum(-1) / (valids == 1).sum(-1)
    return acc, acc_per_part, loss_per_data


@torch.no_grad()
def calc_shape_cd(pts, trans1, trans2, rot1, rot2, valids, chamfer_distance=None):
    """Chamfer distance between the assembled shapes, padded parts pushed to 1e3 (evaluator.py:124-153) -> [B]"""
    chamfer_distance = chamfer_distance or ChamferDistance()
    B, P, N, _ = pts.shape
    valid_mask = valids[..., None, None]
    pts = pts.detach().clone()
    pts = pts.masked_fill(valid_mask == 0, 1e3)
    shape1 = transform_pc(trans1, rot1, pts).flatten(1, 2)
    shape2 = transform_pc(trans2, rot2, pts).flatten(1, 2)
    shape_cd = chamfer_distance(shape1, shape2, bidirectional=True, point_reduction=None, batch_reduction=None)
    shape_cd = shape_cd.view(B, P, N).mean(-1)
    return _valid_mean(shape_cd, valids)
