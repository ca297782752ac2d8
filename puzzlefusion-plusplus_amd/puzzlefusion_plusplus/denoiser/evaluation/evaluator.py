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


_REDUCERS = {
    # metric -> how a [B, P, D] tensor of non-negative per-axis deviations collapses to one number per part
    "mse": lambda dev: dev.pow(2).mean(dim=-1),
    "rmse": lambda dev: dev.pow(2).mean(dim=-1) ** 0.5,
    "mae": lambda dev: dev.abs().mean(dim=-1),
}
PART_ACC_THRESHOLD = 0.01      # a part counts as placed when its Chamfer distance to the ground-truth placement is below this
PADDED_PART_COORD = 1e3        # padded parts are parked far away before whole-shape distances are taken


def _puzzle_average(per_part, valids):
    """[B, P] per-part values -> [B]: mean over the valid parts of each puzzle; a NaN part contributes 0 (evaluator.py:8-22)"""
    weight = valids.detach().float()
    clean = torch.where(torch.isnan(per_part), torch.zeros_like(per_part), per_part)
    return (clean * weight).sum(1) / weight.sum(1)


def _metric(deviation, valids, metric):
    if metric not in _REDUCERS:
        raise AssertionError(f"metric must be one of {sorted(_REDUCERS)}, got {metric!r}")
    return _puzzle_average(_REDUCERS[metric](deviation), valids)


def trans_metrics(trans1, trans2, valids, metric):
    """translation error per puzzle (evaluator.py:25-50)"""
    return _metric(trans1 - trans2, valids, metric)


@torch.no_grad()
def rot_metrics(rot1, rot2, valids, metric):
    """rotation error per puzzle in Euler-angle degrees, each axis taken the short way round the circle (evaluator.py:53-85)"""
    gap = (quaternion_to_euler(rot1, to_degree=True) - quaternion_to_euler(rot2, to_degree=True)).abs()
    return _metric(torch.minimum(gap, 360. - gap), valids, metric)


@torch.no_grad()
def calc_part_acc(pts, trans1, trans2, rot1, rot2, valids, chamfer_distance=None):
    """Part Accuracy (evaluator.py:88-121): bidirectional mean Chamfer distance of every part between its two placements, a part is
    accurate below PART_ACC_THRESHOLD -> (accuracy [B], accurate [B,P] bool, distance [B,P])"""
    cd = chamfer_distance or ChamferDistance()
    n_puzzles, n_parts = pts.shape[:2]
    placed = [transform_pc(t, r, pts).flatten(0, 1) for t, r in ((trans1, rot1), (trans2, rot2))]
    dist = cd(placed[0], placed[1], bidirectional=True, point_reduction="mean", batch_reduction=None).view(n_puzzles, n_parts).type_as(pts)
    real = valids == 1
    accurate = (dist < PART_ACC_THRESHOLD) & real
    return accurate.sum(-1) / real.sum(-1), accurate, dist


@torch.no_grad()
def calc_shape_cd(pts, trans1, trans2, rot1, rot2, valids, chamfer_distance=None):
    """Chamfer distance between the two assembled shapes (evaluator.py:124-153): per-point bidirectional distances averaged per part,
    then over the valid parts -> [B]"""
    cd = chamfer_distance or ChamferDistance()
    n_puzzles, n_parts, n_pts, _ = pts.shape
    parked = pts.detach().clone().masked_fill(valids[..., None, None] == 0, PADDED_PART_COORD)
    whole = [transform_pc(t, r, parked).flatten(1, 2) for t, r in ((trans1, rot1), (trans2, rot2))]
    per_point = cd(whole[0], whole[1], bidirectional=True, point_reduction=None, batch_reduction=None)
    return _puzzle_average(per_point.view(n_puzzles, n_parts, n_pts).mean(-1), valids)
