"""Drop-in for the reference's utils/pn2_utils.py, backed by the HIP kernels.

Same free functions / module names and tensor conventions as the reference (channel-first
[B, C, N] in and out of PointNetSetAbstraction, int64 indices, `[rel_xyz | feats]` column order of
sample_and_group), but FPS, ball query, grouping and the 1x1-conv/BN/ReLU/max chain run in
libpfpp_hip.so.  CPU tensors are rejected — there is no CPU path.
Reference line numbers refer to utils/pn2_utils.py.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from pfpp_hip import ops
from pfpp_hip.packing import PW, PackCache, fold_conv_bn, pack_sa_first


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points [B,N,C], idx [B,S] or [B,S,K] -> gathered rows (:45-62); pure indexing (plumbing)"""
    B = points.shape[0]
    flat = idx.reshape(B, -1).long()
    out = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """[B,N,C] x [B,M,C] -> [B,N,M] squared distances in the expanded form of :21-42,
    through the fp32 GEMM kernel (batched, A.W^T)."""
    B, N, C = src.shape
    M = dst.shape[1]
    cp = (C + 3) // 4 * 4
    a = torch.zeros((B, N, cp), dtype=torch.float32, device=src.device); a[..., :C] = src
    w = torch.zeros((B, M, cp), dtype=torch.float32, device=src.device); w[..., :C] = dst
    out = torch.empty((B, N, M), dtype=torch.float32, device=src.device)
    ops.gemm(a, w, M=N, N=M, K=C, lda=cp, ldw=cp, out=out, ldc=M, batch=B, sA=(N * cp, 0), sW=(M * cp, 0),
             sC=(N * M, 0), alpha=-2.0)
    out += (src ** 2).sum(-1)[:, :, None]
    out += (dst ** 2).sum(-1)[:, None, :]
    return out


def farthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """[B,N,3] -> int64 [B,npoint].  The reference's version (:65-89) starts at a RANDOM point and
    is dead code on the encoder path (PN2 uses torch_cluster.fps with random_start=False, :131-137);
    this one is the deterministic start-0 sampler the encoder uses."""
    idx, _ = ops.fps(xyz.contiguous(), npoint)
    return idx.long()


def query_ball_point(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """first `nsample` indices (ascending) with d <= r^2, padded with the first (:92-112); int64"""
    return ops.ball_query(xyz.contiguous(), new_xyz.contiguous(), radius, nsample).long()


def sample_and_group(npoint: int, radius: float, nsample: int, xyz: torch.Tensor,
                     points: Optional[torch.Tensor], returnfps: bool = False):
    """xyz [B,N,3], points [B,N,D] -> new_xyz [B,S,3], new_points [B,S,ns,3+D] (rel_xyz first) (:115-152)"""
    xyz = xyz.contiguous()
    B, N, _ = xyz.shape
    ops.check_fps_ratio(npoint, N)            # the reference samples ceil(float64(npoint / N) * N) points (:131-134)
    fps_idx, new_xyz = ops.fps(xyz, npoint)
    ball = ops.ball_query(xyz, new_xyz, radius, nsample)
    feats = points.contiguous() if points is not None else None
    D = 0 if feats is None else feats.shape[-1]
    g = ops.group_gather(xyz, new_xyz, feats, ball).view(B, npoint, nsample, D + 4)
    new_points = torch.cat([g[..., D:D + 3], g[..., :D]], dim=-1) if D else g[..., :3].contiguous()
    if returnfps:
        grouped_xyz = index_points(xyz, ball.long())
        return new_xyz, new_points, grouped_xyz, fps_idx.long().reshape(-1)
    return new_xyz, new_points


class PointNetSetAbstraction(nn.Module):
    """PointNet++ set-abstraction level (:175-216).  Parameters are the reference's
    (mlp_convs.{i}: Conv2d 1x1, mlp_bns.{i}: BatchNorm2d); forward runs
    FPS -> ball query -> grouping -> 3 x GEMM(+BN scale/shift, ReLU) -> max over nsample on the GPU.
    .eval(): BatchNorm folded into the GEMM epilogue from its running statistics.  .train(): batch statistics with the running
    buffers updated in place (what the reference's frozen-but-train-mode encoder does, train_denoiser.py:33-35) through the
    same fused-BatchNorm GEMM chain as pfpp_hip.encoder; forward only — parameters stay frozen on this path."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all=False):
        super().__init__()
        if group_all:
            raise NotImplementedError("group_all=True is not used by PN2 (vqvae/model/modules/pn2.py:16-18)")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_c in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_c, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_c))
            last = out_c
        self.group_all = group_all
        self._cache = PackCache()

    def _packed(self):
        srcs = [t for c, b in zip(self.mlp_convs, self.mlp_bns)
                for t in (c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var)]

        def build():
            pk = {}
            for i, (c, b) in enumerate(zip(self.mlp_convs, self.mlp_bns)):
                w, s, t = fold_conv_bn(c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var, b.eps)
                if i == 0:
                    w = pack_sa_first(w, w.shape[1] - 3)
                pk[f"w{i}"], pk[f"s{i}"], pk[f"t{i}"] = PW(w.contiguous(), prescale=False), s, t
            return pk

        return self._cache.get(srcs, build)

    def _train_pack(self):
        """operands of the batch-statistics chain (pfpp_hip.encoder._sa_mlp_train): raw 1x1-conv weights as split planes and the
        module's own BatchNorm tensors — the running buffers are updated in place through these references"""
        srcs = [t for c in self.mlp_convs for t in (c.weight, c.bias)]

        def build():
            pk = {}
            for i, c in enumerate(self.mlp_convs):
                w = c.weight.detach().reshape(c.weight.shape[0], -1)
                if i == 0:
                    w = pack_sa_first(w, w.shape[1] - 3)
                pk[f"sa.w{i}"] = PW(w.contiguous(), prescale=False)
                pk[f"sa.b{i}"] = c.bias.detach().contiguous()
            return pk

        if not hasattr(self, "_train_cache"):
            self._train_cache = PackCache()
        pk = dict(self._train_cache.get(srcs, build))
        for i, b in enumerate(self.mlp_bns):
            pk[f"sa.g{i}"], pk[f"sa.be{i}"] = b.weight.detach(), b.bias.detach()
            pk[f"sa.rm{i}"], pk[f"sa.rv{i}"], pk[f"sa.nbt{i}"] = b.running_mean, b.running_var, b.num_batches_tracked
        if not hasattr(self, "_train_stats"):
            self._train_stats = {}
        pk.update(self._train_stats)
        return pk

    def forward_channels_last(self, xyz: torch.Tensor, feats: Optional[torch.Tensor]):
        """xyz [B,N,3], feats [B,N,D] -> new_xyz [B,S,3], new_feats [B,S,C] (the layout the kernels use)"""
        B = xyz.shape[0]
        ops.check_fps_ratio(self.npoint, xyz.shape[1])
        _, new_xyz = ops.fps(xyz, self.npoint)
        ball = ops.ball_query(xyz, new_xyz, self.radius, self.nsample)
        h = ops.group_gather(xyz, new_xyz, feats, ball)
        n = len(self.mlp_convs)
        if self.training:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise RuntimeError("PointNetSetAbstraction (HIP): the train-mode path is forward only (frozen encoder, "
                                   "train_denoiser.py:33-35): set requires_grad=False on its parameters or run under no_grad")
            if n != 3 or not all(b.track_running_stats for b in self.mlp_bns):
                raise RuntimeError("PointNetSetAbstraction (HIP): train mode needs three conv/BatchNorm pairs with running statistics")
            from pfpp_hip.encoder import _sa_mlp_train

            pk = self._train_pack()
            out = _sa_mlp_train(pk, "sa", h, self.nsample)
            self._train_stats.update({k: v for k, v in pk.items() if ".stats" in k})      # statistics accumulators: allocated once
            return new_xyz, out.view(B, self.npoint, -1)
        pk = self._packed()
        for i in range(n):
            h = ops.linear(h, pk[f"w{i}"], scale=pk[f"s{i}"], shift=pk[f"t{i}"], act="relu",
                           pool=self.nsample if i == n - 1 else 0)
        return new_xyz, h.view(B, self.npoint, -1)

    def forward(self, xyz: torch.Tensor, points: Optional[torch.Tensor]):
        """xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,C,S] (:190-216)"""
        x = xyz.permute(0, 2, 1).contiguous()
        f = points.permute(0, 2, 1).contiguous() if points is not None else None
        new_xyz, new_feats = self.forward_channels_last(x, f)
        return new_xyz.permute(0, 2, 1), new_feats.permute(0, 2, 1)
