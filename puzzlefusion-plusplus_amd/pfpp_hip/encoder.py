"""Fragment encoder on the HIP kernels: rotate -> 3 x [FPS, ball query, group, SA-MLP, max]
-> conv6 -> VQ -> scatter  (SURVEY.md §8a rows a1-a8).

Host orchestration only; all arithmetic is in libpfpp_hip.so.  Activations are kept
channels-last ([rows, C]) so every 1x1 convolution is one GEMM with the BatchNorm /
ReLU / max-over-nsample epilogue fused in.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .packing import PW, fold_conv_bn, pack_sa_first

import os as _os

BN_FUSED = _os.environ.get("PFPP_BN_FUSED", "1") == "1"
# the grouped neighbourhoods [F*S*ns, D+4] are never written out: the first convolution's GEMM gathers its A rows from
# the level's feature table (pfpp_gemm_args.gather_*); 0 = materialise them with pfpp_group_gather as before
GATHER_FUSED = _os.environ.get("PFPP_GATHER_FUSED", "1") == "1"
# eval mode, first level (no input features): grouping + the three folded conv/BN/ReLU + the max in one kernel
SA_FUSED = _os.environ.get("PFPP_SA_FUSED", "1") == "1"

# train mode: levels 1 and 2 as recomputing chain launches (csrc/sa_train.hip) instead of layer-wise GEMMs over [rows, C] activations
SA_TRAIN_CHAIN = _os.environ.get("PFPP_SA_TRAIN_CHAIN", "1") == "1"

# first layer of the levels with features by linearity: conv1 per POINT once (ops.sa_first_table), its value on a grouped row is
# U[point] - W_xyz . centroid — the grouped first convolution (42 / 33 GFLOP at levels 2 / 3) is never computed in train mode
SA_TRAIN_UTAB = _os.environ.get("PFPP_SA_TRAIN_UTAB", "1") == "1"
# the same in eval mode: 0 off, 1 level 2 (ops.sa_mlp2_table), 2 levels 2 and 3 (ops.sa_table_planes feeds level 3's plane GEMMs)
SA_EVAL_UTAB = int(_os.environ.get("PFPP_SA_EVAL_UTAB", "2"))
SA_TRAIN_WIDE = _os.environ.get("PFPP_SA_TRAIN_WIDE", "1") == "1"     # level 3 in train mode as rows launches (sa_wide_train_kernel)

# eval-mode level 3 on the rows kernels of the train-mode chain (0 = elementwise pass + two tiled plane GEMMs: the cross-check), from
# this many grouped rows up (a handful of fragments stays on the tiled path: a persistent rows workgroup loads a 140 KB weight slice first)
SA_EVAL_ROWS = _os.environ.get("PFPP_SA_EVAL_ROWS", "1") == "1"
SA_EVAL_ROWS_MIN = int(_os.environ.get("PFPP_SA_EVAL_ROWS_MIN", "0"))
# the same for level 2 (128 + 3 -> 128 -> 128 -> 256: stage 2 writes the raw second-layer rows, stage 3 keeps the third layer's weights in LDS)
SA_EVAL_ROWS2 = _os.environ.get("PFPP_SA_EVAL_ROWS2", "1") == "1"
SA_EVAL_ROWS2_MIN = int(_os.environ.get("PFPP_SA_EVAL_ROWS2_MIN", "200000"))      # one puzzle in flight (65 K rows): neutral, stays tiled
SAMPLE_FUSED = _os.environ.get("PFPP_SAMPLE_FUSED", "1") != "0"     # FPS + ball query of the three levels in one kernel
SAMPLE_FUSED_MIN = int(_os.environ.get("PFPP_SAMPLE_FUSED_MIN", "32"))    # ... from this many fragments up

# 64-neighbour levels: neighbourhoods the ball query padded beyond their first 32 slots are taken as one half (ops.sa_pad_schedule)
SA_PAD_SKIP = _os.environ.get("PFPP_SA_PAD_SKIP", "1") != "0"
SA_PAD_SKIP_MIN = int(_os.environ.get("PFPP_SA_PAD_SKIP_MIN", "2048"))     # ... from this many neighbourhoods up (one puzzle in flight: the schedule's two launches cost more than they save)

# (name, npoint, radius, nsample) — vqvae/model/modules/pn2.py:16-18
SA_LEVELS = (("sa1", 256, 0.2, 32), ("sa2", 128, 0.4, 64), ("sa3", None, 0.8, 64))


def pack_encoder(sd: Dict[str, torch.Tensor], prefix: str = "") -> Dict[str, torch.Tensor]:
    """sd: live tensors of a VQVAE module keyed by state_dict names -> packed kernel weights"""
    out: Dict[str, torch.Tensor] = {}
    for name, _, _, _ in SA_LEVELS:
        for i in range(3):
            p = f"{prefix}pn2.{name}"
            w, s, t = fold_conv_bn(
                sd[f"{p}.mlp_convs.{i}.weight"], sd[f"{p}.mlp_convs.{i}.bias"],
                sd[f"{p}.mlp_bns.{i}.weight"], sd[f"{p}.mlp_bns.{i}.bias"],
                sd[f"{p}.mlp_bns.{i}.running_mean"], sd[f"{p}.mlp_bns.{i}.running_var"],
            )
            if i == 0:
                w = pack_sa_first(w, w.shape[1] - 3)
            out[f"{name}.w{i}"] = PW(w.contiguous(), prescale=False)     # the fused set-abstraction kernels read these planes directly
            out[f"{name}.s{i}"] = s
            out[f"{name}.t{i}"] = t
    w6 = sd[f"{prefix}pn2.conv6.weight"]
    out["conv6.w"] = PW(w6.reshape(w6.shape[0], -1).contiguous())
    out["conv6.b"] = sd[f"{prefix}pn2.conv6.bias"].contiguous()
    out["codebook"] = sd[f"{prefix}vector_quantization.embedding.weight"].contiguous()
    return out


def pack_encoder_train(sd: Dict[str, torch.Tensor], prefix: str = "") -> Dict[str, torch.Tensor]:
    """train-mode packing: raw conv weights/biases (BatchNorm is NOT folded: it runs on batch statistics), the
    BatchNorm affine parameters and the LIVE running-statistics buffers (updated in place by pfpp_bn_stats)"""
    out: Dict[str, torch.Tensor] = {}
    for name, _, _, _ in SA_LEVELS:
        for i in range(3):
            p = f"{prefix}pn2.{name}"
            w = sd[f"{p}.mlp_convs.{i}.weight"]
            w = w.reshape(w.shape[0], -1)
            if i == 0:
                w = pack_sa_first(w, w.shape[1] - 3)
            out[f"{name}.w{i}"] = PW(w.contiguous(), prescale=False)     # the fused set-abstraction kernels read these planes directly
            out[f"{name}.b{i}"] = sd[f"{p}.mlp_convs.{i}.bias"].contiguous()
            out[f"{name}.g{i}"] = sd[f"{p}.mlp_bns.{i}.weight"].contiguous()
            out[f"{name}.be{i}"] = sd[f"{p}.mlp_bns.{i}.bias"].contiguous()
            out[f"{name}.rm{i}"] = sd[f"{p}.mlp_bns.{i}.running_mean"]
            out[f"{name}.rv{i}"] = sd[f"{p}.mlp_bns.{i}.running_var"]
            out[f"{name}.nbt{i}"] = sd[f"{p}.mlp_bns.{i}.num_batches_tracked"]
    w6 = sd[f"{prefix}pn2.conv6.weight"]
    out["conv6.w"] = PW(w6.reshape(w6.shape[0], -1).contiguous())
    out["conv6.b"] = sd[f"{prefix}pn2.conv6.bias"].contiguous()
    out["codebook"] = sd[f"{prefix}vector_quantization.embedding.weight"].contiguous()
    out["train"] = True
    return out


def _sa_chain_train(pk, name: str, grp, nsample: int) -> torch.Tensor:
    """train-mode level by recomputation (csrc/sa_train.hip): one persistent chain launch per layer; stage k recomputes layers
    1..k-1 in registers and produces layer k's batch statistics — the [rows, C] activations of the layer-wise form are never
    written (level 1) / only the raw second-layer rows are (level 2: its third convolution's weights do not fit in LDS next to the
    others, so stage 3 reads those rows back with that layer's weights resident)"""
    from . import train_ops as T

    xyz, new_xyz, feats, ball = grp
    dev = xyz.device
    F, S = ball.shape[:2]
    rows = F * S * nsample
    ws = [pk[f"{name}.w{i}"] for i in range(3)]
    bs = [pk[f"{name}.b{i}"] for i in range(3)]
    n_chain = 3
    affs = []
    y2 = mx = mn = None
    wide = feats is not None and feats.shape[2] == 256       # level 3: one rows launch per layer (no two weight matrices fit in LDS)
    y_prev = None
    utab = ops.sa_first_table(xyz, feats, ws[0], bs[0]) if (SA_TRAIN_UTAB and feats is not None) else None
    # the padding schedule goes to all stages of the level or to none: with the per-point table every stage takes it
    sched = ops.sa_pad_schedule(ball) if (SA_PAD_SKIP and utab is not None and nsample == 64 and F * S >= SA_PAD_SKIP_MIN) else None
    for i in range(n_chain):
        Cout = ws[i].N
        st = pk.get(f"{name}.stats{i}")
        if st is None:
            st = pk[f"{name}.stats{i}"] = T.bn_stats_buffer(Cout, dev)
        if i == 2:
            mx = torch.empty((F * S, Cout), dtype=torch.float32, device=dev)
            mn = torch.empty((F * S, Cout), dtype=torch.float32, device=dev)
        if utab is not None and i < 2:
            # first layer by linearity: statistics of U[idx] - W_xyz . centroid (no matrix work), then the second layer from gathered rows
            y_cur = torch.empty((rows, Cout), dtype=torch.float32, device=dev) if i == 1 else None
            ops.sa_train_stage(i + 1, xyz, new_xyz, feats, ball, ws, bs, affs, st, y_out=y_cur, u_in=utab, sched=sched)
            y_prev = y2 = y_cur
        elif wide:
            y_cur = torch.empty((rows, Cout), dtype=torch.float32, device=dev) if i < 2 else None
            ops.sa_train_stage(i + 1, xyz, new_xyz, feats, ball, ws, bs, affs, st, y_out=y_cur, y_in=y_prev,
                               out_max=mx if i == 2 else None, out_min=mn if i == 2 else None, sched=sched)
            y_prev = y_cur
        else:
            if feats is not None and i == 1:
                y2 = torch.empty((rows, Cout), dtype=torch.float32, device=dev)
            # level 2: stage 2 writes the raw second-layer rows, stage 3 (weights of the third convolution resident in LDS) reads them
            ops.sa_train_stage(i + 1, xyz, new_xyz, feats, ball, ws, bs, affs, st, y_out=y2 if i >= 1 else None,
                               out_max=mx if i == 2 else None, out_min=mn if i == 2 else None, sched=sched if i == 2 else None)
        affs.append(T.bn_finalize(st, rows, pk[f"{name}.g{i}"], pk[f"{name}.be{i}"], pk[f"{name}.rm{i}"], pk[f"{name}.rv{i}"],
                                  momentum=0.1, eps=1e-5))
    torch._foreach_add_([pk[f"{name}.nbt{i}"] for i in range(3)], 1)
    return T.bn_minmax_apply(mx, mn, affs[2][0], affs[2][1])


def _sa_mlp_train(pk, name: str, A: Optional[torch.Tensor], nsample: int, grp=None) -> torch.Tensor:
    """3 x [1x1 conv -> BatchNorm (batch statistics, running buffers updated) -> ReLU], max over nsample
    (utils/pn2_utils.py:210-216 with the module in .train())"""
    from . import train_ops as T

    if ops.GEMM_MODE == "f16x3" and BN_FUSED and SA_TRAIN_CHAIN and grp is not None:
        widths = tuple(pk[f"{name}.w{i}"].N for i in range(3))
        feats = grp[2]
        if (feats is None and nsample == 32 and widths == (64, 64, 128)) or \
           (feats is not None and nsample == 64 and feats.shape[2] == 128 and widths == (128, 128, 256)) or \
           (SA_TRAIN_WIDE and feats is not None and nsample == 64 and feats.shape[2] == 256 and widths == (256, 256, 512)):
            return _sa_chain_train(pk, name, grp, nsample)
    if ops.GEMM_MODE == "f16x3" and BN_FUSED:
        # fused form: batch statistics come out of the producing GEMM's epilogue, normalise+ReLU is applied by the
        # consuming GEMM while it stages its A tiles, and the last layer emits per-group max AND min instead of
        # its [rows, C] activation (max_p relu(a*y_p + b) = relu(a*(a >= 0 ? max_p y_p : min_p y_p) + b))
        dev = A.device if A is not None else grp[0].device
        rows = A.shape[0] if A is not None else grp[3].numel()
        h, aff = A, None
        for i in range(3):
            Cout = pk[f"{name}.w{i}"].N
            st = pk.get(f"{name}.stats{i}")
            if st is None:                   # allocated (and zeroed) once: pfpp_bn_finalize clears the copies it has summed
                st = pk[f"{name}.stats{i}"] = T.bn_stats_buffer(Cout, dev)
            if i == 0 and A is None:
                h = ops.grouped_linear(*grp, pk[f"{name}.w0"], pk[f"{name}.b0"], stats=st)
            elif i < 2:
                h = ops.linear(h, pk[f"{name}.w{i}"], pk[f"{name}.b{i}"], a_affine=aff, stats=st)
            else:
                mn = torch.empty((rows // nsample, Cout), dtype=torch.float32, device=dev)
                mx = ops.linear(h, pk[f"{name}.w{i}"], pk[f"{name}.b{i}"], a_affine=aff, stats=st, pool=nsample, c_min=mn)
            aff = T.bn_finalize(st, rows, pk[f"{name}.g{i}"], pk[f"{name}.be{i}"], pk[f"{name}.rm{i}"], pk[f"{name}.rv{i}"],
                                momentum=0.1, eps=1e-5)
        torch._foreach_add_([pk[f"{name}.nbt{i}"] for i in range(3)], 1)          # num_batches_tracked of the level: one launch
        return T.bn_minmax_apply(mx, mn, aff[0], aff[1])
    h = A
    for i in range(3):
        y = ops.linear(h, pk[f"{name}.w{i}"], pk[f"{name}.b{i}"])
        mean, var = T.bn_stats(y, pk[f"{name}.rm{i}"], pk[f"{name}.rv{i}"], momentum=0.1)
        pk[f"{name}.nbt{i}"] += 1
        h = T.bn_apply(y, mean, var, pk[f"{name}.g{i}"], pk[f"{name}.be{i}"], eps=1e-5, pool=nsample if i == 2 else 0)
        del y
    return h


def _sa_rows_eval(pk, name: str, grp, nsample: int) -> torch.Tensor:
    """eval-mode level with input features (sa3: 256 + 3 -> 256 -> 256 -> 512; sa2: 128 + 3 -> 128 -> 128 -> 256; 64 neighbours) on the ROWS kernels of the train-mode chain
    (csrc/sa_train.hip sa_wide_train_kernel<256, 2, UG> / <256, 3>: a workgroup keeps a 128-column slice of the layer's weight planes in
    LDS for its lifetime, a wave streams 32 rows at a time) instead of an elementwise pass + two tiled plane GEMMs with their
    [rows, 256] planes in between: 430 instead of 766 us at 154 fragments.  Same entry point as training (pfpp_sa_train_stage) with the
    FOLDED BatchNorm scale / shift as the layers' affines and zero conv biases (the folded shift carries them, pn2_utils.py:210-216 in
    .eval()): layer 1 per point (U[point] - W_xyz . centroid), layer 2 from gathered table rows -> raw y_2, layer 3 -> per-neighbourhood
    max / min of y_3, and max_p relu(s y_p + t) = relu(s (s >= 0 ? max y : min y) + t) exactly (monotone).  The statistics the train
    kernels also accumulate go to a scratch buffer nobody reads."""
    from . import train_ops as T

    xyz, new_xyz, feats, ball = grp
    dev = xyz.device
    F, S = ball.shape[:2]
    rows = F * S * nsample
    ws = [pk[f"{name}.w{i}"] for i in range(3)]
    aff = [(pk[f"{name}.s{i}"], pk[f"{name}.t{i}"]) for i in range(3)]
    sc = pk.get(f"{name}._rows_eval")
    if sc is None:
        sc = pk[f"{name}._rows_eval"] = ([torch.zeros(w.N, dtype=torch.float32, device=dev) for w in ws],
                                         [T.bn_stats_buffer(w.N, dev) for w in ws])
    zb, st = sc
    u = ops.sa_first_table(xyz, feats, ws[0], None)
    y2 = torch.empty((rows, ws[1].N), dtype=torch.float32, device=dev)
    sched = ops.sa_pad_schedule(ball) if (SA_PAD_SKIP and F * S >= SA_PAD_SKIP_MIN) else None      # padded second halves add nothing to a max / min
    ops.sa_train_stage(2, xyz, new_xyz, feats, ball, ws, zb, aff[:1], st[1], y_out=y2, u_in=u, sched=sched)
    mx = torch.empty((F * S, ws[2].N), dtype=torch.float32, device=dev)
    mn = torch.empty((F * S, ws[2].N), dtype=torch.float32, device=dev)
    if feats.shape[2] == 256:      # level 3: one rows launch per layer, the previous layer's raw rows come in as y_in
        ops.sa_train_stage(3, xyz, new_xyz, feats, ball, ws, zb, aff[:2], st[2], y_in=y2, out_max=mx, out_min=mn, sched=sched)
    else:                          # level 2: stage 3 reads the raw rows stage 2 wrote, its 256 x 128 weight planes resident in LDS
        ops.sa_train_stage(3, xyz, new_xyz, feats, ball, ws, zb, aff[:2], st[2], y_out=y2, out_max=mx, out_min=mn, sched=sched)
    return T.bn_minmax_apply(mx, mn, aff[2][0], aff[2][1])


def set_abstraction(pk, name: str, npoint: int, radius: float, nsample: int, xyz: torch.Tensor,
                    feats: Optional[torch.Tensor], capture: Optional[dict] = None, sampled=None):
    """xyz [F,N,3], feats [F,N,D] or None -> new_xyz [F,S,3], new_feats [F,S,C3].  sampled = (fps_idx, new_xyz, ball_idx) when the
    sampling of all levels was done up front (ops.sample_levels)"""
    F = xyz.shape[0]
    if sampled is not None:
        fps_idx, new_xyz, ball = sampled
    else:
        ops.check_fps_ratio(npoint, xyz.shape[1])
        fps_idx, new_xyz = ops.fps(xyz, npoint)
        ball = ops.ball_query(xyz, new_xyz, radius, nsample)
    fused = GATHER_FUSED and ops.GEMM_MODE == "f16x3" and (feats is None or feats.shape[2] % 32 == 0)
    A = None if fused else ops.group_gather(xyz, new_xyz, feats, ball)
    grp = (xyz, new_xyz, None if feats is None else feats.contiguous(), ball)
    if pk.get("train", False):
        if fused and not BN_FUSED:
            A = ops.group_gather(xyz, new_xyz, feats, ball)
        h = _sa_mlp_train(pk, name, A, nsample, grp)
        del A
    elif (SA_FUSED and fused and feats is None and nsample == 32
          and (pk[f"{name}.w0"].N, pk[f"{name}.w1"].N, pk[f"{name}.w2"].N) == (64, 64, 128)):
        h = ops.sa_mlp3_fused(xyz, new_xyz, ball, pk[f"{name}.w0"], pk[f"{name}.w1"], pk[f"{name}.w2"], pk[f"{name}.s0"], pk[f"{name}.t0"],
                              pk[f"{name}.s1"], pk[f"{name}.t1"], pk[f"{name}.s2"], pk[f"{name}.t2"])
        new_feats = h.view(F, npoint, -1)
        if capture is not None:
            capture[f"{name}.fps_idx"] = fps_idx
            capture[f"{name}.ball_idx"] = ball
            capture[f"{name}.new_xyz"] = new_xyz
            capture[f"{name}.new_points"] = new_feats
        return new_xyz, new_feats
    elif (SA_EVAL_ROWS and fused and ops.split_mode() and ops.GEMM_MODE == "f16x3" and not ops.SINGLE_PASS and feats is not None and nsample == 64
          and feats.shape[2] == 256 and (pk[f"{name}.w0"].N, pk[f"{name}.w1"].N, pk[f"{name}.w2"].N) == (256, 256, 512)
          and F * npoint * nsample >= SA_EVAL_ROWS_MIN):
        h = _sa_rows_eval(pk, name, grp, nsample)
    elif (SA_EVAL_ROWS2 and fused and ops.split_mode() and ops.GEMM_MODE == "f16x3" and not ops.SINGLE_PASS and feats is not None and nsample == 64
          and feats.shape[2] == 128 and (pk[f"{name}.w0"].N, pk[f"{name}.w1"].N, pk[f"{name}.w2"].N) == (128, 128, 256)
          and F * npoint * nsample >= SA_EVAL_ROWS2_MIN):
        h = _sa_rows_eval(pk, name, grp, nsample)
    elif (SA_FUSED and fused and feats is not None and nsample == 64 and feats.shape[2] == 128
          and (pk[f"{name}.w0"].N, pk[f"{name}.w1"].N) == (128, 128)):
        # level 2: grouping + layers 1 and 2 in one kernel, layer 3 (+ max over nsample) as a GEMM
        # the activation goes to layer 3 as split-f16 planes (same bytes as fp32): layer 3 is then the LDS-DMA plane GEMM with no
        # conversion work in its loop
        if SA_EVAL_UTAB and ops.split_mode() and ops.GEMM_MODE == "f16x3":
            # first layer per point (linear): the grouped first convolution is not computed (ops.sa_mlp2_table)
            h = ops.sa_mlp2_table(xyz, new_xyz, grp[2], ball, pk[f"{name}.w0"], pk[f"{name}.w1"], pk[f"{name}.s0"], pk[f"{name}.t0"],
                                  pk[f"{name}.s1"], pk[f"{name}.t1"])
        else:
            h = ops.sa_mlp2_fused(xyz, new_xyz, grp[2], ball, pk[f"{name}.w0"], pk[f"{name}.w1"], pk[f"{name}.s0"], pk[f"{name}.t0"],
                                  pk[f"{name}.s1"], pk[f"{name}.t1"], as_planes=ops.split_mode())
        h = ops.linear(h, pk[f"{name}.w2"], scale=pk[f"{name}.s2"], shift=pk[f"{name}.t2"], act="relu", pool=nsample)
        new_feats = h.view(F, npoint, -1)
        if capture is not None:
            capture[f"{name}.fps_idx"] = fps_idx
            capture[f"{name}.ball_idx"] = ball
            capture[f"{name}.new_xyz"] = new_xyz
            capture[f"{name}.new_points"] = new_feats
        return new_xyz, new_feats
    else:
        rows = F * npoint * nsample
        sp = ops.split_mode() and ops.GEMM_MODE == "f16x3"      # activations between the layers as split-f16 planes
        if (fused and sp and SA_EVAL_UTAB >= 2 and feats is not None and nsample == 64
                and (feats.shape[2], pk[f"{name}.w0"].N) in ((256, 256), (128, 128))):
            # first layer per point (linear), then an elementwise pass over the grouped rows (ops.sa_table_planes)
            h = ops.sa_table_planes(grp[0], grp[1], grp[2], grp[3], pk[f"{name}.w0"], pk[f"{name}.s0"], pk[f"{name}.t0"])
        elif fused:
            h = ops.grouped_linear(*grp, pk[f"{name}.w0"], scale=pk[f"{name}.s0"], shift=pk[f"{name}.t0"], act="relu",
                                   out=ops.SplitAct.empty(rows, pk[f"{name}.w0"].N, xyz.device) if sp else None)
        else:
            h = ops.linear(A, pk[f"{name}.w0"], scale=pk[f"{name}.s0"], shift=pk[f"{name}.t0"], act="relu",
                           out=ops.SplitAct.empty(rows, pk[f"{name}.w0"].N, xyz.device) if sp else None)
        del A
        h = ops.linear(h, pk[f"{name}.w1"], scale=pk[f"{name}.s1"], shift=pk[f"{name}.t1"], act="relu",
                       out=ops.SplitAct.empty(rows, pk[f"{name}.w1"].N, xyz.device) if sp else None)
        h = ops.linear(h, pk[f"{name}.w2"], scale=pk[f"{name}.s2"], shift=pk[f"{name}.t2"], act="relu", pool=nsample)
    new_feats = h.view(F, npoint, -1)
    if capture is not None:
        capture[f"{name}.fps_idx"] = fps_idx
        capture[f"{name}.ball_idx"] = ball
        capture[f"{name}.new_xyz"] = new_xyz
        capture[f"{name}.new_points"] = new_feats
    return new_xyz, new_feats


def pn2_encode(pk, pts: torch.Tensor, num_point: int = 25, capture: Optional[dict] = None):
    """pts [F,N,3] (already rotated) -> z_e [F,L,64], xyz [F,L,3]   (pn2.py:57-68)"""
    xyz, feats = pts, None
    # the sampling chain of all three levels depends on coordinates only: one launch (FPS x 3 + ball query x 3 per fragment)
    lv = tuple((npoint or num_point, radius, nsample) for _, npoint, radius, nsample in SA_LEVELS)
    # (a handful of fragments — one puzzle in flight — is latency-bound either way; there the per-level kernels' wider ball-query grids
    # win: 171 vs 189 us at F = 8)
    sampled = (ops.sample_levels(pts, lv) if (SAMPLE_FUSED and pts.shape[0] >= SAMPLE_FUSED_MIN and ops.sample_levels_supported(pts.shape[1], lv))
               else (None,) * 3)
    for (name, npoint, radius, nsample), smp in zip(SA_LEVELS, sampled):
        xyz, feats = set_abstraction(pk, name, npoint or num_point, radius, nsample, xyz, feats, capture, sampled=smp)
    F, L, C3 = feats.shape
    z_e = ops.linear(feats.view(F * L, C3), pk["conv6.w"], pk["conv6.b"]).view(F, L, -1)
    return z_e, xyz


def encode_valid(pk, pts: torch.Tensor, num_point: int = 25, max_frag: int = 2048):
    """VQVAE.encode on a dense list of fragments [F,N,3] -> {"z_q": [F,L,64], "xyz": [F,L,3]}"""
    F = pts.shape[0]
    slot = torch.arange(F, dtype=torch.int32, device=pts.device)
    z_q = torch.empty((F, num_point, pk["conv6.w"].N), dtype=torch.float32, device=pts.device)
    xyz_out = torch.empty((F, num_point, 3), dtype=torch.float32, device=pts.device)
    for f0 in range(0, F, max_frag):
        f1 = min(F, f0 + max_frag)
        z_e, xyz = pn2_encode(pk, pts[f0:f1].contiguous(), num_point)
        ops.vq_encode(z_e, pk["codebook"], slot[: f1 - f0].contiguous(), f1 - f0, z_q=z_q[f0:f1])
        xyz_out[f0:f1] = xyz
    return {"z_q": z_q, "xyz": xyz_out}


def extract_features(pk, part_pcs: torch.Tensor, pose: torch.Tensor, slot: torch.Tensor,
                     num_point: int = 25, max_frag: int = 2048, capture: Optional[dict] = None):
    """Denoiser._extract_features (denoiser.py:66-77): part_pcs [B,P,N,3], pose [B,P,7],
    slot = flattened indices of the valid fragments (int32, ascending)
    -> latent [B,P,L,64], xyz [B,P,L,3] with zeros in the padded slots."""
    B, P, N, _ = part_pcs.shape
    n_slots = B * P
    dev = part_pcs.device
    # both padded outputs out of ONE zero fill (one launch instead of two on the one-puzzle-in-flight chain)
    Cz = pk["conv6.w"].N
    n_lat = n_slots * num_point * Cz
    both = torch.zeros(n_lat + n_slots * num_point * 3, dtype=torch.float32, device=dev)
    latent = both[:n_lat].view(n_slots, num_point, Cz)
    xyz_out = both[n_lat:].view(n_slots, num_point, 3)
    pcs_flat = part_pcs.view(n_slots, N, 3)
    pose_flat = pose.reshape(n_slots, 7).contiguous()
    F = slot.numel()
    for f0 in range(0, F, max_frag):
        sl = slot[f0:f0 + max_frag].contiguous()
        rot = ops.se3_rotate_gather(pcs_flat, pose_flat, sl)
        if capture is not None:
            capture["rotated"] = rot
        z_e, xyz = pn2_encode(pk, rot, num_point, capture)
        if capture is not None:
            capture["z_e"] = z_e
        ops.vq_encode(z_e, pk["codebook"], sl, n_slots, z_q=latent)
        ops.scatter_rows(xyz, sl, n_slots, out=xyz_out)
    return latent.view(B, P, num_point, -1), xyz_out.view(B, P, num_point, 3)
