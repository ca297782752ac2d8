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
  * node merging, :224-286 — `_merge_components`: connected components of the accepted non-reference edges,
    merged cloud = posed clouds of the component minus their centroid, intersect filter + FPS back to 1000 points
    (utils/node_merge_utils.py on the HIP kernels), renormalisation, pivot / init-pose bookkeeping.  `merge_fn`
    may replace it.  Statistical parity only: the reference's FPS starts at a random index.
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


def normalise_edge_hist(ef: torch.Tensor) -> torch.Tensor:
    """[.., E, 6] histogram counts -> [.., E, 7]: the six bins divided by the edge's number of matched points (1 where there are
    none) and that number as the seventh feature (auto_aggl.py:198-201)"""
    cnt = ef.sum(dim=-1, keepdim=True)
    return torch.cat((ef / torch.where(cnt == 0, 1, cnt), cnt), dim=-1).float()


def edge_features_from_hist(hist_pp: torch.Tensor):
    """[B, P, P, 6] per-pair histograms (row = idx1 < column = idx2) -> (edge features [B, P(P-1)/2, 7], edge indices
    [1, P(P-1)/2, 2]) in the upper-triangle order of torch.triu(...).nonzero() (auto_aggl.py:195-201)"""
    B, P = hist_pp.shape[:2]
    pairs = AutoAgglomerative._edge_pairs(P, hist_pp.device)
    return normalise_edge_hist(hist_pp[:, pairs[:, 0], pairs[:, 1]]), pairs.unsqueeze(0)


class AutoAgglomerative(LightningModule):
    def __init__(self, cfg, merge_fn: Optional[Callable] = None, use_graphs: Optional[bool] = None):
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
        # One puzzle per call means ~120 small launches per DDPM step: the step is launch-bound.  Within an outer
        # iteration part_pcs / part_valids / part_scale / ref_part are fixed, so "rotate + encode + denoise" is captured
        # once into a HIP graph (static x / timestep buffers) and replayed for the remaining timesteps.
        import os
        self.use_graphs = (os.environ.get("PFPP_AGGL_GRAPHS", "0") == "1") if use_graphs is None else use_graphs   # opt-in: the B = 1 step is GPU-latency bound (1.9 ms eager = 1.9 ms replayed), capture per outer iteration costs more than the 1.4 ms/step of host time it saves

    # ------------------------------------------------------------------ helpers
    def _extract_features(self, part_pcs, part_valids, x):
        return self.encoder.extract_features(part_pcs, part_valids, x)

    _PAIRS = {}

    @classmethod
    def _edge_pairs(cls, P: int, device) -> torch.Tensor:
        """all (i, j), i < j, of P slots in the order of torch.triu(...).nonzero() = itertools.combinations — built once per (P, device)"""
        key = (P, str(device))
        if key not in cls._PAIRS:
            cls._PAIRS[key] = torch.tensor(list(itertools.combinations(range(P), 2)), dtype=torch.int64, device=device)
        return cls._PAIRS[key]

    @classmethod
    def _edge_mask(cls, num_parts: torch.Tensor, P: int) -> torch.Tensor:
        e = cls._edge_pairs(P, num_parts.device)
        return (e[None, :, 0] < num_parts[:, None]) & (e[None, :, 1] < num_parts[:, None])

    @staticmethod
    def prepare_matching(data_dict, device):
        """flatten the per-edge correspondence lists of the matching data into index arrays once:
        global index of a matched point = start(part) + critical_pcs_idx[start(part) + corr]
        (get_distance_for_matching_pts, node_merge_utils.py:62-89)"""
        n_pcs = data_dict["n_pcs"][0].cpu().numpy().astype(np.int64)
        crit = data_dict["critical_pcs_idx"][0].cpu().numpy().astype(np.int64)
        edges = data_dict["edges"][0].cpu().numpy().astype(np.int64)
        start = np.cumsum(n_pcs) - n_pcs
        as_np = lambda c: (c.detach().cpu().numpy() if torch.is_tensor(c) else np.asarray(c)).reshape(-1, 2).astype(np.int64)
        corrs = [as_np(data_dict["correspondences"][e]) for e in range(edges.shape[0])]
        counts = np.array([c.shape[0] for c in corrs], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(counts)])
        if corrs:
            corr = np.concatenate(corrs, 0)
            s1 = np.repeat(start[edges[:, 1]], counts)          # idx1 = edges[e, 1] owns column 0
            s2 = np.repeat(start[edges[:, 0]], counts)          # idx2 = edges[e, 0] owns column 1
            ia = s1 + crit[s1 + corr[:, 0]]
            ib = s2 + crit[s2 + corr[:, 1]]
        else:
            ia = ib = np.zeros(0, np.int64)
        pairs = [(int(e[1]), int(e[0])) for e in edges]
        P = int(data_dict["n_pcs"].shape[1])                  # n_pcs is [1, P] (one entry per fragment slot)
        i1, i2 = edges[:, 1], edges[:, 0]
        # position of edge (i1 < i2) in the row-major upper triangle = the order of the verifier's 190 edge slots
        pair_pos = i1 * P - i1 * (i1 + 1) // 2 + (i2 - i1 - 1)
        if edges.size and ((i1 >= i2).any() or (i2 >= P).any()):
            raise ValueError("prepare_matching: edges must be (idx2, idx1) with idx1 < idx2 < P")
        point_part = np.repeat(np.arange(n_pcs.size), n_pcs)
        dev_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)
        return {
            "idx_a": dev_i32(ia), "idx_b": dev_i32(ib), "edge_off": dev_i32(off),
            "max_m": int(counts.max()) if counts.size else 0, "pairs": pairs,
            "point_part": dev_i32(point_part),
            "pair_pos": torch.from_numpy(np.ascontiguousarray(pair_pos)).to(device),
        }

    # ------------------------------------------------------------------ test_step / test_batch
    @torch.no_grad()
    def test_step(self, data_dict, idx=0, x_init: Optional[torch.Tensor] = None, noises: Optional[List[torch.Tensor]] = None):
        """one puzzle through the auto-agglomerative loop (auto_aggl.py:86-318), batch size 1 like the reference"""
        if data_dict["part_pcs"].shape[0] != 1:
            raise ValueError("AutoAgglomerative.test_step handles one puzzle per call (docs/test.md:8); use test_batch for several")
        return self.test_batch([data_dict], None if x_init is None else [x_init], None if noises is None else [noises])[0]

    @torch.no_grad()
    def test_batch(self, data_dicts: List[dict], x_inits: Optional[List[torch.Tensor]] = None,
                   noises: Optional[List[List[torch.Tensor]]] = None) -> List[dict]:
        """Several independent puzzles through the loop at once (throughput mode).  The reference runs one puzzle per
        call; puzzles never interact (attention is within a puzzle, batch statistics are not used in eval), so the
        20 DDPM steps of an outer iteration run as ONE batched rotate + encode + denoise over all puzzles still
        active, while the verifier / promotion / merge bookkeeping stays per puzzle.  Every puzzle sees exactly the
        arithmetic of its own test_step up to the summation order inside the batched GEMMs."""
        states = [_PuzzleState(self, d, None if x_inits is None else x_inits[i], None if noises is None else noises[i])
                  for i, d in enumerate(data_dicts)]
        max_iters = self.cfg.verifier.max_iters
        from pfpp_hip.denoiser import CompactLayout

        for it in range(max_iters):
            active = [st for st in states if not st.done]
            if not active:
                break
            cat = (lambda key: torch.cat([getattr(st, key) for st in active], 0)) if len(active) > 1 else (lambda key: getattr(active[0], key))
            part_pcs, part_valids, part_scale, ref_part, reference = (cat(k) for k in ("part_pcs", "part_valids", "part_scale",
                                                                                       "ref_part", "reference"))
            x = cat("x")
            # everything that depends on part_valids only, once per outer iteration (no device->host reads in the steps)
            layout = CompactLayout(part_valids, self.num_points)
            step_fn = self._make_step(part_pcs, part_valids, part_scale, ref_part, layout, x)
            compose = self._batched_compose(active)                                # pivots / init poses are fixed until the merges
            timesteps = self.noise_scheduler.timesteps.tolist()

            def run_steps(fn, x):
                out = []
                for k, t in enumerate(timesteps):
                    eps = fn(x, t)
                    vn = None
                    if noises is not None:
                        vn = torch.cat([st.noises[st.step_no + k] for st in active], 0)
                    x = self.noise_scheduler.step(eps, t, x, variance_noise=vn, ref_part=ref_part, reference=reference).prev_sample
                    out.append(compose(x))                                         # get_param (:151), one launch for all puzzles
                return x, out

            rng = torch.cuda.get_rng_state(x.device) if noises is None else None
            x0 = x
            x, composed = run_steps(step_fn, x0)
            # range guard of the split-f16 GEMMs (weights are pre-scaled at pack time, activations are not): non-finite poses mean an
            # operand reached the fp16 range — redo this outer iteration with the exact fp32 GEMMs from the same draws.  The
            # read-back sits where the loop synchronises anyway (edge features / verifier bookkeeping below)
            if ops.f16x3_range_fallback(x):
                import warnings

                warnings.warn("split-f16 GEMM operand out of the fp16 range: re-running the iteration with exact fp32 GEMMs")
                if rng is not None:
                    torch.cuda.set_rng_state(rng, x.device)
                graphs, self.use_graphs = self.use_graphs, False
                try:
                    with ops.exact_fp32():
                        x, composed = run_steps(self._make_step(part_pcs, part_valids, part_scale, ref_part, layout, x0), x0)
                finally:
                    self.use_graphs = graphs
            for st in active:
                st.step_no += len(timesteps)
            composed = torch.stack(composed, 0)                                    # [steps, sum n_nodes, 7]
            off = 0
            for i, st in enumerate(active):
                st.traj.append(composed[:, off:off + st.n_nodes])
                off += st.n_nodes
                st.x = x[i:i + 1].clone()
            last = it + 1 == max_iters
            # edge features per puzzle (ragged matching data), ONE verifier call for all of them, bookkeeping per puzzle
            todo = [st for st in active if not st.finish_if_last(last)]
            if todo:
                efs = [st.edge_features() for st in todo]
                logits = self.verifier(torch.cat(efs, 0).contiguous(), torch.cat([st.edge_indices for st in todo], 0).contiguous(),
                                       torch.cat([st.edge_valids for st in todo], 0))
                for i, st in enumerate(todo):
                    st.after_verify(logits[i:i + 1])
        # the four metrics for all puzzles in one batched evaluation (the evaluator's functions are batch functions)
        finals = [self._compose(st.x, st.pivot, st.nodes) for st in states]
        metrics = self._evaluate_many([st.data for st in states], finals, [st.n_nodes for st in states])
        return [st.result(finals[i], {k: v[i:i + 1] for k, v in metrics.items()}) for i, st in enumerate(states)]

    @staticmethod
    def _batched_compose(active):
        """-> f(x [n,P,7]) = composed poses of all original parts of all active puzzles, [sum n_nodes, 7] in one launch"""
        dev = active[0].dev
        P = active[0].P
        pivot = torch.cat([st.pivot + i * P for i, st in enumerate(active)]).contiguous()
        nodes = [n for st in active for n in st.nodes]
        if all(n["init_pose"] is None for n in nodes):
            return lambda x: ops.pose_compose(x.reshape(-1, 7).contiguous(), pivot)
        init = torch.stack([(n["init_pose"] if n["init_pose"] is not None else torch.eye(4, device=dev)).reshape(16) for n in nodes]).contiguous()
        has = torch.tensor([n["init_pose"] is not None for n in nodes], dtype=torch.uint8, device=dev)
        return lambda x: ops.pose_compose(x.reshape(-1, 7).contiguous(), pivot, init, has)

    def _make_step(self, part_pcs, part_valids, part_scale, ref_part, layout, x_like):
        """-> f(x, t) = predicted noise.  Eager: rotate + encode + denoise with the precomputed layout.  Graphs: the same
        sequence captured once (after an eager warm-up step that also fills the pack caches) over static buffers."""
        dev = x_like.device
        B = x_like.shape[0]

        def eager(x, ts):
            latent, xyz = self.encoder.extract_features(part_pcs, part_valids, x, slot=layout.slot32)
            return self.denoiser(x, ts, latent, xyz, part_valids, part_scale, ref_part, layout=layout)

        if not self.use_graphs:
            def step(x, t):
                ts = torch.full((B,), t, dtype=torch.int64, device=dev)
                ts._pfpp_t = int(t)      # one timestep for the whole batch: AdaLN rows cached per (t, B)
                return eager(x, ts)

            return step
        state = {"graph": None, "calls": 0}
        x_buf = torch.empty_like(x_like)
        ts_buf = torch.zeros((B,), dtype=torch.int64, device=dev)

        def step(x, t):
            state["calls"] += 1
            if state["calls"] == 1:                      # warm-up: lazy initialisations must not happen under capture
                return eager(x, torch.full((B,), t, dtype=torch.int64, device=dev))
            x_buf.copy_(x)
            ts_buf.fill_(t)
            if state["graph"] is None:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize(dev)
                with torch.cuda.graph(g):
                    state["eps"] = eager(x_buf, ts_buf)
                state["graph"] = g
            state["graph"].replay()
            return state["eps"].clone()

        return step

    @staticmethod
    def _compose(x, pivot, nodes):
        """get_param / extract_final_pred_trans_rots (node_merge_utils.py:246-306): pose of every original part =
        [R|t](pose of its pivot) @ its accumulated init_pose"""
        if all(n["init_pose"] is None for n in nodes):
            return ops.pose_compose(x[0].contiguous(), pivot)
        dev = x.device
        init = torch.stack([(n["init_pose"] if n["init_pose"] is not None else torch.eye(4, device=dev)).reshape(16) for n in nodes])
        has = torch.tensor([n["init_pose"] is not None for n in nodes], dtype=torch.uint8, device=dev)
        return ops.pose_compose(x[0].contiguous(), pivot, init.contiguous(), has)

    def _merge_components(self, merges, nodes, st) -> None:
        """auto_aggl.py:224-286: merge every connected component of the accepted non-reference edges"""
        from utils.node_merge_utils import assign_init_pose, get_final_pose_pts, merge_node, remove_intersect_points_and_fps_ds

        x, part_pcs, part_scale, part_valids = st["x"], st["part_pcs"], st["part_scale"], st["part_valids"]
        st["merged_edges"].extend(merges)
        n = len(nodes)
        parent = list(range(n))                      # union-find over ALL edges merged so far (G keeps its edges)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for a, b in st["merged_edges"]:
            parent[find(a)] = find(b)
        comps = {}
        for i in range(n):
            comps.setdefault(find(i), []).append(i)
        pred_trans, pred_rots = x[0, :, :3], x[0, :, 3:]
        transformed = get_final_pose_pts(part_pcs * part_scale.unsqueeze(-1), x[..., :3], x[..., 3:])[0]     # [P,N,3]
        scale_h = part_scale[0, :, 0].cpu()
        n_pcs = st["n_pcs"]
        start = torch.cumsum(n_pcs[0].long(), 0) - n_pcs[0].long()
        for comp in comps.values():
            if sum(1 for c in comp if nodes[c]["valids"]) <= 1:
                continue
            pivot_new = max(comp, key=lambda c: float(scale_h[c]))
            merge_pcs = merge_node(comp, nodes, transformed)
            centroid = merge_pcs.mean(dim=0)
            merge_pcs = merge_pcs - centroid
            assign_init_pose(nodes, pred_trans, pred_rots, centroid, comp)
            for c in comp:                                    # the by-area points now live in the merged part's frame
                a, b = int(start[c]), int(start[c] + n_pcs[0, c])
                st["pts_by_area"][0, a:b] = st["pts_by_area_t"][a:b] - centroid
                nodes[c]["pivot"] = pivot_new
                st["pivot"][c] = pivot_new
            ds = remove_intersect_points_and_fps_ds(merge_pcs)
            m_scale = ds.abs().max()
            part_scale[0, pivot_new] = m_scale
            part_pcs[0, pivot_new] = ds / m_scale
            part_valids[0, comp] = 0
            part_valids[0, pivot_new] = 1
            for c in comp:
                nodes[c]["valids"] = c == pivot_new
            st["classified"][0, comp] = True

    def _evaluate(self, data_dict, final, n_nodes):
        """the four metrics of test_step (auto_aggl.py:288-318) on the composed final poses"""
        return self._evaluate_many([data_dict], [final], [n_nodes])

    def _evaluate_many(self, data_dicts, finals, n_nodes_list):
        """_evaluate for several puzzles at once: every metric is a [n_puzzles] tensor (same functions, batched inputs)"""
        from puzzlefusion_plusplus.denoiser.evaluation.evaluator import (ChamferDistance, calc_part_acc, calc_shape_cd,
                                                                        rot_metrics, trans_metrics)

        n = len(data_dicts)
        P = data_dicts[0]["part_valids"].shape[1]
        dev = finals[0].device
        pred = torch.zeros(n, P, 7, device=dev)
        pred[:, :, 3] = 1.0
        for i, (final, n_nodes) in enumerate(zip(finals, n_nodes_list)):
            pred[i, :n_nodes] = final
        cat = (lambda key: torch.cat([d[key] for d in data_dicts], 0)) if n > 1 else (lambda key: data_dicts[0][key])
        pts = (cat("part_pcs") * cat("part_scale").unsqueeze(-1)).float()
        valids = cat("part_valids")
        gt_t, gt_r = cat("part_trans").float(), cat("part_rots").float()
        gt_r = torch.where(gt_r.abs().sum(-1, keepdim=True) == 0, torch.tensor([1.0, 0, 0, 0], device=dev), gt_r)
        cd = ChamferDistance()
        pt, pr = pred[..., :3].contiguous(), pred[..., 3:].contiguous()
        acc, _, _ = calc_part_acc(pts, pt, gt_t, pr, gt_r, valids, cd)
        out = {"part_acc": acc, "shape_cd": calc_shape_cd(pts, pt, gt_t, pr, gt_r, valids, cd),
               "rmse_r": rot_metrics(pr, gt_r, valids, "rmse"), "rmse_t": trans_metrics(pt, gt_t, valids, "rmse")}
        for name, lst in (("part_acc", "acc_list"), ("rmse_r", "rmse_r_list"), ("rmse_t", "rmse_t_list"), ("shape_cd", "cd_list")):
            if not hasattr(self, lst):
                setattr(self, lst, [])
            getattr(self, lst).append(out[name])
        return out

    def on_test_epoch_end(self):
        """auto_aggl.py:360-375"""
        total = [torch.mean(torch.cat(v)) for v in (self.acc_list, self.rmse_t_list, self.rmse_r_list, self.cd_list)]
        for name, value in zip(("eval/part_acc", "eval/rmse_t", "eval/rmse_r", "eval/shape_cd"), total):
            self.log(name, value, sync_dist=True)
        self.acc_list, self.rmse_t_list, self.rmse_r_list, self.cd_list = [], [], [], []
        return tuple(total)

    def _save_inference_data(self, data_dict, trajectory, acc):
        """predict_<acc>.npy / gt.npy / init_pose.npy / mesh_file_path.txt per puzzle (auto_aggl.py:322-357)"""
        import os

        from pfpp_hip import io as pfio

        mask = (data_dict["part_valids"][0] == 1)
        n_nodes = trajectory.shape[1]
        did = data_dict["data_id"][0]
        save_dir = os.path.join(self.cfg.experiment_output_path, "inference", str(getattr(self.cfg, "inference_dir", "results")),
                                str(did.item() if hasattr(did, "item") else did))
        gt = torch.cat([data_dict["part_trans"][0], data_dict["part_rots"][0]], dim=-1)[mask]
        init = torch.cat([torch.as_tensor(data_dict["init_pose_t"][0]).float().cpu(), torch.as_tensor(data_dict["init_pose_r"][0]).float().cpu()], -1) \
            if "init_pose_t" in data_dict else torch.zeros(7)
        mesh = data_dict["mesh_file_path"][0] if "mesh_file_path" in data_dict else ""
        return pfio.save_inference_data(save_dir, trajectory=trajectory[:, mask[:n_nodes]].cpu().numpy(), gt=gt.cpu().numpy(),
                                        init_pose=init.numpy(), mesh_file_path=mesh, acc=float(acc[0]))

    def save_inference_data(self, out, path: str):
        np.save(path, out["trajectory"].cpu().numpy())


class _PuzzleState:
    """everything the loop carries for ONE puzzle between outer iterations (the locals of the reference's test_step)"""

    def __init__(self, model: "AutoAgglomerative", data_dict, x_init, noises):
        self.m = model
        dev = data_dict["part_pcs"].device
        self.dev = dev
        gt = torch.cat([data_dict["part_trans"], data_dict["part_rots"]], dim=-1).float().contiguous()
        B, P, N, _ = data_dict["part_pcs"].shape
        if B != 1:
            raise ValueError("every puzzle is passed as its own batch-1 dict")
        self.P = P
        self.x = torch.randn(gt.shape, device=dev) if x_init is None else x_init.clone()
        self.noises = noises
        self.ref_part = data_dict["ref_part"].clone()
        self.reference = torch.zeros_like(gt)
        self.reference[self.ref_part] = gt[self.ref_part]
        self.x[self.ref_part] = self.reference[self.ref_part]
        self.part_valids = data_dict["part_valids"].clone()
        self.part_scale = data_dict["part_scale"].clone()
        self.part_pcs = data_dict["part_pcs"].clone()
        self.num_parts = data_dict["num_parts"].clone()
        self.n_nodes = int(self.num_parts[0])
        self.nodes = [{"pivot": i, "valids": True, "ref_part": False, "init_pose": None} for i in range(self.n_nodes)]
        self.nodes[int(torch.where(self.ref_part)[1][0])]["ref_part"] = True
        self.classified = torch.zeros_like(self.part_valids, dtype=torch.bool)
        self.have_matching = "edges" in data_dict
        self.match = model.prepare_matching(data_dict, dev) if self.have_matching else None
        self.pivot = torch.arange(self.n_nodes, dtype=torch.int32, device=dev)
        self.edge_indices = model._edge_pairs(P, dev)[None]          # == triu(ones, 1).nonzero(): row-major upper triangle
        self.edge_valids = model._edge_mask(self.num_parts, P)
        self.traj, self.step_no, self.verifier_calls, self.n_merges = [], 0, 0, 0
        self.merged_edges: List = []
        self.data = dict(data_dict)
        if self.have_matching:
            self.data["part_pcs_by_area"] = data_dict["part_pcs_by_area"].clone()      # mutated by the merges (:259-262)
        self.done = False

    def finish_if_last(self, last: bool) -> bool:
        if last or not self.have_matching:
            self.done = True
        return self.done

    def edge_features(self) -> torch.Tensor:
        """[1, 190, 7] verifier input of this puzzle from its current poses (auto_aggl.py:153-201)"""
        m, dev, P = self.m, self.dev, self.P
        x, match, pivot = self.x, self.match, self.pivot
        # ---- edge features (:156-201) ---------------------------------------------------------
        pts_t = ops.pose_apply_points(self.data["part_pcs_by_area"][0].float().contiguous(),
                                      pivot[match["point_part"].long()].contiguous(), x[0].contiguous(), normalise=False)
        hist = ops.edge_histogram(pts_t, match["idx_a"], match["idx_b"], match["edge_off"], match["max_m"])
        # the reference scatters into a [P,P,6] matrix and reads its upper triangle (auto_aggl.py:193-197): same thing with the
        # edges' triangle positions precomputed (no index tensors built from Python lists, no boolean-mask read-back per call)
        ef = torch.zeros(1, P * (P - 1) // 2, 6, dtype=torch.int32, device=dev)
        if match["pairs"]:
            ef[0, match["pair_pos"]] = hist
        self._pts_t = pts_t
        return normalise_edge_hist(ef)

    def after_verify(self, logits: torch.Tensor) -> None:
        """threshold -> reference promotion -> merge for this puzzle (auto_aggl.py:203-286); sets `done`"""
        m, x, match, pivot, pts_t = self.m, self.x, self.match, self.pivot, self._pts_t
        self.verifier_calls += 1
        pred = (torch.sigmoid(logits) > m.cfg.verifier.threshold).squeeze(-1) & self.edge_valids
        classified_edges = self.edge_indices[pred].cpu().tolist()
        # ---- reference promotion (:208-222) -----------------------------------------------------
        valid_b = self.part_valids.bool()
        ref_idx = set(torch.where(self.ref_part)[1].cpu().tolist())
        self.classified[0, list(ref_idx)] = True
        larger = valid_b & (self.part_scale.squeeze(2) > 0.05)
        new_ref = [a if a not in ref_idx else b for a, b in classified_edges if (a in ref_idx) != (b in ref_idx)]
        for j in new_ref:
            self.ref_part[0, j] = True
        self.reference = x.clone()
        if bool((self.classified == larger).all()):
            self.done = True
            return
        from utils.node_merge_utils import node_merge_valids_check

        merges = [(a, b) for a, b in classified_edges if node_merge_valids_check((a, b), self.ref_part, self.nodes)]
        if merges:
            state = dict(x=x, part_pcs=self.part_pcs, part_scale=self.part_scale, part_valids=self.part_valids,
                         classified=self.classified, pivot=pivot, pts_by_area=self.data["part_pcs_by_area"], pts_by_area_t=pts_t,
                         match=match, n_pcs=self.data["n_pcs"], merged_edges=self.merged_edges)
            if m.merge_fn is not None:
                m.merge_fn(m, merges, self.nodes, state)
            else:
                m._merge_components(merges, self.nodes, state)
            self.n_merges += 1
        if bool((self.classified == larger).all()):
            self.done = True

    def result(self, final=None, metrics=None) -> dict:
        m = self.m
        if final is None:
            final = m._compose(self.x, self.pivot, self.nodes)
        valid_nodes = self.data["part_valids"][0, :self.n_nodes].bool()
        if metrics is None:
            metrics = m._evaluate(self.data, final, self.n_nodes)
        traj_t = torch.cat(self.traj, 0)
        if getattr(m.cfg, "experiment_output_path", None) is not None and "data_id" in self.data:
            m._save_inference_data(self.data, traj_t, metrics["part_acc"])
        return {
            "metrics": metrics,
            "pred_trans": final[:, :3], "pred_rots": final[:, 3:], "x": self.x,
            "trajectory": traj_t[:, valid_nodes],                        # [T_total, Pv, 7] like predict_*.npy (:322-337)
            "ref_part": self.ref_part, "verifier_calls": self.verifier_calls, "steps": self.step_no, "merges": self.n_merges,
            "part_valids": self.part_valids, "nodes": self.nodes,
        }
