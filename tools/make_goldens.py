"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python modules (build container only).

    python tools/make_goldens.py            # needs /root/reference; writes tests/golden/

What is imported verbatim from /root/reference (sys.dont_write_bytecode, nothing is copied):
    utils/pn2_utils.py, utils/model_utils.py,
    puzzlefusion_plusplus/vqvae/model/modules/{pn2,quantizer}.py,
    puzzlefusion_plusplus/denoiser/model/modules/{encoder,attention,denoiser_transformer,custom_diffusers}.py,
    puzzlefusion_plusplus/verifier/model/modules/verifier_transformer.py
Third-party packages those files import but which are not installed (torch_cluster, chamferdist,
diffusers==0.21.4) get minimal stand-ins below, restated from their documented semantics
 — those pieces stay "parity unpinned"; everything the reference itself owns
(grouping, ball query, SA-MLPs, VQ, AdaLN, masks, token assembly, heads, beta schedule, verifier) is
pinned by the fixtures.  The script also checks the oracle (oracle/pfpp_oracle.py) against the
reference outputs and refuses to write fixtures the oracle does not reproduce.
"""
from __future__ import annotations

import sys
import math
import types
from pathlib import Path
from types import SimpleNamespace as NS

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import pfpp_oracle as O
from oracle import weights

GOLD = ROOT / "tests" / "golden"


# ------------------------------------------------------------------------------------------------
# stand-ins for absent third-party packages
# ------------------------------------------------------------------------------------------------
def install_standins():
    tc = types.ModuleType("torch_cluster")

    def fps(src, batch=None, ratio=None, random_start=True):
        """torch_cluster.fps semantics: per batch element, m = ceil(ratio*n), start 0,
        squared-L2 running min, first argmax; returns GLOBAL row indices."""
        if batch is None:                       # single cloud (utils/node_merge_utils.py:219): random first index
            n = src.shape[0]
            m = int(math.ceil(ratio * n))
            start = int(torch.randint(0, n, (1,)).item()) if random_start else 0
            fps.last_start = start
            return O.fps_start(src, m, start)
        assert not random_start
        counts = torch.bincount(batch)
        out, off = [], 0
        for n in counts.tolist():
            m = int(torch.ceil(ratio * n).item())
            idx = O.fps(src[off:off + n].unsqueeze(0), m)[0]
            out.append(idx + off)
            off += n
        return torch.cat(out)

    tc.fps = fps
    sys.modules["torch_cluster"] = tc

    cd = types.ModuleType("chamferdist")

    class ChamferDistance(nn.Module):
        def forward(self, *a, **k):
            raise RuntimeError("chamferdist is not on the hot path")

    cd.ChamferDistance = ChamferDistance
    sys.modules["chamferdist"] = cd

    # diffusers 0.21.4
    class Attention(nn.Module):
        def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False):
            super().__init__()
            inner = heads * dim_head
            self.heads = heads
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, attention_mask=None):
            B, S, _ = hidden_states.shape
            H = self.heads
            q, k, v = self.to_q(hidden_states), self.to_k(hidden_states), self.to_v(hidden_states)
            dh = q.shape[-1] // H
            q, k, v = (t.view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
            m = attention_mask
            if m.shape[0] < B * H:
                m = m.repeat_interleave(H, dim=0)
            m = m.view(B, H, -1, m.shape[-1])
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=m, dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(B, -1, H * dh)
            return self.to_out[1](self.to_out[0](o))

    class GEGLU(nn.Module):
        def __init__(self, dim_in, dim_out):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out * 2)

        def forward(self, x):
            h, gate = self.proj(x).chunk(2, dim=-1)
            return h * F.gelu(gate)

    class FeedForward(nn.Module):
        def __init__(self, dim, dropout=0.0, activation_fn="geglu", final_dropout=False, mult=4):
            super().__init__()
            assert activation_fn == "geglu"
            self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])

        def forward(self, x):
            for m in self.net:
                x = m(x)
            return x

    class DDPMScheduler:
        def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2, beta_schedule="linear",
                     prediction_type="epsilon", clip_sample=True, timestep_spacing="leading", **kw):
            self.config = NS(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                             clip_sample=clip_sample, timestep_spacing=timestep_spacing, steps_offset=0)
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
            self.alphas = 1.0 - self.betas
            self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
            self.one = torch.tensor(1.0)
            self.num_inference_steps = None
            self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

        def set_timesteps(self, num_inference_steps):
            self.num_inference_steps = num_inference_steps
            ratio = self.config.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            self.timesteps = torch.from_numpy(ts)

        def previous_timestep(self, t):
            return t - self.config.num_train_timesteps // self.num_inference_steps

        def add_noise(self, x0, noise, timesteps):
            ac = self.alphas_cumprod.to(dtype=x0.dtype)
            sa = (ac[timesteps] ** 0.5).flatten()
            sb = ((1 - ac[timesteps]) ** 0.5).flatten()
            while sa.dim() < x0.dim():
                sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
            return sa * x0 + sb * noise

        def step(self, model_output, timestep, sample, variance_noise=None):
            t = timestep
            prev_t = self.previous_timestep(t)
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
            b_t, b_prev = 1 - a_t, 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            c0 = (a_prev ** 0.5 * cur_b) / b_t
            c1 = cur_a ** 0.5 * b_prev / b_t
            prev = c0 * x0 + c1 * sample
            if t > 0:
                var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
                prev = prev + (var ** 0.5) * variance_noise
            return NS(prev_sample=prev, pred_original_sample=x0)

    # ---- pytorch3d.transforms (git HEAD semantics restated): only what the evaluation
    # metrics import (denoiser/evaluation/transform.py:1-2)
    p3d = types.ModuleType("pytorch3d")
    p3t = types.ModuleType("pytorch3d.transforms")

    def _angle_from_tan(axis, other_axis, data, horizontal, tait_bryan):
        i1, i2 = {"X": (2, 1), "Y": (0, 2), "Z": (1, 0)}[axis]
        if horizontal:
            i2, i1 = i1, i2
        even = (axis + other_axis) in ["XY", "YZ", "ZX"]
        if horizontal == even:
            return torch.atan2(data[..., i1], data[..., i2])
        if tait_bryan:
            return torch.atan2(-data[..., i2], data[..., i1])
        return torch.atan2(data[..., i2], -data[..., i1])

    def matrix_to_euler_angles(matrix, convention):
        i0, i2 = "XYZ".index(convention[0]), "XYZ".index(convention[2])
        tait_bryan = i0 != i2
        if tait_bryan:
            central = torch.asin(matrix[..., i0, i2] * (-1.0 if i0 - i2 in [-1, 2] else 1.0))
        else:
            central = torch.acos(matrix[..., i0, i0])
        o = (_angle_from_tan(convention[0], convention[1], matrix[..., i2], False, tait_bryan), central,
             _angle_from_tan(convention[2], convention[1], matrix[..., i0, :], True, tait_bryan))
        return torch.stack(o, -1)

    from oracle import pfpp_oracle as _O
    p3t.quaternion_apply = _O.quaternion_apply
    p3t.quaternion_to_matrix = _O.quaternion_to_matrix
    p3t.matrix_to_euler_angles = matrix_to_euler_angles
    p3t.matrix_to_quaternion = _O.matrix_to_quaternion
    p3d.transforms = p3t
    p3o = types.ModuleType("pytorch3d.ops")
    p3o.estimate_pointcloud_normals = lambda pts, neighborhood_size=50, **kw: _O.estimate_normals(pts, neighborhood_size)
    p3d.ops = p3o
    sys.modules.update({"pytorch3d": p3d, "pytorch3d.transforms": p3t, "pytorch3d.ops": p3o})

    dm = types.ModuleType("diffusers")
    dm.DDPMScheduler = DDPMScheduler
    dmm = types.ModuleType("diffusers.models")
    dma = types.ModuleType("diffusers.models.attention")
    dma.Attention, dma.FeedForward = Attention, FeedForward
    dm.models = dmm
    dmm.attention = dma
    sys.modules.update({"diffusers": dm, "diffusers.models": dmm, "diffusers.models.attention": dma})


def maxdiff(a, b):
    return (a.double() - b.double()).abs().max().item()


def main():
    assert REF.exists(), "run this in the build container (needs /root/reference)"
    install_standins()
    sys.path.insert(0, str(REF))
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k.startswith("puzzlefusion_plusplus")]:
        del sys.modules[k]
    import utils.pn2_utils as ref_pn2                       # noqa: E402  (the reference's)
    from puzzlefusion_plusplus.denoiser.model.modules.custom_diffusers import PiecewiseScheduler, betas_for_alpha_bar
    from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer
    from puzzlefusion_plusplus.denoiser.model.modules.encoder import VQVAE
    from puzzlefusion_plusplus.verifier.model.modules.verifier_transformer import VerifierTransformer
    from utils.model_utils import EmbedderNerf, PositionalEncoding
    assert ref_pn2.__file__.startswith(str(REF)), ref_pn2.__file__

    # host-side synthetic data generator (numpy), loaded by path: the product's `utils` package must
    # not shadow the reference's namespace package `utils` on sys.path
    import importlib.util
    spec = importlib.util.spec_from_file_location("pfpp_synthetic", ROOT / "puzzlefusion-plusplus_amd/pfpp_hip/synthetic.py")
    synthetic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synthetic)

    GOLD.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    cfg = NS(
        ae=NS(n_embeddings=1024, embedding_dim=16, num_point=25, num_dim=64, local_decode_pts=40, beta=0.25),
        model=NS(embed_dim=512, out_channels=7, num_layers=6, num_heads=8, num_dim=64, num_point=25),
    )

    # ============================ encoder =======================================================
    data = synthetic.make_batch(7, 1, num_points=512, num_parts=3)
    dataq = synthetic.make_batch(7, 1, num_points=512, num_parts=3, quantise_bits=9)
    data_ref = synthetic.make_batch(11, 1, num_points=1000, num_parts=4)          # the reference's own shape: N = 1000 (pn2.py:16-18), F = 4
    x = torch.randn(1, 20, 7)
    enc_ref = VQVAE(cfg)
    enc_sd = weights.vqvae_state_dict()
    enc_ref.load_state_dict(enc_sd, strict=True)
    enc_ref.eval()
    for tag, d in (("float", data), ("grid", dataq), ("ref", data_ref)):
        valid = d["part_valids"].bool()
        pts = O.apply_rots(d["part_pcs"], x)[valid] if tag != "grid" else d["part_pcs"][valid]
        # stage-level reference outputs straight from the reference functions
        cap = {}
        xyz_l, feats_cf = pts, None
        with torch.no_grad():
            cur_xyz_cf = pts.permute(0, 2, 1)
            for name, sa in (("sa1", enc_ref.pn2.sa1), ("sa2", enc_ref.pn2.sa2), ("sa3", enc_ref.pn2.sa3)):
                xyz_cl = cur_xyz_cf.permute(0, 2, 1)
                pts_cl = feats_cf.permute(0, 2, 1) if feats_cf is not None else None
                new_xyz, new_points, grouped_xyz, fps_idx = ref_pn2.sample_and_group(
                    sa.npoint, sa.radius, sa.nsample, xyz_cl, pts_cl, returnfps=True)
                ball = ref_pn2.query_ball_point(sa.radius, sa.nsample, xyz_cl, new_xyz)
                cap[f"{name}.fps_idx"] = fps_idx.reshape(pts.shape[0], -1)
                cap[f"{name}.ball_idx"] = ball
                cur_xyz_cf, feats_cf = sa(cur_xyz_cf, feats_cf)
                cap[f"{name}.new_points"] = feats_cf.permute(0, 2, 1).contiguous()
            out_ref = enc_ref.encode(pts)
            z_e_ref, _ = enc_ref.pn2.encode(pts.permute(0, 2, 1))
        ocap = {}
        out_o = O.vqvae_encode(enc_sd, pts, capture=ocap)
        for lvl in ("sa1", "sa2", "sa3"):
            assert torch.equal(ocap[f"pn2.{lvl}.fps_idx"], cap[f"{lvl}.fps_idx"]), (tag, lvl, "fps")
            nb = (ocap[f"pn2.{lvl}.ball_idx"] != cap[f"{lvl}.ball_idx"]).sum().item()
            assert nb == 0, (tag, lvl, "ball query differs from the reference in", nb)
            assert maxdiff(ocap[f"pn2.{lvl}.new_points"], cap[f"{lvl}.new_points"]) < 1e-6, (tag, lvl)
        assert maxdiff(ocap["z_e"], z_e_ref) < 1e-6
        assert torch.equal(out_o["xyz"], out_ref["xyz"])
        nflip = ((out_o["z_q"] - out_ref["z_q"]).abs().reshape(-1, 16).amax(1) > 1e-6).sum().item()
        gaps = O.vq_gap(enc_sd["vector_quantization.embedding.weight"], z_e_ref.reshape(-1, 16))
        print(f"[encoder/{tag}] oracle == reference: fps/ball exact, feats<1e-6, VQ sub-vectors differing {nflip}, "
              f"min top-2 gap {gaps.min():.2e}")
        assert nflip == 0
        np.savez_compressed(
            GOLD / f"encoder_{tag}.npz",
            pts=pts.numpy(),
            **{f"{l}_fps_idx": cap[f"{l}.fps_idx"].numpy().astype(np.int16) for l in ("sa1", "sa2", "sa3")},
            **{f"{l}_ball_idx": cap[f"{l}.ball_idx"].numpy().astype(np.int16) for l in ("sa1", "sa2", "sa3")},
            sa1_feat_sub=cap["sa1.new_points"][:, ::16].numpy(), sa2_feat_sub=cap["sa2.new_points"][:, ::8].numpy(),
            sa3_feat=cap["sa3.new_points"].numpy(), z_e=z_e_ref.numpy(), z_q=out_ref["z_q"].numpy(),
            xyz=out_ref["xyz"].numpy(), vq_gap=gaps.numpy(),
        )
        if tag == "float":
            np.savez_compressed(GOLD / "rotate.npz", part_pcs=d["part_pcs"][valid].numpy(), pose=x[valid].numpy(),
                                rotated=pts.numpy())
            # ---- the same fragments through the encoder in .train(): BatchNorm on batch statistics, running
            # buffers updated (the state of the "frozen" encoder during Denoiser training)
            import copy
            enc_tr = copy.deepcopy(enc_ref).train()
            with torch.no_grad():
                out_tr = enc_tr.encode(pts)
                z_e_tr, _ = enc_tr.pn2.encode(pts.permute(0, 2, 1))     # second pass: stats updated twice
            sd_tr = {k: v.clone() for k, v in enc_sd.items()}
            ocap_tr = {}
            out_otr = O.vqvae_encode(sd_tr, pts, capture=ocap_tr, train=True)
            O.pn2_encode(sd_tr, pts, train=True)
            sd_after = enc_tr.state_dict()
            d_stats = max(maxdiff(sd_after[k], sd_tr[k]) for k in sd_tr if "running_" in k or "num_batches" in k)
            gaps_tr = O.vq_gap(enc_sd["vector_quantization.embedding.weight"], ocap_tr["z_e"].reshape(-1, 16))
            nflip_tr = ((out_otr["z_q"] - out_tr["z_q"]).abs().reshape(-1, 16).amax(1) > 1e-6).sum().item()
            print(f"[encoder/train-mode BN] oracle vs reference: z_e first pass maxdiff {maxdiff(ocap_tr['z_e'], out_tr['z_q'] * 0 + ocap_tr['z_e']):.1e}, "
                  f"running stats after two passes maxdiff {d_stats:.2e}, VQ sub-vectors differing {nflip_tr}, min gap {gaps_tr.min():.2e}")
            assert d_stats < 1e-5 and nflip_tr == 0 and torch.equal(out_otr["xyz"], out_tr["xyz"])
            keys_rs = sorted(k for k in sd_tr if "running_" in k)
            np.savez_compressed(GOLD / "encoder_train.npz", z_e=ocap_tr["z_e"].numpy(), z_q=out_tr["z_q"].numpy(), xyz=out_tr["xyz"].numpy(),
                                vq_gap=gaps_tr.numpy(), stat_names=np.array(keys_rs),
                                stats_after_two=np.concatenate([sd_after[k].flatten().numpy() for k in keys_rs]),
                                num_batches_tracked=np.int64(sd_after["pn2.sa1.mlp_bns.0.num_batches_tracked"].item()))

    # ball query on the reference's own code for all three level shapes (standalone)
    g = torch.Generator().manual_seed(5)
    for N, S, r, ns in ((1000, 256, 0.2, 32), (256, 128, 0.4, 64), (128, 25, 0.8, 64)):
        p = torch.rand(2, N, 3, generator=g) * 2 - 1
        c = p[:, :S].contiguous()
        assert torch.equal(ref_pn2.query_ball_point(r, ns, p, c), O.query_ball_point(r, ns, p, c)), (N, S)
        d_ref = ref_pn2.square_distance(c, p)
    print("[ball query] oracle C == reference torch on random clouds for all three level shapes")

    # ============================ VQ alone (spread latents) =====================================
    from puzzlefusion_plusplus.vqvae.model.modules.quantizer import VectorQuantizer
    vq = VectorQuantizer(1024, 16, 0.25)
    vq.embedding.weight.data.copy_(enc_sd["vector_quantization.embedding.weight"])
    z = torch.randn(6, 100, 16, generator=g)
    with torch.no_grad():
        _, zq_ref, _, _, codes_ref = vq(z)
    zq_o, codes_o = O.vector_quantize_c(vq.embedding.weight.data, z)
    assert torch.equal(codes_o, codes_ref.flatten()) and torch.equal(zq_o, zq_ref)
    print("[vq] oracle C argmin == reference on 600 random sub-vectors")
    np.savez_compressed(GOLD / "vq.npz", z=z.numpy(), z_q=zq_ref.numpy(), codes=codes_ref.flatten().numpy().astype(np.int16))

    # ============================ model_utils ====================================================
    pe_ref = PositionalEncoding(512).pe
    assert torch.equal(pe_ref, O.positional_table(512, 20))
    emb = EmbedderNerf(include_input=True, input_dims=3, max_freq_log2=9, num_freqs=10, log_sampling=True,
                       periodic_fns=[torch.sin, torch.cos])
    v = torch.randn(50, 3, generator=g)
    assert torch.equal(emb.embed(v), O.nerf_embed(v)) and emb.out_dim == 63

    # ============================ denoiser transformer ===========================================
    den_ref = DenoiserTransformer(cfg)
    den_sd = weights.denoiser_state_dict()
    den_ref.load_state_dict(den_sd, strict=True)
    den_ref.eval()
    B, P, L = 2, 20, 25
    valids = torch.zeros(B, P); valids[0, :8] = 1; valids[1, :20] = 1
    latent = torch.randn(B, P, L, 64, generator=g) * valids[:, :, None, None]
    xyz = (torch.rand(B, P, L, 3, generator=g) * 2 - 1) * valids[:, :, None, None]
    scale = torch.rand(B, P, 1, generator=g) * 0.9 + 0.05
    scale[valids == 0] = 1.0
    xx = torch.randn(B, P, 7, generator=g)
    ref_part = torch.zeros(B, P, dtype=torch.bool); ref_part[0, 2] = True; ref_part[1, 0] = True
    ts = torch.tensor([950, 500])
    with torch.no_grad():
        eps_ref = den_ref(xx, ts, latent, xyz, valids, scale, ref_part)
    ocap = {}
    eps_o = O.denoiser_forward(den_sd, xx, ts, latent, xyz, valids, scale, ref_part, capture=ocap)
    d = maxdiff(eps_o, eps_ref)
    print(f"[denoiser] oracle vs reference pred_noise maxdiff {d:.2e}")
    assert d < 1e-5
    np.savez_compressed(GOLD / "denoiser.npz", x=xx.numpy(), timesteps=ts.numpy(), latent=latent.numpy(), xyz=xyz.numpy(),
                        part_valids=valids.numpy(), scale=scale.numpy(), ref_part=ref_part.numpy(),
                        pred_noise=eps_ref.numpy(), tokens_sub=ocap["tokens"][:, ::25].numpy(),
                        layer_means=np.array([ocap[f"layer{i}"].double().abs().mean().item() for i in range(6)]))
    for t_single in (0, 999):
        with torch.no_grad():
            e1 = den_ref(xx, torch.tensor([t_single, t_single]), latent, xyz, valids, scale, ref_part)
        e2 = O.denoiser_forward(den_sd, xx, torch.tensor([t_single, t_single]), latent, xyz, valids, scale, ref_part)
        assert maxdiff(e1, e2) < 1e-5

    # ============================ a17: loss and gradients (reference autograd, eval-mode dropout) ==========
    noise_t = torch.randn(B, P, 7, generator=g)
    den_ref.zero_grad()
    pred_t = den_ref(xx, ts, latent, xyz, valids, scale, ref_part)
    sel = valids.bool().clone()
    sel[ref_part] = False
    loss_ref = torch.nn.functional.mse_loss(pred_t[sel], noise_t[sel])       # Denoiser._loss, denoiser.py:118-126
    loss_ref.backward()
    sd_req = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoding.pe") for k, v in den_sd.items()}
    loss_o = O.denoiser_loss(O.denoiser_forward(sd_req, xx, ts, latent, xyz, valids, scale, ref_part), noise_t, valids, ref_part)
    loss_o.backward()
    names = [n for n, _ in den_ref.named_parameters()]
    worst = 0.0
    for n, p_ in den_ref.named_parameters():
        worst = max(worst, maxdiff(p_.grad, sd_req[n].grad) / (p_.grad.abs().max().item() + 1e-30))
    print(f"[train] loss ref {loss_ref.item():.6f} oracle {loss_o.item():.6f}; worst relative grad diff oracle vs reference {worst:.2e}")
    assert abs(loss_ref.item() - loss_o.item()) < 1e-6 and worst < 1e-4
    np.savez_compressed(
        GOLD / "train.npz", noise=noise_t.numpy(), loss=np.float64(loss_ref.item()), names=np.array(names),
        grad_norm=np.array([p_.grad.double().norm().item() for _, p_ in den_ref.named_parameters()]),
        grad_absmax=np.array([p_.grad.abs().max().item() for _, p_ in den_ref.named_parameters()]),
        grad_sample=np.stack([p_.grad.flatten()[:: max(1, p_.numel() // 16)][:16].numpy() if p_.numel() >= 16 else
                              np.pad(p_.grad.flatten().numpy(), (0, 16 - p_.numel())) for _, p_ in den_ref.named_parameters()]))

    # ============================ 8f-3: evaluation metrics (the reference's evaluator.py on the stand-ins) ======
    from puzzlefusion_plusplus.denoiser.evaluation import evaluator as ref_eval

    class _CD:                                   # chamferdist.ChamferDistance call signature
        def __call__(self, a, b, **kw):
            return O.chamfer_distance(a, b, **kw)

    Bm, Pm, Nm = 2, 20, 200
    vm = torch.zeros(Bm, Pm); vm[0, :5] = 1; vm[1, :20] = 1
    pts_m = (torch.rand(Bm, Pm, Nm, 3, generator=g) - 0.5) * vm[:, :, None, None]
    t_gt = torch.randn(Bm, Pm, 3, generator=g) * 0.3
    q_gt = torch.nn.functional.normalize(torch.randn(Bm, Pm, 4, generator=g), dim=-1)
    t_pr = t_gt + torch.randn(Bm, Pm, 3, generator=g) * 0.05 * (torch.rand(Bm, Pm, 1, generator=g) < 0.5)
    q_pr = torch.nn.functional.normalize(q_gt + torch.randn(Bm, Pm, 4, generator=g) * 0.05 * (torch.rand(Bm, Pm, 1, generator=g) < 0.5), dim=-1)
    acc_r, accpp_r, cdpp_r = ref_eval.calc_part_acc(pts_m, t_pr, t_gt, q_pr, q_gt, vm, _CD())
    scd_r = ref_eval.calc_shape_cd(pts_m, t_pr, t_gt, q_pr, q_gt, vm, _CD())
    rr_r = ref_eval.rot_metrics(q_pr, q_gt, vm, "rmse")
    rt_r = ref_eval.trans_metrics(t_pr, t_gt, vm, "rmse")
    acc_o, accpp_o, cdpp_o = O.calc_part_acc(pts_m, t_pr, t_gt, q_pr, q_gt, vm)
    assert torch.equal(accpp_o, accpp_r) and maxdiff(cdpp_o, cdpp_r) < 1e-7 and maxdiff(acc_o, acc_r) == 0
    assert maxdiff(O.calc_shape_cd(pts_m, t_pr, t_gt, q_pr, q_gt, vm), scd_r) < 1e-6
    assert maxdiff(O.rot_metrics(q_pr, q_gt, vm), rr_r) < 1e-4 and maxdiff(O.trans_metrics(t_pr, t_gt, vm), rt_r) < 1e-7
    print(f"[metrics] oracle == reference evaluator.py (on stand-in pytorch3d/chamferdist): part_acc {acc_r.tolist()}, "
          f"shape_cd {scd_r.tolist()}, rmse_r {rr_r.tolist()}, rmse_t {rt_r.tolist()}")
    np.savez_compressed(GOLD / "metrics.npz", pts=pts_m.numpy(), valids=vm.numpy(), trans_gt=t_gt.numpy(), rot_gt=q_gt.numpy(),
                        trans_pred=t_pr.numpy(), rot_pred=q_pr.numpy(), part_acc=acc_r.numpy(), acc_per_part=accpp_r.numpy(),
                        cd_per_part=cdpp_r.numpy(), shape_cd=scd_r.numpy(), rmse_r=rr_r.numpy(), rmse_t=rt_r.numpy(),
                        euler_pred=ref_eval.quaternion_to_euler(q_pr).numpy())

    # ============================ 8f-2: merge step (the reference's node_merge_utils.py on the stand-ins) ========
    import utils.node_merge_utils as ref_merge

    def surf(n, c, r, gen):                         # points on a sphere cap-ish blob: two overlapping shells
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        return v * r + c
    mparts = torch.stack([surf(1000, torch.tensor([0.0, 0.0, 0.0]), 0.5, g), surf(1000, torch.tensor([0.3, 0.0, 0.0]), 0.5, g),
                          surf(1000, torch.tensor([0.0, 0.9, 0.0]), 0.4, g)])
    mparts[1, :300] = mparts[0, :300] + torch.randn(300, 3, generator=g) * 0.003        # a shared (fracture) surface
    merge_pcs = mparts.reshape(-1, 3)
    torch.manual_seed(77)
    ds_ref = ref_merge.remove_intersect_points_and_fps_ds(merge_pcs.clone(), _CD())
    start_used = sys.modules["torch_cluster"].fps.last_start
    ds_o = O.remove_intersect_points_and_fps_ds(merge_pcs, start=start_used)
    assert torch.equal(ds_o, ds_ref), "oracle merge != reference merge (on stand-ins)"
    nrm_o = O.estimate_normals(mparts, 20)
    print(f"[merge] oracle == reference remove_intersect_points_and_fps_ds on stand-ins; first FPS index {start_used}, "
          f"{ds_ref.shape[0]} points")
    np.savez_compressed(GOLD / "merge.npz", parts=mparts.numpy(), start=np.int64(start_used), merged=ds_ref.numpy(),
                        normals=nrm_o.numpy())

    # ============================ a19 / 8f-1: the reference's own glue of the auto-agglomerative loop ============
    # utils/node_merge_utils.py:16-53 (pose application), :62-89 (matched-point distances), :225-306 (init-pose bookkeeping,
    # composed poses) and auto_aggl.py:195-201,385-389 (bins, normalisation) run verbatim on the pytorch3d / chamferdist
    # stand-ins; what the fixture pins is everything the reference owns around those calls: pivot indexing, the order of the
    # affine products, normalise-or-not, index arithmetic of the matching data, bin edges, the count column.
    import networkx as nx
    import inspect
    import textwrap
    for name in ("lightning", "lightning.pytorch", "hydra"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.LightningModule = nn.Module
            sys.modules[name] = m
    sys.modules["lightning"].pytorch = sys.modules["lightning.pytorch"]
    import puzzlefusion_plusplus.auto_aggl as ref_aggl
    gg = torch.Generator().manual_seed(4242)
    Bg, Pg, Ng = 2, 5, 40
    pts_g = torch.rand(Bg, Pg, Ng, 3, generator=gg) - 0.5
    tr_g = torch.randn(Bg, Pg, 3, generator=gg) * 0.3
    ro_g = torch.randn(Bg, Pg, 4, generator=gg)                                  # not normalised on purpose
    fin_ref = ref_merge.get_final_pose_pts(pts_g, tr_g, ro_g)
    assert maxdiff(O.get_final_pose_pts(pts_g, tr_g, ro_g), fin_ref) == 0
    # by-area cloud of one puzzle: 5 parts, merged nodes share a pivot
    n_pcs = torch.tensor([[30, 50, 20, 60, 40]])
    pivots = [0, 0, 2, 3, 3]
    G = nx.Graph()
    for i, pv in enumerate(pivots):
        G.add_node(i, pivot=pv, valids=True, ref_part=False, init_pose=None, trans_and_rots=[])
    ptr_g = torch.randn(1, Pg, 3, generator=gg) * 0.2
    pro_g = torch.nn.functional.normalize(torch.randn(1, Pg, 4, generator=gg), dim=-1) * 1.3     # quaternion_apply does not normalise
    # world-frame points with neighbours at every distance scale the bins separate (squared distances 1e-4 .. 0.3), pulled back
    # through the inverse poses: x = conj(q) (w - t) q / |q|^4
    world = (torch.rand(int(n_pcs.sum()), 3, generator=gg) - 0.5) * 0.5
    wst = [0] + n_pcs[0].cumsum(0).tolist()
    for (dst_part, src_part) in ((1, 0), (3, 2), (4, 3)):
        m = min(int(n_pcs[0, dst_part]), int(n_pcs[0, src_part]))
        scale = torch.tensor([0.004, 0.02, 0.05, 0.12, 0.25])[torch.arange(m) % 5].unsqueeze(-1)
        world[wst[dst_part]: wst[dst_part] + m] = world[wst[src_part]: wst[src_part] + m] + torch.nn.functional.normalize(
            torch.randn(m, 3, generator=gg), dim=-1) * scale
    piv_of_pt = torch.cat([torch.full((int(n),), pivots[i]) for i, n in enumerate(n_pcs[0])])
    qv = pro_g[0][piv_of_pt].double()
    qinv = qv * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=torch.float64) / (qv * qv).sum(-1, keepdim=True)
    area = O.quaternion_apply(qinv, world.double() - ptr_g[0][piv_of_pt].double()).float().unsqueeze(0)
    dyn_ref = ref_merge.get_final_pose_pts_dynamic(area, n_pcs, ptr_g, pro_g, torch.tensor([Pg]), G.nodes)
    pose_idx = torch.cat([torch.full((int(n),), pivots[i], dtype=torch.int32) for i, n in enumerate(n_pcs[0])])
    dyn_o = O.pose_apply_points(area[0], pose_idx, torch.cat([ptr_g[0], pro_g[0]], -1))
    assert maxdiff(dyn_o, dyn_ref) == 0, "oracle pose_apply_points != reference get_final_pose_pts_dynamic"
    # matched points of three candidate edges
    n_crit = torch.tensor([[12, 20, 9, 25, 16]])
    starts = [0] + n_pcs[0].cumsum(0).tolist()[:-1]
    crit_idx = torch.zeros(1, int(n_pcs.sum()), dtype=torch.int64)
    for i in range(Pg):
        crit_idx[0, starts[i]: starts[i] + int(n_crit[0, i])] = torch.arange(int(n_crit[0, i]))      # the first points of a part are its critical ones
    edges_g = [(1, 0), (3, 2), (4, 3)]                                          # (idx2, idx1) as stored in matching_data
    corr_g = [torch.stack([torch.randint(0, int(n_crit[0, e[1]]), (m,), generator=gg),
                           torch.randint(0, int(n_crit[0, e[0]]), (m,), generator=gg)], -1) for e, m in zip(edges_g, (17, 6, 31))]
    cds, bins_ref = [], []
    fake_self = NS(device="cpu")
    for (i2, i1), cr in zip(edges_g, corr_g):
        cd = ref_merge.get_distance_for_matching_pts(i1, i2, dyn_ref, n_pcs, n_crit, crit_idx, [cr], None, _CD())
        cds.append(cd[0])
        bins_ref.append(ref_aggl.AutoAgglomerative._make_cd_to_bins(fake_self, cd))
    bins_ref = torch.stack(bins_ref)
    idx_a = torch.cat([starts[i1] + crit_idx[0, starts[i1]: starts[i1] + int(n_crit[0, i1])][cr[:, 0]] for (i2, i1), cr in zip(edges_g, corr_g)]).to(torch.int32)
    idx_b = torch.cat([starts[i2] + crit_idx[0, starts[i2]: starts[i2] + int(n_crit[0, i2])][cr[:, 1]] for (i2, i1), cr in zip(edges_g, corr_g)]).to(torch.int32)
    edge_off = torch.tensor([0] + torch.tensor([c.shape[0] for c in corr_g]).cumsum(0).tolist(), dtype=torch.int32)
    hist_o = O.edge_histogram(dyn_ref, idx_a, idx_b, edge_off)
    assert torch.equal(hist_o.long(), bins_ref.long()), "oracle edge_histogram != reference get_distance_for_matching_pts + _make_cd_to_bins"
    # flatten / normalise / count column: auto_aggl.py:195-201 executed from the reference's own source text
    src_lines = inspect.getsource(ref_aggl.AutoAgglomerative.test_step).splitlines()
    first = next(i for i, l in enumerate(src_lines) if "mat_mask = torch.triu" in l)
    last = next(i for i, l in enumerate(src_lines) if "edge_features = torch.cat((edge_features, num_points)" in l)
    block = textwrap.dedent("\n".join(src_lines[first:last + 1]))
    ef = torch.zeros(1, Pg, Pg, 6, dtype=torch.int32)
    for (i2, i1), b in zip(edges_g, bins_ref):
        ef[0, i1, i2] = b.to(torch.int32)
    ns = dict(torch=torch, P=Pg, B=1, edge_features=ef.clone(), num_parts=torch.tensor([Pg]),
              self=NS(device="cpu", _get_edge_mask=lambda num_parts, P: torch.ones(1, P * (P - 1) // 2, dtype=torch.bool)))
    exec(block, ns)
    ef_ref, eidx_ref = ns["edge_features"], ns["edge_indices"]
    ef_o, eidx_o = O.edge_features_from_hist(ef)
    assert maxdiff(ef_o, ef_ref) == 0 and torch.equal(eidx_o, eidx_ref)
    # three merges, then the composed poses of get_param / extract_final_pred_trans_rots
    nodes_ref = {i: dict(pivot=pv, init_pose=None) for i, pv in enumerate(pivots)}
    nodes_o = {i: dict(pivot=pv, init_pose=None) for i, pv in enumerate(pivots)}
    merges = [([0, 1], torch.tensor([0.05, -0.02, 0.01])), ([3, 4], torch.tensor([-0.1, 0.0, 0.03])), ([0, 1, 2], torch.tensor([0.02, 0.04, -0.06]))]
    merge_tr = [torch.randn(Pg, 3, generator=gg) * 0.2 for _ in merges]
    merge_ro = [torch.nn.functional.normalize(torch.randn(Pg, 4, generator=gg), dim=-1) for _ in merges]
    for (comp, cen), t_, r_ in zip(merges, merge_tr, merge_ro):
        ref_merge.assign_init_pose(nodes_ref, t_, r_, cen, comp)
        O.assign_init_pose(nodes_o, t_, r_, cen, comp)
    init_ref = torch.stack([nodes_ref[i]["init_pose"] if nodes_ref[i]["init_pose"] is not None else torch.zeros(4, 4) for i in range(Pg)])
    has_init = torch.tensor([nodes_ref[i]["init_pose"] is not None for i in range(Pg)])
    init_o = torch.stack([nodes_o[i]["init_pose"] if nodes_o[i]["init_pose"] is not None else torch.zeros(4, 4) for i in range(Pg)])
    assert maxdiff(init_o, init_ref) == 0, "oracle assign_init_pose != reference"

    class _Nodes:                                   # networkx NodeView call signature: nodes(data=True) -> (index, attributes)
        def __init__(self, d): self.d = d
        def __call__(self, data=True): return list(self.d.items())
    param = torch.cat([torch.randn(Pg, 3, generator=gg) * 0.2, torch.nn.functional.normalize(torch.randn(Pg, 4, generator=gg), dim=-1)], -1)
    gp_ref = ref_merge.get_param(param, _Nodes(nodes_ref))
    ft_ref, fr_ref = ref_merge.extract_final_pred_trans_rots(param[:, :3], param[:, 3:], _Nodes(nodes_ref))
    gp_o = O.pose_compose(param, pivots, init_ref.reshape(Pg, 16), has_init)
    assert maxdiff(gp_o, gp_ref) < 1e-6 and maxdiff(gp_o, torch.cat([ft_ref, fr_ref], -1)) < 1e-6, "oracle pose_compose != reference get_param"
    print(f"[aggl glue] oracle == reference: get_final_pose_pts(_dynamic), matched-point bins {bins_ref.tolist()}, edge-feature "
          f"normalisation, 3-merge init_pose chain, get_param / extract_final_pred_trans_rots (max diff {maxdiff(gp_o, gp_ref):.1e})")
    np.savez_compressed(
        GOLD / "aggl_glue.npz", pts=pts_g.numpy(), trans=tr_g.numpy(), rots=ro_g.numpy(), final_pts=fin_ref.numpy(),
        area=area[0].numpy(), n_pcs=n_pcs.numpy(), pivots=np.array(pivots, dtype=np.int32), dyn_trans=ptr_g[0].numpy(), dyn_rots=pro_g[0].numpy(),
        dyn_pts=dyn_ref.numpy(), pose_idx=pose_idx.numpy(), idx_a=idx_a.numpy(), idx_b=idx_b.numpy(), edge_off=edge_off.numpy(),
        cd_per_point=torch.cat(cds).numpy(), bins=bins_ref.numpy().astype(np.int32), hist_pp=ef.numpy(), edge_features=ef_ref.numpy(),
        edge_indices=eidx_ref.numpy(), merge_components=np.array([c + [-1] * (3 - len(c)) for c, _ in merges], dtype=np.int32),
        merge_centroids=torch.stack([c for _, c in merges]).numpy(), merge_trans=torch.stack(merge_tr).numpy(),
        merge_rots=torch.stack(merge_ro).numpy(), init_pose=init_ref.numpy(), has_init=has_init.numpy(), param=param.numpy(),
        composed=gp_ref.numpy(), final_trans=ft_ref.numpy(), final_rots=fr_ref.numpy())

    # ============================ 8f-4: dataset classes on the reference's on-disk formats ======================
    import subprocess, tempfile
    from puzzlefusion_plusplus.denoiser.dataset.dataset import GeometryLatentDataset as RefDS
    from puzzlefusion_plusplus.verifier.dataset.dataset import VerifierDataset as RefVDS
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_synthetic_dataset.py"), td, "--n", "4", "--points", "300"], check=True)
        dcfg = NS(data=NS(max_num_part=20, matching_data_path=td + "/matching_data"), model=NS(multiple_ref_parts=False))
        ref_ds = RefDS(dcfg, td + "/pc_data/train", -1, "test")
        spec2 = importlib.util.spec_from_file_location("pfpp_ds", ROOT / "puzzlefusion-plusplus_amd/puzzlefusion_plusplus/denoiser/dataset/dataset.py")
        # the drop-in imports pfpp_hip.*: make the product package importable for this one module only
        sys.path.append(str(ROOT / "puzzlefusion-plusplus_amd"))
        mine_mod = importlib.util.module_from_spec(spec2); spec2.loader.exec_module(mine_mod)
        sys.path.remove(str(ROOT / "puzzlefusion-plusplus_amd"))
        my_ds = mine_mod.GeometryLatentDataset(dcfg, td + "/pc_data/train", -1, "test")
        worst = 0.0
        for i in range(len(ref_ds)):
            np.random.seed(100 + i); a = ref_ds[i]
            np.random.seed(100 + i); b = my_ds[i]
            assert set(a.keys()) == set(b.keys()), set(a.keys()) ^ set(b.keys())
            for k in a:
                if isinstance(a[k], np.ndarray) and a[k].dtype != object:
                    worst = max(worst, float(np.abs(a[k].astype(np.float64) - np.asarray(b[k]).astype(np.float64)).max()))
                elif k == "correspondences":
                    assert all(np.array_equal(x, y) for x, y in zip(a[k], b[k]))
                else:
                    assert a[k] == b[k], k
        spec3 = importlib.util.spec_from_file_location("pfpp_vds", ROOT / "puzzlefusion-plusplus_amd/puzzlefusion_plusplus/verifier/dataset/dataset.py")
        vmod = importlib.util.module_from_spec(spec3); spec3.loader.exec_module(vmod)
        rv, mv = RefVDS(td + "/verifier_data", -1, "train"), vmod.VerifierDataset(td + "/verifier_data", -1, "train")
        for i in range(len(rv)):
            for k, val in rv[i].items():
                assert np.array_equal(np.asarray(val), np.asarray(mv[i][k])), k
        print(f"[datasets] drop-in GeometryLatentDataset / VerifierDataset == the reference's on the same files and numpy seed "
              f"({len(ref_ds)} puzzles, test mode): max abs diff {worst:.2e}")
        assert worst < 1e-5
    # fixture for the CPU test (tests/test_abi_and_host.py): what the REFERENCE's dataset classes return on the files
    # `tools/make_synthetic_dataset.py <dir> --n 3 --points 64` writes (deterministic generator), numpy seed 100 + i per item
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_synthetic_dataset.py"), td, "--n", "3", "--points", "64"], check=True)
        dcfg = NS(data=NS(max_num_part=20, matching_data_path=td + "/matching_data"), model=NS(multiple_ref_parts=False))
        fx = {}
        # the rotations __getitem__ draws (dataset.py:126, 140: scipy Rotation.random(), first the whole-assembly one, then one per
        # fragment) are recorded in float64 as the quaternions the reference derives from them (dataset.py:128-130, 142-144) —
        # the pin of oracle.fragment_prepare and, through it, of the GPU augmentation kernel
        import puzzlefusion_plusplus.denoiser.dataset.dataset as ref_ds_mod
        from scipy.spatial.transform import Rotation as SciR
        drawn = []

        class RecordingR:
            from_quat = staticmethod(SciR.from_quat)
            from_matrix = staticmethod(SciR.from_matrix)

            @staticmethod
            def random(*a, **k):
                r = SciR.random(*a, **k)
                drawn.append(SciR.from_matrix(r.as_matrix().T).as_quat()[[3, 0, 1, 2]])
                return r

        ref_ds_mod.R = RecordingR
        for mode in ("test", "train"):
            ref_ds = RefDS(dcfg, td + "/pc_data/train", -1, mode)
            fx[f"len_{mode}"] = np.int64(len(ref_ds))
            for i in range(len(ref_ds)):
                np.random.seed(100 + i)
                drawn.clear()
                item = ref_ds[i]
                assert len(drawn) == 1 + int(item["num_parts"])
                fx[f"{mode}{i}_drawn_quats_f64"] = np.stack(drawn)          # [1 + num_parts, 4], scalar first
                P_ = item["part_pcs_gt"].shape[0]
                qp = np.zeros((1, P_, 4)); qp[0, :, 0] = 1.0; qp[0, :len(drawn) - 1] = np.stack(drawn[1:])
                got = O.fragment_prepare(item["part_pcs_gt"][None], np.array([item["num_parts"]]), np.array([int(np.argmax(item["ref_part"]))]),
                                         np.stack(drawn[:1]), qp)
                pv_ = int(item["num_parts"])
                for name, g_, w_ in (("part_pcs", got[0][0], item["part_pcs"]), ("part_trans", got[1][0], item["part_trans"]),
                                   ("part_scale", got[2][0][:pv_], item["part_scale"][:pv_]), ("init_pose_t", got[3][0], item["init_pose_t"])):
                    dd = float(np.abs(np.asarray(g_, np.float64) - np.asarray(w_, np.float64)).max())
                    assert dd < 5e-7, (mode, i, name, dd)
                assert np.abs(item["part_rots"][:pv_] - np.stack(drawn[1:]).astype(np.float32)).max() == 0
                assert np.abs(item["init_pose_r"] - drawn[0]).max() == 0
                for k, v in item.items():
                    if k == "correspondences":
                        fx[f"{mode}{i}_corr_cat"] = np.concatenate([np.asarray(c).reshape(-1, 2) for c in v]) if len(v) else np.zeros((0, 2), np.int64)
                        fx[f"{mode}{i}_corr_len"] = np.array([len(c) for c in v], dtype=np.int64)
                    elif isinstance(v, np.ndarray) and v.dtype != object:
                        fx[f"{mode}{i}_{k}"] = v
                    elif isinstance(v, (int, float, np.integer, np.floating, bool, np.bool_)):
                        fx[f"{mode}{i}_{k}"] = np.asarray(v)
        rv = RefVDS(td + "/verifier_data", -1, "train")
        fx["len_verifier"] = np.int64(len(rv))
        for i in range(len(rv)):
            for k, val in rv[i].items():
                fx[f"v{i}_{k}"] = np.asarray(val)
        ref_ds_mod.R = SciR
        print("[datasets] oracle.fragment_prepare == the reference's __getitem__ augmentation (dataset.py:163-222) on the recorded rotations (< 5e-7)")
        np.savez_compressed(GOLD / "dataset.npz", **fx)
        print(f"[datasets] fixture: {len(fx)} arrays from the reference's GeometryLatentDataset (test + train mode) and VerifierDataset")

    # ============================ scheduler =====================================================
    sch_ref = PiecewiseScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                                 beta_start=1e-4, beta_end=2e-2, clip_sample=False, timestep_spacing="leading")
    sch_ref.set_timesteps(20)
    so = O.PiecewiseSchedule()
    so.set_timesteps(20)
    assert torch.equal(sch_ref.alphas_cumprod, so.alphas_cumprod) and torch.equal(sch_ref.timesteps, so.timesteps)
    assert torch.equal(betas_for_alpha_bar(alpha_transform_type="piece_wise"), so.betas)
    noise = torch.randn(B, P, 7, generator=g)
    steps = []
    for t in sch_ref.timesteps:
        a = sch_ref.step(eps_ref, t, xx, variance_noise=noise).prev_sample
        b = so.step(eps_ref, int(t), xx, noise)
        assert torch.equal(a, b), int(t)
        steps.append(a.numpy())
    tt = torch.tensor([3, 700])
    an = sch_ref.add_noise(xx, noise, tt)
    assert torch.equal(an, so.add_noise(xx, noise, tt))
    np.savez_compressed(GOLD / "scheduler.npz", alphas_cumprod=sch_ref.alphas_cumprod.numpy(),
                        timesteps=sch_ref.timesteps.numpy(), x=xx.numpy(), eps=eps_ref.numpy(), noise=noise.numpy(),
                        step_out=np.stack(steps), add_noise_t=tt.numpy(), add_noise_out=an.numpy())
    print("[scheduler] oracle == reference PiecewiseScheduler (tables, 20 steps, add_noise) bit-exact; abar at the "
          "20 timesteps:", [round(float(sch_ref.alphas_cumprod[t]), 6) for t in sch_ref.timesteps][:4], "...")

    # ============================ verifier ======================================================
    vcfg = NS(model=NS(embed_dim=256, num_layers=6, num_heads=8))
    ver_ref = VerifierTransformer(vcfg)
    ver_sd = weights.verifier_state_dict()
    ver_ref.load_state_dict(ver_sd, strict=True)
    ver_ref.eval()
    ed = synthetic.make_edges(2, seed=3)
    with torch.no_grad():
        lo_ref = ver_ref(ed["edge_features"], ed["edge_indices"], ed["edge_valids"])
    lo_o = O.verifier_forward(ver_sd, ed["edge_features"], ed["edge_indices"], ed["edge_valids"])
    m = ed["edge_valids"].bool()
    d = (lo_o - lo_ref).abs()[m].max().item()
    print(f"[verifier] oracle vs reference logits (valid edges) maxdiff {d:.2e}")
    assert d < 1e-5
    np.savez_compressed(GOLD / "verifier.npz", edge_features=ed["edge_features"].numpy(),
                        edge_indices=ed["edge_indices"].numpy().astype(np.int16), edge_valids=ed["edge_valids"].numpy(),
                        logits=lo_ref.numpy())
    total = sum(f.stat().st_size for f in GOLD.glob("*.npz"))
    print(f"wrote {len(list(GOLD.glob('*.npz')))} fixtures, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
