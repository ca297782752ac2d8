"""Bring-up diagnostics: run every HIP kernel against the CPU oracle and print the differences.
(developer tool; the pass/fail versions of these checks live in tests/ under -m gpu)

    python tools/gpu_check.py [--quick]
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))

import numpy as np
import torch

from oracle import pfpp_oracle as O
from oracle import weights
from pfpp_hip import denoiser as D
from pfpp_hip import encoder as E
from pfpp_hip import ops, synthetic
from pfpp_hip import verifier as V
from pfpp_hip.scheduler import PiecewiseScheduler

dev = torch.device("cuda:0")
FAILS = []


def report(name, ok, msg=""):
    print(f"[{'ok' if ok else 'FAIL'}] {name} {msg}", flush=True)
    if not ok:
        FAILS.append(name)


def maxdiff(a, b):
    return (a.detach().cpu().float() - b.detach().cpu().float()).abs().max().item()


def to_dev(sd):
    return {k: v.to(dev) for k, v in sd.items()}


def main():
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    print("device:", torch.cuda.get_device_name(0), flush=True)

    # ---------------------------------------------------------------- GEMM variants
    g = torch.Generator().manual_seed(1)
    for (M, N, K, act, pool, scale) in [
        (1000, 192, 132, "relu", 0, True), (4096, 128, 64, "relu", 64, True), (2048, 64, 4, "relu", 32, True),
        (777, 512, 148, "none", 0, False), (500, 1536, 512, "none", 0, False), (640, 3, 256, "none", 0, False),
        (333, 256, 2048, "gelu", 0, False), (256, 512, 512, "silu", 0, False), (8192, 256, 260, "relu", 64, True),
    ]:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        Kp = (K + 3) // 4 * 4
        Ap = torch.zeros(M, Kp); Ap[:, :K] = A
        Wp = torch.zeros(N, Kp); Wp[:, :K] = W
        b = torch.randn(N, generator=g)
        s = torch.rand(N, generator=g) + 0.5
        ref = A.double() @ W.double().t()
        ref = ref * s.double() + b.double() if scale else ref + b.double()
        ref = {"relu": torch.relu, "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu,
               "none": lambda v: v}[act](ref)
        if pool:
            ref = ref.view(M // pool, pool, N).max(1)[0]
        out = ops.linear(Ap.to(dev), Wp.to(dev), None if scale else b.to(dev), act=act, pool=pool, K=K,
                         scale=s.to(dev) if scale else None, shift=b.to(dev) if scale else None)
        d = maxdiff(out, ref.float())
        report(f"gemm M{M} N{N} K{K} {act} pool{pool}", d < 2e-4, f"maxdiff {d:.2e}")
    # residual + geglu
    M, C = 700, 512
    A = torch.randn(M, C, generator=g); W = torch.randn(4096, C, generator=g) / C ** 0.5; b = torch.randn(4096, generator=g)
    from pfpp_hip.packing import pack_geglu
    wp, bp = pack_geglu(W, b)
    out = ops.linear(A.to(dev), wp.to(dev), bp.to(dev), act="geglu")
    hh, gg = (A.double() @ W.double().t() + b.double()).chunk(2, -1)
    d = maxdiff(out, (hh * torch.nn.functional.gelu(gg)).float())
    report("gemm geglu", d < 2e-4 and out.shape == (M, 2048), f"maxdiff {d:.2e}")
    R = torch.randn(M, C, generator=g); W2 = torch.randn(C, 2048, generator=g) / 45.0; b2 = torch.randn(C, generator=g)
    U = torch.randn(M, 2048, generator=g)
    h = R.clone().to(dev)
    ops.gemm(U.to(dev), W2.to(dev), M=M, N=C, K=2048, lda=2048, ldw=2048, out=h, ldc=C, bias=b2.to(dev), residual=h, ldr=C)
    d = maxdiff(h, (U.double() @ W2.double().t() + b2.double() + R.double()).float())
    report("gemm residual in-place", d < 2e-4, f"maxdiff {d:.2e}")

    # ---------------------------------------------------------------- point ops
    batch = synthetic.make_batch(0, 4 if quick else 8, num_points=1000)
    B, P, N, _ = batch["part_pcs"].shape
    x = torch.randn(B, P, 7)
    valid = batch["part_valids"].bool()
    slot = torch.nonzero(valid.flatten()).flatten().to(torch.int32)
    rot_ref = O.apply_rots(batch["part_pcs"], x)[valid]
    rot = ops.se3_rotate_gather(batch["part_pcs"].view(B * P, N, 3).to(dev), x.view(B * P, 7).to(dev), slot.to(dev))
    rot_ref_c = O.apply_rots_c(batch["part_pcs"], x)[valid]
    print(f"      (oracle torch vs oracle C on this host: {(rot_ref != rot_ref_c).sum().item()} differing values)")
    report("se3_rotate_gather bit-exact vs oracle C", torch.equal(rot.cpu(), rot_ref_c), f"maxdiff {maxdiff(rot, rot_ref_c):.2e}")
    report("se3_rotate_gather bit-exact", torch.equal(rot.cpu(), rot_ref), f"maxdiff {maxdiff(rot, rot_ref):.2e} F={slot.numel()}")

    for Npts, S in ((1000, 256), (1024, 256), (512, 256), (256, 128), (128, 25), (2048, 256), (300, 77)):
        pts = rot_ref[:, :Npts].contiguous() if Npts <= 1000 else torch.rand(5, Npts, 3) * 2 - 1
        ref_idx = O.fps(pts, S)
        t0 = time.time()
        idx, nx = ops.fps(pts.to(dev), S)
        torch.cuda.synchronize()
        same = torch.equal(idx.cpu().long(), ref_idx)
        nx_ok = torch.equal(nx.cpu(), O.index_points(pts, ref_idx))
        report(f"fps N{Npts} S{S}", same and nx_ok, f"mismatch {(idx.cpu().long() != ref_idx).sum().item()} ({time.time() - t0:.3f}s)")

    for (Npts, S, r, ns) in ((1000, 256, 0.2, 32), (256, 128, 0.4, 64), (128, 25, 0.8, 64), (1024, 256, 0.2, 32)):
        pts = rot_ref[:, :Npts].contiguous() if Npts <= 1000 else torch.rand(5, Npts, 3) * 2 - 1
        fi = O.fps(pts, S)
        nx = O.index_points(pts, fi)
        ref = O.query_ball_point(r, ns, pts, nx)
        got = ops.ball_query(pts.to(dev), nx.to(dev), r, ns)
        report(f"ball_query N{Npts} S{S} r{r}", torch.equal(got.cpu().long(), ref), f"mismatch {(got.cpu().long() != ref).sum().item()}")
        ref_t = O.query_ball_point_torch(r, ns, pts, nx)
        print(f"      (oracle C vs torch-matmul restatement on this host: {(ref_t != ref).sum().item()} differing entries)")

    # ---------------------------------------------------------------- encoder end to end
    enc_sd = weights.vqvae_state_dict()
    pk = E.pack_encoder(to_dev(enc_sd))
    cap_ref, cap = {}, {}
    out_ref = O.vqvae_encode(enc_sd, rot_ref, capture=cap_ref)
    lat, xyz = E.extract_features(pk, batch["part_pcs"].to(dev), x.to(dev), slot.to(dev), capture=cap)
    torch.cuda.synchronize()
    for lvl in ("sa1", "sa2", "sa3"):
        report(f"encoder {lvl} fps idx", torch.equal(cap[f"{lvl}.fps_idx"].cpu().long(), cap_ref[f"pn2.{lvl}.fps_idx"]))
        report(f"encoder {lvl} ball idx", torch.equal(cap[f"{lvl}.ball_idx"].cpu().long(), cap_ref[f"pn2.{lvl}.ball_idx"]))
        d = maxdiff(cap[f"{lvl}.new_points"], cap_ref[f"pn2.{lvl}.new_points"])
        report(f"encoder {lvl} features", d < 1e-4, f"maxdiff {d:.2e}")
    d = maxdiff(cap["z_e"], cap_ref["z_e"])
    report("encoder z_e", d < 1e-4, f"maxdiff {d:.2e}")
    lat_ref = torch.zeros(B * P, 25, 64); lat_ref[slot.long()] = out_ref["z_q"]
    xyz_ref = torch.zeros(B * P, 25, 3); xyz_ref[slot.long()] = out_ref["xyz"]
    dl = (lat.cpu().view(B * P, 25, 64) - lat_ref).abs()
    bad = (dl.view(B * P, 100, 16).amax(-1) > 1e-4).sum().item()
    report("encoder z_q (scattered)", bad <= 2, f"sub-vectors differing {bad} / {slot.numel() * 100}, maxdiff {dl.max():.2e}")
    report("encoder xyz (scattered)", torch.equal(xyz.cpu().view(B * P, 25, 3), xyz_ref))

    # VQ kernel alone on spread-out latents
    z = torch.randn(50, 25, 64)
    cb = enc_sd["vector_quantization.embedding.weight"]
    zq_ref, codes_ref = O.vector_quantize_c(cb, z.reshape(50, 100, 16))
    zq, codes = ops.vq_encode(z.to(dev), cb.to(dev), torch.arange(50, dtype=torch.int32, device=dev), 50, return_codes=True)
    report("vq codes bit-exact", torch.equal(codes.cpu().long().flatten(), codes_ref), f"mismatch {(codes.cpu().long().flatten() != codes_ref).sum().item()}")
    report("vq z_q bit-exact", torch.equal(zq.cpu().view(50, 100, 16), zq_ref))

    # ---------------------------------------------------------------- denoiser
    den_sd = weights.denoiser_state_dict()
    pkd = D.pack_denoiser(to_dev(den_sd), 6)
    ts = torch.tensor([950, 500, 0, 123, 999, 7, 650, 300][:B])
    lat_c, xyz_c = lat_ref.view(B, P, 25, 64), xyz_ref.view(B, P, 25, 3)
    cref, cgot = {}, {}
    t0 = time.time()
    eps_ref = O.denoiser_forward(den_sd, x, ts, lat_c, xyz_c, batch["part_valids"], batch["part_scale"], batch["ref_part"], capture=cref)
    t_cpu = time.time() - t0
    eps = D.denoiser_forward(pkd, x.to(dev), ts.to(dev), lat_c.to(dev), xyz_c.to(dev), batch["part_valids"].to(dev),
                             batch["part_scale"].to(dev), batch["ref_part"].to(dev), num_layers=6, num_heads=8, capture=cgot)
    torch.cuda.synchronize()
    d = maxdiff(cgot["tokens"].view(B, -1, 512), cref["tokens"])
    report("denoiser tokens", d < 1e-4, f"maxdiff {d:.2e}")
    vm = batch["part_valids"].bool().repeat_interleave(25, dim=1)
    for i in range(6):
        dd = (cgot[f"layer{i}"].cpu().view(B, -1, 512) - cref[f"layer{i}"]).abs()
        report(f"denoiser layer{i}", dd.max().item() < 2e-4 * (i + 2), f"maxdiff {dd.max():.2e} (valid tokens {dd[vm].max():.2e})")
    d = maxdiff(eps, eps_ref)
    report("denoiser pred_noise", d < 1e-4, f"maxdiff {d:.2e} |eps| mean {eps_ref.abs().mean():.3f} (cpu oracle {t_cpu:.2f}s)")

    # ---------------------------------------------------------------- scheduler
    so, sg = O.PiecewiseSchedule(), PiecewiseScheduler()
    so.set_timesteps(20); sg.set_timesteps(20)
    report("scheduler tables", torch.equal(so.alphas_cumprod, sg.alphas_cumprod) and torch.equal(so.timesteps, sg.timesteps))
    noise = torch.randn(B, P, 7)
    worst = 0.0
    for t in so.timesteps.tolist():
        a = so.step(eps_ref, t, x, noise)
        bgot = sg.step(eps_ref.to(dev), t, x.to(dev), variance_noise=noise.to(dev)).prev_sample
        worst = max(worst, maxdiff(bgot, a))
    report("ddpm step (20 timesteps)", worst < 1e-5, f"maxdiff {worst:.2e}")
    tt = torch.randint(0, 1000, (B,))
    d = maxdiff(sg.add_noise(x.to(dev), noise.to(dev), tt.to(dev)), so.add_noise(x, noise, tt))
    report("add_noise", d < 1e-6, f"maxdiff {d:.2e}")

    # ---------------------------------------------------------------- verifier
    ver_sd = weights.verifier_state_dict()
    pkv = V.pack_verifier(to_dev(ver_sd), 6)
    ed = synthetic.make_edges(B)
    lo_ref = O.verifier_forward(ver_sd, ed["edge_features"], ed["edge_indices"], ed["edge_valids"])
    lo = V.verifier_forward(pkv, ed["edge_features"].to(dev), ed["edge_indices"].to(dev), ed["edge_valids"].to(dev),
                            num_layers=6, num_heads=8)
    m = ed["edge_valids"].bool()
    d = (lo.cpu() - lo_ref).abs()[m].max().item()
    report("verifier logits (valid edges)", d < 1e-4, f"maxdiff {d:.2e} |logit| mean {lo_ref[m].abs().mean():.3f}")

    # ---------------------------------------------------------------- pose ops
    pose = torch.randn(P, 7)
    pts = torch.randn(P, 100, 3)
    a = O.get_final_pose_pts(pts.unsqueeze(0), pose[None, :, :3], pose[None, :, 3:])[0]
    bgot = ops.pose_apply(pts.to(dev), pose.to(dev))
    report("pose_apply bit-exact", torch.equal(bgot.cpu(), a), f"maxdiff {maxdiff(bgot, a):.2e}")
    pivot = torch.randint(0, P, (P,), dtype=torch.int32)
    init = torch.eye(4).repeat(P, 1, 1)
    init[:, :3, :3] = O.quaternion_to_matrix(torch.nn.functional.normalize(torch.randn(P, 4), dim=-1))
    init[:, :3, 3] = torch.randn(P, 3)
    has = (torch.rand(P) < 0.5).to(torch.uint8)
    a = O.pose_compose(pose, pivot.tolist(), init.view(P, 16), has.tolist())
    bgot = ops.pose_compose(pose.to(dev), pivot.to(dev), init.view(P, 16).contiguous().to(dev), has.to(dev))
    d = maxdiff(bgot, a)
    report("pose_compose", d < 1e-5, f"maxdiff {d:.2e}")

    print("\nFAILED:" if FAILS else "\nALL OK", FAILS)
    return 1 if FAILS else 0


if __name__ == "__main__":
    sys.exit(main())
