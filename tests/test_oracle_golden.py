"""CPU: the oracle must reproduce the golden vectors produced by the reference's own modules
(tools/make_goldens.py).  Indices bit-exact; floating point within 1e-6."""
import numpy as np
import torch

from oracle import pfpp_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_rotate_matches_golden_and_c(golden, oracle_lib):
    g = golden("rotate")
    pcs, pose = T(g["part_pcs"]), T(g["pose"])
    rot = O.apply_rots(pcs.unsqueeze(0), pose.unsqueeze(0))[0]
    assert torch.equal(rot, T(g["rotated"]))
    assert torch.equal(O.apply_rots_c(pcs.unsqueeze(0), pose.unsqueeze(0))[0], rot)  # C restatement == torch ops


def test_encoder_stages_match_golden(golden, weights_sd, oracle_lib):
    sd = weights_sd("vqvae")
    for tag in ("float", "grid", "ref"):      # "ref": the reference shape, F = 4 fragments x N = 1000 points
        g = golden(f"encoder_{tag}")
        cap = {}
        out = O.vqvae_encode(sd, T(g["pts"]), capture=cap)
        for lvl in ("sa1", "sa2", "sa3"):
            assert np.array_equal(cap[f"pn2.{lvl}.fps_idx"].numpy(), g[f"{lvl}_fps_idx"].astype(np.int64)), (tag, lvl)
            assert np.array_equal(cap[f"pn2.{lvl}.ball_idx"].numpy(), g[f"{lvl}_ball_idx"].astype(np.int64)), (tag, lvl)
        assert np.abs(cap["pn2.sa1.new_points"][:, ::16].numpy() - g["sa1_feat_sub"]).max() < 1e-6
        assert np.abs(cap["pn2.sa2.new_points"][:, ::8].numpy() - g["sa2_feat_sub"]).max() < 1e-6
        assert np.abs(cap["pn2.sa3.new_points"].numpy() - g["sa3_feat"]).max() < 1e-6
        assert np.abs(cap["z_e"].numpy() - g["z_e"]).max() < 1e-6
        assert np.array_equal(out["xyz"].numpy(), g["xyz"])
        assert np.abs(out["z_q"].numpy() - g["z_q"]).max() < 1e-6


def test_ball_query_c_equals_torch_restatement(oracle_lib):
    gen = torch.Generator().manual_seed(11)
    for N, S, r, ns in ((1000, 256, 0.2, 32), (256, 128, 0.4, 64), (128, 25, 0.8, 64), (70, 9, 0.3, 5)):
        p = torch.rand(2, N, 3, generator=gen) * 2 - 1
        c = p[:, :S].contiguous()
        assert torch.equal(O.query_ball_point(r, ns, p, c), O.query_ball_point_torch(r, ns, p, c))


def test_ball_query_edge_cases(oracle_lib):
    # every point identical: all in range, first nsample indices; and a centroid with a single neighbour
    p = torch.zeros(1, 40, 3)
    idx = O.query_ball_point(0.2, 8, p, p[:, :3].contiguous())
    assert torch.equal(idx[0, 0], torch.arange(8))
    p2 = torch.zeros(1, 10, 3); p2[0, :, 0] = torch.arange(10).float()
    idx2 = O.query_ball_point(0.2, 4, p2, p2[:, 5:6].contiguous())
    assert idx2[0, 0].tolist() == [5, 5, 5, 5]          # padded with the first hit


def test_fps_properties(oracle_lib):
    gen = torch.Generator().manual_seed(3)
    p = torch.rand(3, 300, 3, generator=gen)
    idx = O.fps(p, 77)
    assert idx[:, 0].eq(0).all()
    for f in range(3):
        assert len(set(idx[f].tolist())) == 77
    # duplicate points: ties resolve to the lowest index
    q = torch.zeros(1, 64, 3); q[0, 32:] = 1.0
    i2 = O.fps(q, 2)
    assert i2[0].tolist() == [0, 32]


def test_vq_golden(golden, weights_sd, oracle_lib):
    g = golden("vq")
    cb = weights_sd("vqvae")["vector_quantization.embedding.weight"]
    zq, codes = O.vector_quantize_c(cb, T(g["z"]))
    assert np.array_equal(codes.numpy(), g["codes"].astype(np.int64))
    assert np.array_equal(zq.numpy(), g["z_q"])
    zq2, codes2 = O.vector_quantize(cb, T(g["z"]))
    assert torch.equal(codes2, codes)


def test_denoiser_golden(golden, weights_sd):
    g = golden("denoiser")
    cap = {}
    eps = O.denoiser_forward(weights_sd("denoiser"), T(g["x"]), T(g["timesteps"]), T(g["latent"]), T(g["xyz"]),
                             T(g["part_valids"]), T(g["scale"]), T(g["ref_part"]), capture=cap)
    assert np.abs(eps.numpy() - g["pred_noise"]).max() < 1e-5
    assert np.abs(cap["tokens"][:, ::25].numpy() - g["tokens_sub"]).max() < 1e-6


def test_scheduler_golden(golden):
    g = golden("scheduler")
    s = O.PiecewiseSchedule()
    s.set_timesteps(20)
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert np.array_equal(s.timesteps.numpy(), g["timesteps"])
    # known answers of SURVEY.md §8a row a16
    known = [.26999, .49599, .67199, .79799, .87399, .89999, .91351, .926286, .938041, .948776, .95849, .967184,
             .974857, .98151, .987143, .991755, .995347, .997918, .999469, 1.0]
    got = [float(s.alphas_cumprod[t]) for t in s.timesteps]
    assert np.allclose(got, known, atol=2e-5)
    x, eps, noise = T(g["x"]), T(g["eps"]), T(g["noise"])
    for i, t in enumerate(s.timesteps.tolist()):
        assert np.array_equal(s.step(eps, t, x, noise).numpy(), g["step_out"][i]), t
    assert np.array_equal(s.add_noise(x, noise, T(g["add_noise_t"])).numpy(), g["add_noise_out"])


def test_verifier_golden(golden, weights_sd):
    g = golden("verifier")
    lo = O.verifier_forward(weights_sd("verifier"), T(g["edge_features"]), T(g["edge_indices"].astype(np.int64)),
                            T(g["edge_valids"]))
    m = g["edge_valids"].astype(bool)
    assert np.abs(lo.numpy() - g["logits"])[m].max() < 1e-5


def test_quaternion_roundtrip():
    q = torch.nn.functional.normalize(torch.randn(100, 4), dim=-1)
    q = torch.where(q[:, :1] < 0, -q, q)
    m = O.quaternion_to_matrix(q)
    assert torch.allclose(m @ m.transpose(-1, -2), torch.eye(3).expand(100, 3, 3), atol=1e-5)
    assert torch.allclose(O.matrix_to_quaternion(m), q, atol=1e-5)
    p = torch.randn(100, 3)
    assert torch.allclose(O.quaternion_apply(q, p), (m @ p[..., None])[..., 0], atol=1e-5)


def test_aggl_glue_oracle_vs_reference_golden(golden):
    """oracle restatements of the auto-agglomerative glue == the reference's own functions (fixture written by
    tools/make_goldens.py from utils/node_merge_utils.py:16-53,62-89,225-306 and auto_aggl.py:195-201,385-389)"""
    from oracle import pfpp_oracle as O

    g = golden("aggl_glue")
    T = torch.from_numpy
    assert torch.equal(O.get_final_pose_pts(T(g["pts"]), T(g["trans"]), T(g["rots"])), T(g["final_pts"]))
    pose = torch.cat([T(g["dyn_trans"]), T(g["dyn_rots"])], -1)
    dyn = O.pose_apply_points(T(g["area"]), T(g["pose_idx"]), pose)
    assert torch.equal(dyn, T(g["dyn_pts"]))
    hist = O.edge_histogram(dyn, T(g["idx_a"]), T(g["idx_b"]), T(g["edge_off"]))
    assert np.array_equal(hist.numpy(), g["bins"])
    ef, eidx = O.edge_features_from_hist(T(g["hist_pp"]))
    assert np.array_equal(ef.numpy(), g["edge_features"]) and np.array_equal(eidx.numpy(), g["edge_indices"])
    P = g["pivots"].shape[0]
    nodes = {i: dict(pivot=int(g["pivots"][i]), init_pose=None) for i in range(P)}
    for comp, cen, tr, ro in zip(g["merge_components"], g["merge_centroids"], g["merge_trans"], g["merge_rots"]):
        O.assign_init_pose(nodes, T(tr), T(ro), T(cen), [int(c) for c in comp if c >= 0])
    init = torch.stack([nodes[i]["init_pose"] if nodes[i]["init_pose"] is not None else torch.zeros(4, 4) for i in range(P)])
    assert np.abs(init.numpy() - g["init_pose"]).max() == 0
    comp = O.pose_compose(T(g["param"]), g["pivots"].tolist(), init.reshape(P, 16), T(g["has_init"]))
    assert np.abs(comp.numpy() - g["composed"]).max() < 1e-6


def _augmentation_cases(g):
    """(inputs, recorded rotations, reference outputs) per item of the dataset fixture (tests/golden/dataset.npz: what the
    reference's GeometryLatentDataset.__getitem__ returned, with the scipy rotations it drew recorded in float64)"""
    for mode in ("test", "train"):
        for i in range(int(g[f"len_{mode}"])):
            k = f"{mode}{i}_"
            q = g[k + "drawn_quats_f64"]
            P = g[k + "part_pcs_gt"].shape[0]
            qp = np.zeros((1, P, 4)); qp[0, :, 0] = 1.0; qp[0, :len(q) - 1] = q[1:]
            yield (g[k + "part_pcs_gt"][None], np.array([int(g[k + "num_parts"])]), np.array([int(np.argmax(g[k + "ref_part"]))]), q[:1], qp,
                   {n: g[k + n] for n in ("part_pcs", "part_trans", "part_scale", "init_pose_t", "part_rots", "init_pose_r")})


def test_fragment_prepare_oracle_vs_reference_dataset_golden(golden):
    """8f-4 pin: oracle.fragment_prepare against the outputs of the reference's own __getitem__ (denoiser/dataset/dataset.py:163-222:
    _rotate_whole_part, _recenter_ref, _recenter_pc, _rotate_pc, max-abs scale) on the rotations it drew"""
    n = 0
    for gt, num, ref, qg, qp, want in _augmentation_cases(golden("dataset")):
        pcs, trans, scale, init_t = O.fragment_prepare(gt, num, ref, qg, qp)
        pv = int(num[0])
        assert np.abs(pcs[0] - want["part_pcs"]).max() < 5e-7
        assert np.abs(trans[0] - want["part_trans"]).max() < 5e-7
        assert np.abs(scale[0][:pv] - want["part_scale"][:pv]).max() < 5e-7
        assert np.abs(init_t[0].astype(np.float64) - want["init_pose_t"]).max() < 5e-7
        assert np.array_equal(want["part_rots"][:pv], qp[0, :pv].astype(np.float32)) and np.array_equal(want["init_pose_r"], qg[0])
        n += 1
    assert n == 4
