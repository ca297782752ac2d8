"""CPU: the C-ABI library loads and exports every symbol include/pfpp.h declares; host-side logic
(packing, scheduler tables, state_dict layout, configs, synthetic data, error behaviour)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "pfpp.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfpp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    from pfpp_hip import _lib

    lib = ctypes.CDLL(str(hip_lib))
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pfpp.h but not exported"
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(syms), (bound ^ set(syms))
    assert _lib.load().pfpp_version() == 1


def header_struct_fields(name: str):
    text = (ROOT / "include" / "pfpp.h").read_text()
    body = text[text.index(f"typedef struct {name}"):text.index("} " + name + ";")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(?:const\s+)?(?:float|double|void|int64_t|int32_t)\s*\*?\s*(.*)", decl)
        names += [n.strip(" *") for n in m.group(1).split(",")]
    return names


def test_gemm_args_struct_matches_header():
    from pfpp_hip._lib import GemmArgs, GemmGradArgs

    assert header_struct_fields("pfpp_gemm_args") == [f[0] for f in GemmArgs._fields_]
    assert header_struct_fields("pfpp_gemm_grad_args") == [f[0] for f in GemmGradArgs._fields_]


def test_wrappers_reject_cpu_tensors(hip_lib):
    from pfpp_hip import ops

    with pytest.raises(ValueError, match="GPU"):
        ops.fps(torch.zeros(1, 64, 3), 8)
    with pytest.raises(ValueError, match="GPU"):
        ops.linear(torch.zeros(4, 4), torch.zeros(4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pfpp_hip import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.PfppError, match="missing"):
        _lib.load()


def test_packing_geglu_and_sa_first():
    from pfpp_hip.packing import fold_conv_bn, pack_geglu, pack_sa_first, pad_k

    w = torch.arange(8 * 64 * 3, dtype=torch.float32).reshape(512, 3)     # inner = 256
    b = torch.arange(512, dtype=torch.float32)
    wp, bp = pack_geglu(w, b)
    # packed row 64*k + j (j < 32) = value row 32*k + j; row 64*k + 32 + j = gate row 256 + 32*k + j
    for k in (0, 3, 7):
        for j in (0, 31):
            assert torch.equal(wp[64 * k + j], w[32 * k + j]) and torch.equal(wp[64 * k + 32 + j], w[256 + 32 * k + j])
            assert bp[64 * k + j] == b[32 * k + j] and bp[64 * k + 32 + j] == b[256 + 32 * k + j]
    w1 = torch.randn(16, 131)
    p1 = pack_sa_first(w1, 128)
    assert p1.shape == (16, 132) and torch.equal(p1[:, :128], w1[:, 3:]) and torch.equal(p1[:, 128:131], w1[:, :3])
    assert p1[:, 131].abs().max() == 0
    assert pad_k(torch.ones(3, 7)).shape == (3, 8)
    # BN folding == conv + batch_norm in eval mode
    conv = torch.nn.Conv2d(5, 7, 1); bn = torch.nn.BatchNorm2d(7).eval()
    bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 2); bn.bias.data.uniform_(-1, 1)
    x = torch.randn(2, 5, 3, 4)
    w_, s, t = fold_conv_bn(conv.weight.data, conv.bias.data, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var)
    got = torch.einsum("oc,bchw->bohw", w_, x) * s[None, :, None, None] + t[None, :, None, None]
    assert torch.allclose(got, bn(conv(x)), atol=1e-5)


def test_scheduler_host_tables_match_golden(golden):
    from pfpp_hip.scheduler import PiecewiseScheduler
    from oracle import pfpp_oracle as O

    g = golden("scheduler")
    s = PiecewiseScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                           beta_start=1e-4, beta_end=2e-2, clip_sample=False, timestep_spacing="leading")
    s.set_timesteps(20)
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert np.array_equal(s.timesteps.numpy(), g["timesteps"])
    assert s.config.num_train_timesteps == 1000
    o = O.PiecewiseSchedule(); o.set_timesteps(20)
    for t in s.timesteps.tolist():
        assert s.step_coefficients(t) == tuple(float(c) for c in o.step_coefficients(t))
    with pytest.raises(ValueError):
        PiecewiseScheduler(prediction_type="sample")


def test_state_dict_layout_matches_reference_keys(weights_sd):
    """strict load of state_dicts keyed exactly like the reference's modules"""
    from pfpp_hip import config
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser
    from puzzlefusion_plusplus.verifier.model.verifier import Verifier

    d = Denoiser(config.denoiser_config())
    keys = set(d.state_dict().keys())
    want = {f"denoiser.{k}" for k in weights_sd("denoiser")} | {f"encoder.{k}" for k in weights_sd("vqvae")}
    assert keys == want
    assert sum(p.numel() for p in d.denoiser.parameters()) == 57_618_183
    d.denoiser.load_state_dict(weights_sd("denoiser"), strict=True)
    d.encoder.load_state_dict(weights_sd("vqvae"), strict=True)
    v = Verifier(config.verifier_config())
    assert set(v.state_dict().keys()) == {f"verifier.{k}" for k in weights_sd("verifier")}
    assert sum(p.numel() for p in v.verifier.parameters()) == 7_892_737
    # an unfrozen encoder under autograd is refused loudly (gradients do not flow into it on this path)
    d.train()
    with pytest.raises(RuntimeError, match="frozen"):
        d.encoder.encode(torch.zeros(1, 256, 3))


def test_pack_cache_invalidation():
    from pfpp_hip.packing import PackCache

    c = PackCache()
    w = torch.zeros(3)
    calls = []
    build = lambda: calls.append(1) or {"w": w.clone()}
    c.get([w], build); c.get([w], build)
    assert len(calls) == 1
    w.add_(1)                    # in-place update bumps the version counter
    assert c.get([w], build)["w"][0] == 1 and len(calls) == 2


def test_synthetic_batch_invariants():
    from pfpp_hip import synthetic

    a = synthetic.make_batch(5, 3, num_points=256)
    b = synthetic.make_batch(5, 3, num_points=256)
    for k in a:
        assert torch.equal(a[k], b[k])                       # deterministic per puzzle id
    assert a["part_pcs"].shape == (3, 20, 256, 3) and a["part_pcs"].dtype == torch.float32
    valid = a["part_valids"].bool()
    assert (a["num_parts"] == valid.sum(1)).all() and (a["num_parts"] >= 2).all()
    assert torch.allclose(a["part_pcs"][valid].abs().amax((1, 2)), torch.ones(int(valid.sum())))
    assert a["part_pcs"][~valid].abs().max() == 0 and (a["part_scale"][~valid] == 1).all()
    assert (a["ref_part"].sum(1) == 1).all() and (a["ref_part"] & ~valid).sum() == 0
    assert torch.allclose(a["part_rots"][valid].norm(dim=-1), torch.ones(int(valid.sum())), atol=1e-6)
    q = synthetic.make_batch(5, 1, num_points=256, quantise_bits=9)["part_pcs"]
    assert torch.equal(q * 512, (q * 512).round())
    e = synthetic.make_edges(2)
    assert e["edge_features"].shape == (2, 190, 7) and e["edge_indices"].dtype == torch.int64


def test_shard_helpers():
    from pfpp_hip.parallel import balanced_assignment, shard_range

    for total, world in ((32, 8), (10, 4), (3, 8)):
        covered = []
        for r in range(world):
            a, b = shard_range(total, r, world)
            covered += list(range(a, b))
        assert covered == list(range(total))
    groups = balanced_assignment([20, 2, 3, 19, 5, 7, 2, 18], 2)
    assert sorted(sum(groups, [])) == list(range(8))
    loads = [sum([20, 2, 3, 19, 5, 7, 2, 18][i] for i in g) for g in groups]
    assert abs(loads[0] - loads[1]) <= 2
