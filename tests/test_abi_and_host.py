"""CPU: the C-ABI library loads and exports every symbol include/pfpp.h declares; host-side logic
(packing, scheduler tables, state_dict layout, configs, synthetic data, error behaviour)."""
import ctypes
import math
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
    assert _lib.load().pfpp_version() == _lib.ABI_VERSION == 2
    info = _lib.build_info(_lib.load())
    assert info["abi"] == "2" and info["arch"] == "gfx950" and info["fma_mix_insts"] == "off" and info["packed_fp32_ops"] == "off", info


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


def test_every_ctypes_mirror_has_the_size_the_library_was_compiled_with(hip_lib):
    """a struct that grew in include/pfpp.h without its mirror in pfpp_hip/_lib.py would make the library read past the caller's
    memory; load() refuses such a pair — here every struct the header defines must be in that check, with equal sizes"""
    import ctypes as C

    from pfpp_hip import _lib

    text = (ROOT / "include" / "pfpp.h").read_text()
    declared = set(re.findall(r"^}\s*pfpp_(\w+);", text, flags=re.M))
    declared |= set(re.findall(r"^typedef struct pfpp_\w+ \{[^}]*\} pfpp_(\w+);", text, flags=re.M))
    lib = _lib.load()
    assert declared == set(_lib.STRUCT_MIRRORS), declared ^ set(_lib.STRUCT_MIRRORS)
    for name, mirror in _lib.STRUCT_MIRRORS.items():
        assert lib.pfpp_abi_sizeof(name.encode()) == C.sizeof(mirror), name
    assert lib.pfpp_abi_sizeof(b"no_such_struct") == -1


def test_wrappers_reject_cpu_tensors(hip_lib):
    from pfpp_hip import ops

    with pytest.raises(ValueError, match="GPU"):
        ops.fps(torch.zeros(1, 64, 3), 8)
    with pytest.raises(ValueError, match="GPU"):
        ops.linear(torch.zeros(4, 4), torch.zeros(4, 4))


def test_round5_entry_points_refuse_bad_arguments_before_any_launch(hip_lib):
    """argument checks of the round-5 entry points (no GPU needed: they return before a launch): null pointers, widths the kernels do not
    cover, a non-positive gradient scale — error code and message through pfpp_last_error; the wrappers refuse CPU tensors"""
    import ctypes as C

    from pfpp_hip import _lib, train_ops as T

    lib = _lib.load()
    one = C.c_void_p(4096)                        # any non-null, 16-byte aligned address: never dereferenced on these paths
    assert lib.pfpp_ada_linear_bwd(None, one, one, one, one, one, one, 12, 32, 512, 1024, None) != 0 and b"null" in lib.pfpp_last_error()
    assert lib.pfpp_ada_linear_bwd(one, one, one, one, one, one, one, 12, 32, 500, 1024, None) != 0 and b"unsupported" in lib.pfpp_last_error()
    assert lib.pfpp_ada_linear_bwd(one, one, one, one, one, one, one, 12, 0, 512, 1024, None) != 0
    assert lib.pfpp_token_embed_bwd(one, one, one, one, one, one, one, one, 154, 25, 500, C.c_float(1.0), None) != 0
    assert lib.pfpp_token_embed_bwd(one, one, one, one, one, one, one, one, 154, 25, 512, C.c_float(0.0), None) != 0 and b"scale" in lib.pfpp_last_error()
    assert lib.pfpp_token_embed_bwd(one, None, one, one, one, one, one, one, 154, 25, 512, C.c_float(1.0), None) != 0
    assert lib.pfpp_token_features_t(one, one, one, one, None, None, one, one, 154, 25, None) != 0
    assert lib.pfpp_embed_pack_weights(one, one, one, one, one, one, one, 500, None) != 0
    assert lib.pfpp_token_features_t_cols(154, 25) == 3856 and lib.pfpp_ada_linear_bwd_scratch_floats(12, 1024) == 12 * 1024 * 32
    with pytest.raises(ValueError, match="GPU"):
        T.ada_linear_bwd(torch.zeros(2, 4, 64), torch.zeros(2, 4, 32), torch.zeros(2, 64, 32), torch.zeros(2, 64, 32), torch.zeros(2, 64))
    with pytest.raises(ValueError, match="GPU"):
        T.token_embed_bwd(torch.zeros(50, 32), torch.zeros(320, 64, dtype=torch.float16), torch.zeros(320, 64, dtype=torch.float16),
                          torch.zeros(32, 148), torch.zeros(32), torch.zeros(32, 147), torch.zeros(32), torch.zeros(2, 32), 2, 25)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pfpp_hip import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.PfppError, match="missing"):
        _lib.load()


def test_library_built_without_the_correctness_switches_is_refused_at_load(monkeypatch, tmp_path):
    """VERDICT r5 item 5.  A libpfpp_hip.so that was NOT compiled through pfpp_hip/build.py (here: csrc/lib.hip with a bare hipcc line,
    no `-fma-mix-insts` / `-packed-fp32-ops` target-feature switches, hence no attestation macros) must not get as far as a launch:
    pfpp_build_info() says "unattested" and _lib.load() raises; PFPP_PACKED_FP32=1 waives the second switch only."""
    import shutil
    import subprocess

    from pfpp_hip import _lib, build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = tmp_path / "libpfpp_foreign.so"
    base = [hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", f"-I{build.INCLUDE}", f"-I{build.CSRC}", str(build.CSRC / "lib.hip"), "-o", str(so)]
    subprocess.run(base, check=True, stderr=subprocess.DEVNULL)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", so)
    monkeypatch.delenv("PFPP_PACKED_FP32", raising=False)
    with pytest.raises(_lib.PfppError, match="fma_mix_insts=unattested"):
        _lib.load()
    # the mix switch attested, the packed one not: refused unless the lab override is set (then it fails later, on the symbols this stub lacks)
    so2 = tmp_path / "libpfpp_foreign2.so"           # (another file: dlopen hands an already loaded path back)
    subprocess.run(base[:-1] + [str(so2), "-DPFPP_ATTEST_NO_MIX=1"], check=True, stderr=subprocess.DEVNULL)
    monkeypatch.setattr(_lib, "LIB_PATH", so2)
    with pytest.raises(_lib.PfppError, match="packed fp32"):
        _lib.load()
    monkeypatch.setenv("PFPP_PACKED_FP32", "1")
    with pytest.raises(AttributeError):
        _lib.load()
    assert _lib.build_info(ctypes.CDLL(str(so2)))["packed_fp32_ops"] == "unattested"


def test_attention_mode_is_state_of_the_calling_thread(hip_lib):
    """pfpp_set_attention_mode is thread-local (the library keeps no process-global mutable state): another host thread keeps the default"""
    import threading

    from pfpp_hip import _lib

    lib = _lib.load()
    assert lib.pfpp_set_attention_mode(0) == 0 and lib.pfpp_get_attention_mode() == 0
    seen = []
    th = threading.Thread(target=lambda: seen.append(lib.pfpp_get_attention_mode()))
    th.start(); th.join()
    assert seen == [-1]
    assert lib.pfpp_set_attention_mode(-1) == 0 and lib.pfpp_get_attention_mode() == -1
    assert lib.pfpp_set_attention_mode(7) != 0


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


def test_fragment_blocked_weight_planes_hold_the_mfma_operand_of_every_lane():
    """packing.PW.frag() (include/pfpp.h pfpp_pw.fhi / flo, read by csrc/lnlin_small.hip and csrc/gemm_small.hip): block (row tile r,
    k-step s) is the B operand of one v_mfma_f32_32x32x16_f16 as its 64 lanes hold it — lane 32 h + n has W[32 r + n][16 s + 8 h .. + 8) —
    blocks ordered [r][s], 1 KB each, nothing added or dropped; weights whose shape does not tile are refused"""
    from pfpp_hip.packing import PW

    g = torch.Generator().manual_seed(3)
    w = torch.randn(96, 64, generator=g)
    pw = PW(w)
    fh, fl = pw.frag()
    assert fh.is_contiguous() and fh.shape == (3, 4, 2, 32, 8) and fh.dtype == torch.float16 and fl.shape == fh.shape
    flat_h, flat_l = fh.reshape(-1, 64, 8), fl.reshape(-1, 64, 8)          # [block][lane][8 halfs]
    for r, s_, lane in ((0, 0, 0), (2, 3, 63), (1, 2, 37), (0, 1, 32)):
        n, h = lane % 32, lane // 32
        blk = r * 4 + s_
        assert torch.equal(flat_h[blk, lane], pw.hi[32 * r + n, 16 * s_ + 8 * h: 16 * s_ + 8 * h + 8])
        assert torch.equal(flat_l[blk, lane], pw.lo[32 * r + n, 16 * s_ + 8 * h: 16 * s_ + 8 * h + 8])
    assert torch.equal(fh.permute(0, 3, 1, 2, 4).reshape(96, 64), pw.hi)       # a permutation of the plane
    assert pw.frag()[0] is fh                                                  # built once
    with pytest.raises(ValueError):
        PW(torch.randn(40, 64, generator=g)).frag()
    with pytest.raises(ValueError):
        PW(torch.randn(32, 147, generator=g)).frag()


def test_plane_scale_and_prescaled_weight_planes():
    """packing.plane_scale: a power of two that lifts max |w| into [2^12, 2^13); PW planes then stand for scale * w to 22 bits whatever
    the tensor's magnitude, and nothing reaches the fp16 overflow"""
    from pfpp_hip.packing import PW, plane_scale

    g = torch.Generator().manual_seed(0)
    for mag in (1e-7, 3e-3, 1.0, 77.0, 6e4, 3e8):
        w = torch.randn(64, 48, generator=g) * mag
        s = plane_scale(w)
        assert math.log2(s) % 1 == 0 and 2 ** 12 <= float(w.abs().max()) * s < 2 ** 13
        pw = PW(w)
        assert pw.scale == s and torch.isfinite(pw.hi.float()).all() and torch.isfinite(pw.lo.float()).all()
        back = (pw.hi.double() + pw.lo.double()) / s
        big = w.abs() >= w.abs().max() * 2.0 ** -16                      # elements within 16 octaves of the largest: full 22 bits
        assert ((back - w.double()).abs()[big] <= w.double().abs()[big] * 2.0 ** -21).all()
    assert plane_scale(torch.zeros(4, 8)) == 1.0 and PW(torch.zeros(4, 8)).scale == 1.0
    # (near-)denormal tensors: the exponent is clamped, scale and 1 / scale stay finite fp32 numbers, the planes finite
    tiny = torch.full((4, 8), 1e-42)
    s = plane_scale(tiny)
    assert s == 2.0 ** 100 and torch.isfinite(torch.tensor(s, dtype=torch.float32)) and float(torch.tensor(1.0 / s, dtype=torch.float32)) > 0
    pw = PW(tiny)
    assert torch.isfinite(pw.hi.float()).all() and torch.isfinite(pw.lo.float()).all()
    assert PW(torch.randn(4, 8, generator=g), prescale=False).scale == 1.0


def test_balanced_assignment_and_fragment_counts():
    """what bench.py --gpus N deals out: synthetic.num_parts_of == the count make_puzzle draws; equal-count assignment keeps the
    per-rank batch size and tightens the spread of valid fragments"""
    from pfpp_hip import synthetic
    from pfpp_hip.parallel import balanced_assignment

    for pid in (0, 7, 1003, 2031):
        assert synthetic.num_parts_of(pid) == int(synthetic.make_puzzle(pid, num_points=64)["part_valids"].sum())
    pool = [1000 * r + i for r in range(4) for i in range(8)]
    counts = [synthetic.num_parts_of(i) for i in pool]
    assign = balanced_assignment(counts, 4, equal_count=True)
    assert sorted(sum(assign, [])) == list(range(32)) and all(len(a) == 8 for a in assign)
    loads = [sum(counts[j] for j in a) for a in assign]
    in_order = [sum(counts[8 * r: 8 * r + 8]) for r in range(4)]
    assert max(loads) - min(loads) <= max(in_order) - min(in_order) and max(loads) - min(loads) <= max(counts)
    with pytest.raises(ValueError):
        balanced_assignment(counts[:-1], 4, equal_count=True)
    ids = [3, 1001]
    b = synthetic.make_batch(0, 2, num_points=64, ids=ids)
    assert int(b["part_valids"][1].sum()) == synthetic.num_parts_of(1001)


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


def test_on_disk_formats_round_trip(tmp_path):
    """synthetic puzzles written in the reference's pc_data / matching_data / verifier_data
    layouts are read back by the dataset drop-ins with the reference's keys, shapes and invariants"""
    import subprocess
    import sys
    from types import SimpleNamespace as NS

    subprocess.run([sys.executable, str(ROOT / "tools" / "make_synthetic_dataset.py"), str(tmp_path), "--n", "4", "--points", "200"],
                   check=True)
    from puzzlefusion_plusplus.denoiser.dataset.dataset import GeometryLatentDataset, build_test_dataloader
    from puzzlefusion_plusplus.verifier.dataset.dataset import VerifierDataset

    cfg = NS(data=NS(max_num_part=20, matching_data_path=str(tmp_path / "matching_data"), data_val_dir=str(tmp_path / "pc_data" / "train"),
                     overfit=-1, val_batch_size=1, num_workers=0), model=NS(multiple_ref_parts=True))
    ds = GeometryLatentDataset(cfg, str(tmp_path / "pc_data" / "train"), -1, "train")
    assert len(ds) == 3
    np.random.seed(0)
    s = ds[0]
    pv = s["num_parts"]
    for key, shape in (("part_pcs", (20, 200, 3)), ("part_rots", (20, 4)), ("part_trans", (20, 3)), ("part_scale", (20, 1)),
                       ("part_valids", (20,)), ("ref_part", (20,)), ("graph", (20, 20)), ("part_pcs_gt", (20, 200, 3))):
        assert tuple(np.asarray(s[key]).shape) == shape, key
    assert np.abs(s["part_pcs"][:pv]).max(axis=(1, 2)).round(5).tolist() == [1.0] * pv     # max-abs normalised
    assert np.abs(s["part_pcs"][pv:]).max() == 0 and (s["part_scale"][pv:] == 1).all()
    assert np.abs(s["part_pcs"][:pv].mean(1)).max() < 1e-5                                  # recentred
    # the stored pose takes the normalised fragment back to the (rotated, recentred) assembly
    from scipy.spatial.transform import Rotation as R

    q = s["part_rots"][0][[1, 2, 3, 0]]
    back = R.from_quat(q).apply(s["part_pcs"][0] * s["part_scale"][0]) + s["part_trans"][0]
    glob = R.from_quat(s["init_pose_r"][[1, 2, 3, 0]]).inv().apply(ds.data_list[0]["part_pcs_gt"][0]) - s["init_pose_t"]
    assert np.abs(back - glob).max() < 1e-4
    # test mode: matching data attached, by-area points in each part's own frame
    t = GeometryLatentDataset(cfg, str(tmp_path / "pc_data" / "train"), -1, "test")
    st = t[1]
    assert st["part_pcs_by_area"].shape == (5000, 3) and len(st["correspondences"]) == st["edges"].shape[0]
    assert st["edges"].shape[1] == 2 and (st["edges"][:, 0] > st["edges"][:, 1]).all()       # (idx2, idx1)
    batch = next(iter(build_test_dataloader(cfg)))
    assert batch["part_pcs"].shape == (1, 20, 200, 3) and batch["part_pcs_by_area"].shape == (1, 5000, 3)
    # device_augment: geometry only
    raw = GeometryLatentDataset(cfg, str(tmp_path / "pc_data" / "train"), -1, "train", device_augment=True)[0]
    assert "part_pcs" not in raw and raw["part_pcs_gt"].shape == (20, 200, 3)
    v = VerifierDataset(str(tmp_path / "verifier_data"), -1, "all")
    e = v[0]
    assert e["edge_features"].shape == (190, 7) and e["edge_indices"].shape == (190, 2) and e["edge_valids"].sum() == e["num_edges"]
    n = e["num_edges"]
    assert np.allclose(e["edge_features"][:n, :6].sum(1)[e["edge_features"][:n, 6] > 0], 1.0, atol=1e-5)
    # inference outputs in the renderer's layout
    from pfpp_hip import io as pfio

    files = pfio.save_inference_data(str(tmp_path / "inference" / "7"), trajectory=np.zeros((6, pv, 7), np.float32),
                                     gt=np.zeros((pv, 7), np.float32), init_pose=np.zeros(7, np.float32), mesh_file_path="a/b", acc=0.5)
    assert [Path(f).name for f in files] == ["predict_0.5.npy", "gt.npy", "init_pose.npy", "mesh_file_path.txt"]
    assert np.load(files[0]).shape == (6, pv, 7)


def test_dataset_dropins_equal_the_reference_loaders(golden, tmp_path):
    """the drop-in GeometryLatentDataset / VerifierDataset read the reference's on-disk formats and return what
    the REFERENCE's own dataset classes return on the same files and numpy seed (tests/golden/dataset.npz, written by
    tools/make_goldens.py from puzzlefusion_plusplus/denoiser/dataset/dataset.py and verifier/dataset/dataset.py of the reference)"""
    import subprocess
    import sys
    from pathlib import Path
    from types import SimpleNamespace as NS

    import numpy as np

    root = Path(__file__).resolve().parents[1]
    subprocess.run([sys.executable, str(root / "tools" / "make_synthetic_dataset.py"), str(tmp_path), "--n", "3", "--points", "64"],
                   check=True, capture_output=True)
    from puzzlefusion_plusplus.denoiser.dataset.dataset import GeometryLatentDataset
    from puzzlefusion_plusplus.verifier.dataset.dataset import VerifierDataset

    g = golden("dataset")
    cfg = NS(data=NS(max_num_part=20, matching_data_path=str(tmp_path / "matching_data")), model=NS(multiple_ref_parts=False))
    checked = 0
    for mode in ("test", "train"):
        ds = GeometryLatentDataset(cfg, str(tmp_path / "pc_data" / "train"), -1, mode)
        assert len(ds) == int(g[f"len_{mode}"])
        for i in range(len(ds)):
            np.random.seed(100 + i)
            item = ds[i]
            for k, v in item.items():
                if k == "correspondences":
                    cat = np.concatenate([np.asarray(c).reshape(-1, 2) for c in v]) if len(v) else np.zeros((0, 2), np.int64)
                    assert np.array_equal(cat, g[f"{mode}{i}_corr_cat"]) and [len(c) for c in v] == g[f"{mode}{i}_corr_len"].tolist()
                    checked += 1
                elif f"{mode}{i}_{k}" in g:
                    want = g[f"{mode}{i}_{k}"]
                    got = np.asarray(v)
                    assert got.shape == want.shape, (mode, i, k)
                    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 1e-6, (mode, i, k)
                    checked += 1
            assert all(key.split("_", 1)[1] in item or key.endswith(("corr_cat", "corr_len", "drawn_quats_f64")) for key in g if key.startswith(f"{mode}{i}_"))
    vd = VerifierDataset(str(tmp_path / "verifier_data"), -1, "train")
    assert len(vd) == int(g["len_verifier"])
    for i in range(len(vd)):
        for k, val in vd[i].items():
            assert np.array_equal(np.asarray(val), g[f"v{i}_{k}"]), (i, k)
            checked += 1
    assert checked >= 60


def test_packed_fp32_forwarding_scanner_finds_the_pattern():
    """tools/diag/pk_hazard_scan.py (the survey behind build.py's NO_PK: 108 packed fp32 results in 39 kernels were consumed by the next
    vector instruction with only scalar fillers in between): a consumer behind an s_addc is reported, the same behind an s_nop or further
    down the stream is not"""
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools" / "diag"))
    import pk_hazard_scan as scanner

    head = "0000000000001000 <k>:\n"
    bad = head + ("\tv_pk_add_f32 v[24:25], v[26:27], v[24:25]          // 000000001000: D3B24018\n"
                  "\ts_addc_u32 s13, s13, 0                             // 000000001008: 820D800D\n"
                  "\tv_min_f32_e32 v17, v2, v24                         // 00000000100C: 14223102\n"
                  "\ts_endpgm                                           // 000000001010: BF810000\n")
    good = head + ("\tv_pk_add_f32 v[24:25], v[26:27], v[24:25]          // 000000001000: D3B24018\n"
                   "\ts_nop 0                                            // 000000001008: BF800000\n"
                   "\tv_min_f32_e32 v17, v2, v24                         // 00000000100C: 14223102\n"
                   "\ts_endpgm                                           // 000000001010: BF810000\n")
    other = head + ("\tv_pk_add_f32 v[24:25], v[26:27], v[24:25]          // 000000001000: D3B24018\n"
                    "\ts_addc_u32 s13, s13, 0                             // 000000001008: 820D800D\n"
                    "\tv_min_f32_e32 v17, v2, v30                         // 00000000100C: 14223D02\n"
                    "\ts_endpgm                                           // 000000001010: BF810000\n")
    assert sum(len(v) for v in scanner.scan(bad.splitlines(True)).values()) == 1
    assert not scanner.scan(good.splitlines(True)) and not scanner.scan(other.splitlines(True))


def test_library_has_no_fused_mixed_precision_conversions(hip_lib, tmp_path):
    """The hi / lo split (csrc/pfpp_common.h) needs ONE fp16 rounding of its argument: `hi = f16(x)`, `lo = f16(x - hi)`.  With the
    gfx950 mix instructions available the compiler may take the stored hi from `v_fma_mixlo_f16` (the exact a * b + c rounded once)
    while the subtraction sees `v_cvt_f16_f32` of the rounded fp32 value — two hi's that differ by an fp16 ulp in the rare
    double-rounding case.  pfpp_hip.build compiles every translation unit with that target feature off; this
    disassembles the gfx950 code objects of the library that was built and checks that no such instruction is left — so a compiler
    bump that renames / ignores the flag shows up here, not as a silent one-ulp operand error."""
    import shutil
    import subprocess

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not available")
    lib = tmp_path / "libpfpp_hip.so"
    shutil.copy(hip_lib, lib)
    subprocess.run([str(objdump), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = sorted(tmp_path.glob("libpfpp_hip.so.*gfx950*"))
    assert objs, "no gfx950 code object found in libpfpp_hip.so"
    n_inst = n_cvt = 0
    for o in objs:
        proc = subprocess.Popen([str(objdump), "-d", "--mcpu=gfx950", str(o)], stdout=subprocess.PIPE, text=True)
        for line in proc.stdout:
            if "v_fma_mix" in line or "v_mad_mix" in line:
                proc.kill()
                raise AssertionError(f"mixed-precision fused conversion in the library: {line.strip()}")
            if "v_pk_add_f32" in line or "v_pk_mul_f32" in line or "v_pk_fma_f32" in line:
                # round 5: with the packed fp32 instructions fps_kernel selected a wrong farthest point once in
                # 10^2 .. 10^4 launches next to a GEMM of another stream (stale running minimum in the upper lanes of a wave);
                # pfpp_hip.build switches the target feature off for the whole library (tools/diag/fps_race.py; GPU side:
                # test_sampling_chain_is_exact_next_to_a_gemm_on_another_stream)
                proc.kill()
                raise AssertionError(f"packed fp32 instruction in the library: {line.strip()}")
            n_inst += "v_mfma_" in line
            n_cvt += "v_cvt_f16_f32" in line or "v_cvt_pk_f16_f32" in line
        assert proc.wait() == 0
    assert n_inst > 1000 and n_cvt > 1000          # it really was the kernels' code that was read


def test_library_never_touches_a_register_with_an_lds_read_in_flight(hip_lib, tmp_path):
    """The cause of round 4's 'two-rank' gradient discrepancy.  The plane GEMM issues its fragment
    reads from inline asm (ds_read_b128 / ds_read_b64_tr_b16) and waits for them with an asm statement that names the destination
    registers; the compiler's own s_waitcnt insertion does not know those registers are in flight.  In the column-sum instantiations of
    the weight-gradient GEMM hipcc placed eight `v_mov_b64` copies of freshly read fragments on a control-flow edge BEFORE our wait —
    harmless while the LDS pipeline answers within ~200 instructions, wrong (one 32 x 32 tile of dW off by one 16-deep step) when
    another process's LDS-bound waves share the CU.  csrc/gemm_pl.hip now waits for every fragment read inside the region that issued
    it; this test disassembles the built library and runs tools/diag/lds_hazard_scan.py (lgkmcnt model over each kernel's control-flow
    graph) over every kernel: no instruction may read or write a VGPR that an outstanding LDS read targets."""
    import shutil
    import subprocess
    import sys

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools" / "diag"))
    import lds_hazard_scan as scanner

    # the scanner itself: a copy of a pending register is found, the same copy behind the wait is not
    head = "0000000000001000 <k>:\n"
    bad = head + ("\tds_read_b64_tr_b16 v[10:11], v2                 // 000000001000: D9C60000\n"
                  "\tv_mov_b64_e32 v[4:5], v[10:11]                 // 000000001008: 7E087110\n"
                  "\ts_waitcnt lgkmcnt(0)                           // 00000000100C: BF8CC07F\n"
                  "\ts_endpgm                                       // 000000001010: BF810000\n")
    good = head + ("\tds_read_b64_tr_b16 v[10:11], v2                 // 000000001000: D9C60000\n"
                   "\ts_waitcnt lgkmcnt(0)                           // 000000001008: BF8CC07F\n"
                   "\tv_mov_b64_e32 v[4:5], v[10:11]                 // 00000000100C: 7E087110\n"
                   "\ts_endpgm                                       // 000000001010: BF810000\n")
    assert len(scanner.scan(bad.splitlines(True))) == 1 and not scanner.scan(good.splitlines(True))

    lib = tmp_path / "libpfpp_hip.so"
    shutil.copy(hip_lib, lib)
    subprocess.run([str(objdump), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = sorted(tmp_path.glob("libpfpp_hip.so.*gfx950*"))
    assert objs, "no gfx950 code object found in libpfpp_hip.so"
    n_kernels = n_reads = n_wd = 0
    for o in objs:
        text = subprocess.run([str(objdump), "-d", "--mcpu=gfx950", str(o)], check=True, capture_output=True, text=True).stdout
        n_kernels += text.count(">:\n")
        n_reads += text.count("ds_read_b64_tr_b16")
        rep = scanner.scan(text.splitlines(True))
        assert not rep, {scanner.demangle(k)[:120]: [(h[1][:60], h[2][1][:60]) for h in v[:3]] for k, v in rep.items()}
        # the same hazard class on the vector-memory side: the weight-direct GEMM (the only kernel that loads VGPRs from inline asm)
        # must never have one of its asm-loaded registers copied
        for k, moves in scanner.moves_of_loaded_registers(text.splitlines(True), "gemm_wd_").items():
            n_wd += 1
            assert not moves, (scanner.demangle(k)[:100], moves[:4])
    assert n_kernels > 100 and n_reads > 500 and n_wd >= 5          # it really was the kernels' code that was scanned
