"""GPU (-m gpu): the training step — DenoiserTrainEngine forward / loss / backward /
AdamW against (1) the gradients of the REFERENCE module's autograd committed in tests/golden/train.npz and
(2) torch autograd through the CPU oracle on the same inputs, including the train-mode dropouts (the masks
are read back from the kernel's counter-based generator and handed to the oracle)."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



def _free_port() -> int:
    """an unused TCP port on 127.0.0.1 (rendezvous of the spawned ranks)"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def T(a):
    return torch.from_numpy(np.asarray(a))


class NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_module(weights_sd, dev):
    from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer

    cfg = NS(model=NS(embed_dim=512, out_channels=7, num_layers=6, num_heads=8, num_dim=64, num_point=25))
    m = DenoiserTransformer(cfg)
    m.load_state_dict(weights_sd("denoiser"), strict=True)
    return m.to(dev)


def golden_inputs(golden, dev=None):
    g, t = golden("denoiser"), golden("train")
    keys = ("x", "timesteps", "latent", "xyz", "part_valids", "scale", "ref_part")
    inp = [T(g[k]) for k in keys]
    noise = T(t["noise"])
    if dev is not None:
        inp = [v.to(dev) for v in inp]
        noise = noise.to(dev)
    return inp, noise, t


def rel(got, want):
    return float((got.double().cpu() - want.double()).abs().max() / (want.double().abs().max() + 1e-30))


def test_loss_and_grads_vs_reference_golden(golden, weights_sd, dev):
    from pfpp_hip.train import DenoiserTrainEngine

    eng = DenoiserTrainEngine(make_module(weights_sd, dev))
    inp, noise, t = golden_inputs(golden, dev)
    loss = eng.loss_and_grads(*inp, noise, train=False)
    assert abs(float(loss) - float(t["loss"])) < 2e-5 * float(t["loss"])
    named = dict(eng.module.named_parameters())
    worst_norm, worst_sample = 0.0, 0.0
    for i, name in enumerate(t["names"].tolist()):
        gr = named[name].grad
        assert gr is not None and gr.data_ptr() == eng.flat.view(eng.flat.grads, name).data_ptr()
        n_ref = float(t["grad_norm"][i])
        worst_norm = max(worst_norm, abs(float(gr.double().norm()) - n_ref) / (n_ref + 1e-30))
        flat = gr.flatten()
        smp = flat[:: max(1, flat.numel() // 16)][:16].cpu().numpy() if flat.numel() >= 16 else np.pad(flat.cpu().numpy(), (0, 16 - flat.numel()))
        worst_sample = max(worst_sample, float(np.abs(smp - t["grad_sample"][i]).max()) / (float(t["grad_absmax"][i]) + 1e-30))
    assert worst_norm < 1e-4, worst_norm
    assert worst_sample < 1e-4, worst_sample


def _oracle_grads(weights_sd, inp, noise, drop=None):
    from oracle import pfpp_oracle as O

    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoding.pe") for k, v in weights_sd("denoiser").items()}
    pred = O.denoiser_forward(sd, *inp, drop=drop)
    loss = O.denoiser_loss(pred, noise, inp[4], inp[6])
    loss.backward()
    return sd, pred.detach(), loss.detach()


def test_train_mode_dropout_vs_oracle_autograd(golden, weights_sd, dev):
    """full train-mode step: the oracle applies the very masks the kernels generate"""
    from pfpp_hip import train_ops as TO
    from pfpp_hip.train import DenoiserTrainEngine

    eng = DenoiserTrainEngine(make_module(weights_sd, dev))
    inp_c, noise_c, _ = golden_inputs(golden)
    inp = [v.to(dev) for v in inp_c]
    seed = 20240917
    B, P, L = 2, 20, 25
    valid = inp_c[4].reshape(-1).bool()
    slot = torch.nonzero(valid).flatten()
    Fv = slot.numel()

    def drop(site, tensor):
        width = tensor.shape[-1]
        p = eng.p_token if site == 0 else eng.p_layer
        keep_c = TO.dropout_mask(Fv * L * width, p, seed, site, dev).view(Fv, L, width).cpu().to(tensor.dtype)
        keep = torch.ones(B * P, L, width, dtype=tensor.dtype)
        keep[slot] = keep_c                                   # compact row order = ascending valid slots
        return tensor * keep.view(B, P * L, width) / (1 - p)

    sd, pred_o, loss_o = _oracle_grads(weights_sd, inp_c, noise_c, drop)
    pred, ctx = eng.forward(*inp, seed=seed, train=True)
    sel = valid.view(B, P)
    assert (pred.cpu() - pred_o)[sel].abs().max() < 1e-4
    eng.flat.zero_grad()
    loss = eng.loss_and_grads(*inp, noise_c.to(dev), seed=seed, train=True)
    assert abs(float(loss) - float(loss_o)) < 2e-5 * float(loss_o)
    named = dict(eng.module.named_parameters())
    worst = max(rel(named[n].grad, sd[n].grad) for n in named)
    assert worst < 2e-4, worst


@pytest.mark.parametrize("armed", [False, True])
def test_two_optimizer_steps_vs_oracle(golden, weights_sd, dev, armed):
    """two AdamW steps (eval-mode dropout).  Adam's first updates are sign(g)*lr, so elements whose gradient is
    within rounding of zero may legitimately move the other way: bounded count, bounded size.
    armed: the optimizer-in-backward form (arm_optimizer: every layer's slice updated under the rest of the backward,
    optimizer_step finishes the remaining ranges) must land on the same parameters."""
    from oracle import pfpp_oracle as O
    from pfpp_hip.train import DenoiserTrainEngine

    eng = DenoiserTrainEngine(make_module(weights_sd, dev))
    inp_c, noise_c, _ = golden_inputs(golden)
    inp = [v.to(dev) for v in inp_c]
    names = [n for n, _ in eng.module.named_parameters()]
    sd = {k: v.clone() for k, v in weights_sd("denoiser").items()}
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    lr = 2e-4
    for step in (1, 2):
        req = {k: t.clone().requires_grad_(k in m) for k, t in sd.items()}
        O.denoiser_loss(O.denoiser_forward(req, *inp_c), noise_c, inp_c[4], inp_c[6]).backward()
        with torch.no_grad():
            O.adamw_step([sd[n] for n in names], [req[n].grad for n in names], [m[n] for n in names], [v[n] for n in names], step)
        eng.flat.zero_grad()
        if armed:
            eng.arm_optimizer(lr=lr)
        eng.loss_and_grads(*inp, noise_c.to(dev), train=False)
        if armed:
            assert set(eng.flat.layer_ranges) <= set(eng._early)             # every layer's slice went early (+ the timestep tables' untouched rows)
        eng.optimizer_step(lr=lr)
        assert eng._armed is None and eng._early == []
    named = dict(eng.module.named_parameters())
    total, off = 0, 0
    for n in names:
        d = (named[n].detach().cpu() - sd[n]).abs()
        assert float(d.max()) <= 2 * 2 * lr * 1.01, n
        total += d.numel()
        off += int((d > 2e-6).sum())
    assert off / total < 2e-3, off / total
    # the refreshed split planes track the parameters
    f = eng.flat
    assert ((f.hi.float() + f.lo.float() - f.params).abs() <= 2.0 ** -21 * f.params.abs() + 1e-7).all()
    # and the next forward uses them: eval prediction == oracle forward with the updated weights
    pred, _ = eng.forward(*inp, train=False)
    want = O.denoiser_forward(sd, *inp_c)
    sel = inp_c[4].bool()
    assert (pred.cpu() - want)[sel].abs().max() < 2e-3     # two sign-descent steps apart on ~1e-3 of the weights


@pytest.mark.parametrize("armed", [False, True])
def test_timestep_table_rows_without_a_gradient_are_left_alone(golden, weights_sd, dev, armed, monkeypatch):
    """round 6: the closing AdamW of a step walks the AdaLN timestep tables through the bitmap of rows that ever received a gradient
    (engine._tab_active, set by silu_embed_bwd) — the other rows' update is exactly the identity with the reference's hyper-parameters
    (the two update kernels bit for bit on equal inputs: test_adamw_over_the_active_rows_equals_one_pass_over_the_tables).  Three steps
    with different timestep draws on an engine with the bitmap and on one without it (PFPP_TRAIN_TABLES_ACTIVE=0, every row every
    step): in BOTH every table row outside the marked set still holds its initial bits with zero moments — i.e. skipping them changes
    nothing — the marked set is exactly the rows drawn (one beyond the 1,000 training timesteps included) plus a row whose moment was
    written from outside (tables_state_changed() picks it up), and the marked rows moved."""
    from pfpp_hip.train import DenoiserTrainEngine

    inp_c, noise_c, _ = golden_inputs(golden)
    drawn = set()
    for active in ("1", "0"):
        monkeypatch.setenv("PFPP_TRAIN_TABLES_ACTIVE", active)
        eng = DenoiserTrainEngine(make_module(weights_sd, dev))
        assert (eng._tab_active is not None) == (active == "1")
        f = eng.flat
        n_emb, C = f.named["transformer_layers.0.norm1.emb.weight"].shape
        n_tab = f.offset["transformer_layers.0.norm1.linear.weight"]
        p0 = f.params[:n_tab].clone()
        inp = [v.to(dev) for v in inp_c]
        for step in range(3):
            t = (inp_c[1].to(dev) + 37 * step) % 1000                   # other rows every step
            if step == 1:
                t[0] = 2500                                             # a row outside the training range
            inp[1] = t
            drawn |= set(int(x) for x in t.tolist())
            eng.flat.zero_grad()
            if armed:
                eng.arm_optimizer(lr=2e-4, zero_grad=True)
            eng.loss_and_grads(*inp, noise_c.to(dev), train=False)
            eng.optimizer_step(lr=2e-4, zero_grad=armed)
            if step == 1:
                f.exp_avg[5 * n_emb * C + 2900 * C + 3] = 1e-3          # table 5, row 2900: a moment from "a checkpoint"
                eng.tables_state_changed()
        torch.cuda.synchronize()
        want = sorted(drawn | {2900})
        rest = torch.ones(n_emb, dtype=torch.bool, device=dev)
        rest[torch.tensor(want, device=dev)] = False
        tab = lambda buf: buf[:n_tab].view(-1, n_emb, C)
        assert torch.equal(tab(f.params)[:, rest], p0.view(-1, n_emb, C)[:, rest])
        assert float(tab(f.exp_avg)[:, rest].abs().max()) == 0.0 and float(tab(f.exp_avg_sq)[:, rest].abs().max()) == 0.0
        assert torch.equal((tab(f.hi).float() + tab(f.lo).float())[:, rest], (tab(f.hi).float() + tab(f.lo).float())[:, rest])
        moved = (tab(f.params) != p0.view(-1, n_emb, C)).any(dim=2).any(dim=0)
        assert bool(moved[torch.tensor(sorted(drawn), device=dev)].all()) and bool(moved[2900])
        if active == "1":
            bits = eng._tab_active.cpu().numpy().view("uint32")
            assert [r for r in range(n_emb) if (int(bits[r >> 5]) >> (r & 31)) & 1] == want
        del eng, f, p0, tab, moved, rest
    # two engines' worth of flat buffers go back to the driver: left in torch's cache they sent the per-step allocations of the timing test
    # further down (test_module_surface_runs_the_benchmarked_schedule) to hipMalloc every iteration when the whole suite ran in one process
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_closing_adamw_runs_under_the_weight_gradient_streams_tail(golden, weights_sd, dev, monkeypatch):
    """round 6: in an armed single-rank step the join with the weight-gradient stream moves from the end of the backward behind the
    closing AdamW launches of optimizer_step() (tables, AdaLN linears, embeddings, heads: none of their gradients comes from that
    stream's tail; the heads' wide weight gradients are ordered by an event).  Three armed steps with the deferred join against three
    with the join at the end of the backward (PFPP_TRAIN_TAIL_OVERLAP=0): first and second moments of EVERY parameter agree to the
    run-to-run noise of the gradient atomics (a launch that read a gradient before it was final would be off by the gradient itself),
    the engine is joined again after optimizer_step(), and the overflow flag of the step reaches the host."""
    from pfpp_hip.train import DenoiserTrainEngine

    inp_c, noise_c, _ = golden_inputs(golden)
    res = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("PFPP_TRAIN_TAIL_OVERLAP", overlap)
        eng = DenoiserTrainEngine(make_module(weights_sd, dev))
        inp = [v.to(dev) for v in inp_c]
        for step in range(3):
            eng.flat.zero_grad()
            eng.arm_optimizer(lr=2e-4, zero_grad=True)
            eng.loss_and_grads(*inp, noise_c.to(dev), train=False)
            assert eng._join_pending == (overlap == "1")
            eng.optimizer_step(lr=2e-4, zero_grad=True)
            assert not eng._join_pending
        torch.cuda.synchronize()
        assert eng.overflow_steps == 0
        f = eng.flat
        res.append((f.exp_avg.clone(), f.exp_avg_sq.clone(), dict(f.offset), f.order, {n: f.named[n].numel() for n in f.order}))
    (m1, v1, off, order, numel), (m0, v0, _, _, _) = res
    for n in order:
        a, b = off[n], off[n] + numel[n]
        sm, sv = float(m0[a:b].abs().max()), float(v0[a:b].abs().max())
        assert float((m1[a:b] - m0[a:b]).abs().max()) <= 2e-5 * sm + 1e-12, n
        assert float((v1[a:b] - v0[a:b]).abs().max()) <= 4e-5 * sv + 1e-20, n


def test_module_train_mode_autograd_and_optimizer(golden, weights_sd, dev):
    """the drop-in surface: module.train(); loss.backward(); FusedAdamW.step() == the engine driven directly"""
    import torch.nn.functional as F

    from pfpp_hip.optim import FusedAdamW
    from pfpp_hip.train import DenoiserTrainEngine

    inp, noise, _ = golden_inputs(golden, dev)
    sel = inp[4].bool() & ~inp[6]
    m = make_module(weights_sd, dev)
    keys_before = list(m.state_dict().keys())
    m.train()
    opt = FusedAdamW(m.train_engine())
    assert list(m.state_dict().keys()) == keys_before
    torch.manual_seed(123)
    pred = m(*inp)
    assert pred.requires_grad
    loss = F.mse_loss(pred[sel], noise[sel])
    opt.zero_grad()
    loss.backward()
    g_mod = m.train_engine().flat.grads.clone()
    # the same step through the engine API (same seed stream)
    eng = DenoiserTrainEngine(make_module(weights_sd, dev))
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    loss2 = eng.loss_and_grads(*inp, noise, seed=seed, train=True)
    assert abs(float(loss) - float(loss2)) < 1e-6 * float(loss2)
    assert rel(g_mod, eng.flat.grads.cpu()) < 1e-5          # atomics: summation order differs between runs
    opt.step()
    eng.optimizer_step()
    assert rel(m.train_engine().flat.params, eng.flat.params.cpu()) < 1e-5 or \
        float((m.train_engine().flat.params - eng.flat.params).abs().gt(1e-6).float().mean()) < 1e-3
    assert int(opt.state[m.ref_part_emb.weight]["step"]) == 1
    # optimizer state_dict has torch.optim.AdamW's layout and round-trips
    sd_opt = opt.state_dict()
    assert set(sd_opt["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    opt2 = FusedAdamW(eng)
    opt2.load_state_dict(sd_opt)
    assert torch.equal(eng.flat.exp_avg, m.train_engine().flat.exp_avg) and eng.step_count == 1
    # eval forward after the step uses the updated weights (pack cache invalidated); load_state_dict refreshes planes
    m.eval()
    with torch.no_grad():
        e1 = m(*inp)
    m.load_state_dict(weights_sd("denoiser"), strict=True)
    m.train()
    with torch.no_grad():
        torch.manual_seed(5)
        t1 = m(*inp)
    fresh = make_module(weights_sd, dev).train()
    with torch.no_grad():
        torch.manual_seed(5)
        t2 = fresh(*inp)
    assert torch.equal(t1, t2)
    assert (e1 - t1).abs().max() > 0


def test_reference_optimizer_checkpoint_loads_by_position(weights_sd, dev):
    """the reference builds torch.optim.AdamW(self.parameters()) over the WHOLE Denoiser module (denoiser.py:230-237): transformer
    parameters in registration order, then the frozen encoder's without state.  An optimizer state_dict of that layout loads into
    Denoiser.configure_optimizers()'s FusedAdamW and every moment lands on the parameter of the same NAME (torch maps state by
    position; norm1 / norm2 tables have identical shapes, so a permuted group would go unnoticed without this check)"""
    from pfpp_hip import config
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    torch.manual_seed(0)
    model = Denoiser(config.denoiser_config())
    model.encoder.load_state_dict(weights_sd("vqvae")); model.denoiser.load_state_dict(weights_sd("denoiser"))
    model = model.to(dev).train()
    for p_ in model.encoder.parameters():
        p_.requires_grad = False
    names = [n for n, _ in model.named_parameters()]
    # a "reference" optimizer state: plain torch AdamW over the module's parameters, one step on synthetic gradients
    ref_opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    gen = torch.Generator(device=dev).manual_seed(1)
    for n, p_ in model.named_parameters():
        if p_.requires_grad:
            p_.grad = torch.randn(p_.shape, device=dev, generator=gen) * 1e-3
    ref_opt.step()
    sd = ref_opt.state_dict()
    want = {names[i]: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in sd["state"].items()}
    assert len(sd["param_groups"][0]["params"]) == len(names) and all(n.startswith("denoiser.") for n in want)
    for p_ in model.parameters():
        p_.grad = None
    opt = model.configure_optimizers()
    opt = opt["optimizer"] if isinstance(opt, dict) else opt
    assert len(opt.param_groups[0]["params"]) == len(names)
    opt.load_state_dict(sd)
    flat = model.denoiser.train_engine().flat
    for n in flat.order:
        full = "denoiser." + n
        assert torch.equal(flat.view(flat.exp_avg, n), want[full]["exp_avg"]), n
        assert torch.equal(flat.view(flat.exp_avg_sq, n), want[full]["exp_avg_sq"]), n
    assert model.denoiser.train_engine().step_count == 1
    # and back: the state_dict written by the fused optimizer has the same positional layout
    sd2 = opt.state_dict()
    assert sd2["param_groups"][0]["params"] == sd["param_groups"][0]["params"] and sorted(sd2["state"]) == sorted(sd["state"])
    # an optimizer over the transformer's parameters only refuses the reference layout loudly
    from pfpp_hip.optim import FusedAdamW

    with pytest.raises(ValueError, match="parameters"):
        FusedAdamW(model.denoiser.train_engine()).load_state_dict(sd)


def test_foreign_zero_grad_does_not_leave_stale_gradients(golden, weights_sd, dev):
    """module.zero_grad() / another optimizer's zero_grad(set_to_none=True) sets every .grad to None; the next backward re-points
    them at the flat buffer — which must then start from zero, not from the previous iteration's gradients"""
    inp, noise, _ = golden_inputs(golden, dev)
    m = make_module(weights_sd, dev)
    eng = m.train_engine()
    eng.loss_and_grads(*inp, noise, seed=3, train=False)
    torch.cuda.synchronize()
    g1 = eng.flat.grads.clone()
    m.zero_grad(set_to_none=True)                       # what a generic training loop does
    assert all(p.grad is None for p in m.parameters())
    eng.loss_and_grads(*inp, noise, seed=3, train=False)
    torch.cuda.synchronize()
    assert rel(eng.flat.grads.cpu(), g1.cpu()) < 1e-5   # not 2 x g1
    # dropping only some gradients clears only their slices: the others keep accumulating, like torch parameters do
    m.ref_part_emb.weight.grad = None
    eng.loss_and_grads(*inp, noise, seed=3, train=False)
    torch.cuda.synchronize()
    assert rel(eng.flat.view(eng.flat.grads, "ref_part_emb.weight").cpu(), eng.flat.view(g1, "ref_part_emb.weight").cpu()) < 1e-5
    assert rel(eng.flat.view(eng.flat.grads, "shape_embedding.bias").cpu(), 2 * eng.flat.view(g1, "shape_embedding.bias").cpu()) < 1e-5


def test_optimizer_step_with_fused_zero_grad(golden, weights_sd, dev):
    """optimizer_step(zero_grad=True) == optimizer_step() followed by zero_grad(): same parameters, cleared gradients, and a
    backward that follows without any zero_grad() accumulates onto zeros (the buffer is marked dirty again)"""
    inp, noise, _ = golden_inputs(golden, dev)
    res = []
    for fused in (False, True):
        m = make_module(weights_sd, dev)
        eng = m.train_engine()
        eng.loss_and_grads(*inp, noise, seed=3, train=False)
        g1 = eng.flat.grads.clone()
        eng.optimizer_step(lr=1e-3, weight_decay=1e-2, zero_grad=fused)
        if fused:
            assert float(eng.flat.grads.abs().max()) == 0.0
        eng.flat.zero_grad()
        eng.loss_and_grads(*inp, noise, seed=3, train=False)      # step 2
        g2 = eng.flat.grads.clone()
        eng.optimizer_step(lr=1e-3, weight_decay=1e-2, zero_grad=fused)
        eng.loss_and_grads(*inp, noise, seed=3, train=False)      # no zero_grad() in between: accumulates onto whatever is there
        torch.cuda.synchronize()
        res.append((eng.flat.params.clone(), g2, eng.flat.grads.clone(), g1))
    (p0, g0, a0, f0), (p1, g1, a1, f1) = res
    # The two runs differ by atomics-order noise only (most of the time not at all).  Adam normalises: where a gradient element is zero
    # up to that noise (parameters the loss does not depend on) the parameter moves by +lr in one run and -lr in the other, so the
    # parameters are compared where both steps' gradients stand clear of the noise floor, and only bounded elsewhere (2 steps x 2 lr)
    dp = (p0 - p1).double().abs()
    clear = (f0.abs() > 1e-3 * f0.abs().max()) & (g0.abs() > 1e-3 * g0.abs().max())
    assert float(clear.double().mean()) > 0.02, float(clear.double().mean())
    assert float(dp[clear].max()) <= 2e-5, (float(dp[clear].max()), int((dp[clear] > 2e-5).sum()))
    assert float(dp.max()) <= 4.1e-3, float(dp.max())
    assert rel(f0.cpu(), f1.cpu()) < 1e-4 and rel(g0.cpu(), g1.cpu()) < 2e-3, (rel(f0.cpu(), f1.cpu()), rel(g0.cpu(), g1.cpu()))
    assert rel(a0.cpu(), (g0 + a1).cpu()) < 2e-3, rel(a0.cpu(), (g0 + a1).cpu())      # unfused: old gradient still there; fused: started from zero


def test_training_loop_as_benchmarked_converges(dev):
    """the loop bench.py times (next iteration's encoder on its own stream, per-layer AdamW under the backward with the gradient clear
    fused in, dynamic gradient scale) on one small fixed batch: the loss stays finite and falls, parameters stay finite"""
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    wl = bench.TrainWorkload(4, 512, None, 0, dev)
    losses = []
    for i in range(120):
        wl.step()
        if i % 10 == 0 or i == 119:
            torch.cuda.synchronize()                      # the step runs on its own stream
            losses.append(float(wl.last_loss))
    assert all(math.isfinite(v) for v in losses)
    assert sum(losses[-3:]) / 3 < 0.9 * (sum(losses[:3]) / 3), losses
    assert bool(torch.isfinite(wl.engine.flat.params).all())
    assert wl.engine.step_count == 120


def test_dynamic_gradient_scale_follows_the_loss_gradient(golden, weights_sd, dev):
    """grad_scale (the power of two that lifts the backward operands into the fp16 range before their split) tracks max |dLoss/dpred|
    with two backward passes of delay, and a 1e-4 x smaller loss gradient gives gradients as accurate as the full-size one"""
    inp, noise, _ = golden_inputs(golden, dev)
    m = make_module(weights_sd, dev)
    eng = m.train_engine()
    assert eng.grad_scale == 4096.0
    pred, ctx = eng.forward(*inp, seed=3, train=False)
    n = pred.shape[0] * pred.shape[1]
    d = (pred - noise).reshape(n, 7).float().contiguous() * (2.0 / n)
    m.zero_grad(set_to_none=True)
    eng.backward(ctx, d.clone())
    torch.cuda.synchronize()
    g_ref = eng.flat.grads.clone()
    for _ in range(3):                                   # the scale adapts to the small seed gradient
        m.zero_grad(set_to_none=True)
        pred, ctx = eng.forward(*inp, seed=3, train=False)
        eng.backward(ctx, d * 1e-4)
    torch.cuda.synchronize()
    amax = float(d.abs().max()) * 1e-4
    assert eng.grad_scale >= 4096.0 * 16 and 8.0 <= amax * eng.grad_scale < 16.0
    assert rel(eng.flat.grads.cpu() * 1e4, g_ref.cpu()) < 5e-5


def test_overflow_guard_skips_non_finite_gradients_and_backs_the_scale_off(golden, weights_sd, dev):
    """ADVICE r2 (medium), contract as of round 4: a backward whose fp16 gradient planes overflowed hands inf / NaN
    gradients to AdamW.  The guard is PER ELEMENT: an element whose own gradient is non-finite is left untouched (parameter, both
    moments, planes; its gradient is still cleared) and raises the device-side flag; every element with a finite gradient IS
    updated — a partial optimizer step made of valid updates only (the kernel never reads the flag, so replicas decide alike), with
    the step count / bias correction advancing as usual.  This is not the GradScaler's whole-step skip.  The flag reaches the host
    two steps later without a device read in the step and lowers the gradient scale; a real overflow (a seed gradient 2^30 x larger
    than the lagged scale expects) leaves the model finite and training recovers."""
    inp, noise, _ = golden_inputs(golden, dev)
    m = make_module(weights_sd, dev)
    eng = m.train_engine()
    f = eng.flat
    hp = dict(lr=1e-3, weight_decay=1e-2)

    def step(poison=None, scale=1.0):
        f.zero_grad()
        pred, ctx = eng.forward(*inp, seed=3, train=False)
        n = pred.shape[0] * pred.shape[1]
        eng.backward(ctx, ((pred - noise).reshape(n, 7).float() * (2.0 / n * scale)).contiguous())
        if poison is not None:
            poison()
        eng.optimizer_step(**hp)

    step()
    torch.cuda.synchronize()
    p1, m1, v1 = f.params.clone(), f.exp_avg.clone(), f.exp_avg_sq.clone()
    a, b = f.layer_ranges[-1]

    def poison():
        f.grads[5] = float("inf")
        f.grads[a + 7] = float("nan")

    step(poison)
    torch.cuda.synchronize()
    assert torch.isfinite(f.params).all() and torch.isfinite(f.exp_avg).all() and torch.isfinite(f.exp_avg_sq).all()
    assert torch.equal(f.params[5], p1[5]) and torch.equal(f.exp_avg[a + 7], m1[a + 7]) and torch.equal(f.exp_avg_sq[5], v1[5])
    assert torch.equal(f.hi.float() + f.lo.float(), (f.params.half().float() + (f.params - f.params.half().float()).half().float()))
    assert eng.overflow_steps == 0                      # not known on the host yet
    step(); step()
    assert eng.overflow_steps == 1 and eng._backoff == 1.0 / 16
    # a real overflow: gradient planes written with a scale tuned for a 2^30 x smaller seed gradient
    before = f.params.clone()
    step(scale=2.0 ** 30)
    torch.cuda.synchronize()
    assert torch.isfinite(f.params).all() and torch.isfinite(f.exp_avg).all() and torch.isfinite(f.exp_avg_sq).all()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    assert eng.overflow_steps >= 2 and torch.isfinite(f.params).all()
    pred, _ = eng.forward(*inp, seed=3, train=False)
    assert torch.isfinite(pred).all() and not torch.equal(before, f.params)


def test_encoder_train_mode_batchnorm_vs_reference_golden(golden, weights_sd, dev):
    """the frozen encoder in .train() (batch-statistics BatchNorm, running buffers updated) against the reference
    module's outputs and buffers after two passes (tests/golden/encoder_train.npz)"""
    from pfpp_hip import config
    from puzzlefusion_plusplus.vqvae.model.modules.vq_vae import VQVAE

    g = golden("encoder_train")
    pts = T(golden("encoder_float")["pts"]).to(dev)
    enc = VQVAE(config.denoiser_config())
    enc.load_state_dict(weights_sd("vqvae"), strict=True)
    enc = enc.to(dev).train()
    for p in enc.parameters():
        p.requires_grad = False
    out = enc.encode(pts)
    assert np.array_equal(out["xyz"].cpu().numpy(), g["xyz"])
    bad = (np.abs(out["z_q"].cpu().numpy() - g["z_q"]).reshape(-1, 16).max(1) > 1e-4)
    assert not (bad & (g["vq_gap"] > 1e-4)).any()
    assert bad.mean() < 0.01
    enc.encode(pts)                                   # second pass: buffers move again
    sd = enc.state_dict()
    got = np.concatenate([sd[k].flatten().cpu().numpy() for k in g["stat_names"].tolist()])
    assert np.abs(got - g["stats_after_two"]).max() < 2e-5
    assert int(sd["pn2.sa1.mlp_bns.0.num_batches_tracked"]) == int(g["num_batches_tracked"])
    # eval mode afterwards folds the UPDATED running statistics (pack cache invalidated by the train passes)
    enc.eval()
    from oracle import pfpp_oracle as O

    want = O.vqvae_encode({k: v.cpu() for k, v in sd.items()}, pts.cpu())
    e = enc.encode(pts)
    gap = O.vq_gap(sd["vector_quantization.embedding.weight"].cpu(), O.pn2_encode({k: v.cpu() for k, v in sd.items()}, pts.cpu())[0].reshape(-1, 16))
    bad = ((e["z_q"].cpu() - want["z_q"]).abs().reshape(-1, 16).amax(1) > 1e-4)
    assert not (bad & (gap > 1e-4)).any()
    # an unfrozen encoder under autograd is refused loudly (no silent missing gradients)
    enc.train()
    for p in enc.parameters():
        p.requires_grad = True
    with pytest.raises(RuntimeError, match="frozen"):
        enc.encode(pts)


def test_padding_schedule_lists_the_two_half_neighbourhoods_first(dev):
    """pfpp_sa_pad_schedule against numpy on real ball-query output (12 fragments, level-2 shape: 256 points, 128 centroids, radius 0.4,
    64 slots): a permutation of the neighbourhoods, those with more than 32 points in range first, ascending inside each class, the class
    boundary in the last element; and the premise of the skip itself — the slots beyond the in-range count all repeat slot 0."""
    from pfpp_hip import ops

    gen = torch.Generator().manual_seed(3)
    xyz = ((torch.rand(12, 256, 3, generator=gen) * 2 - 1) * torch.tensor([1.0, 0.7, 0.15])).to(dev)      # flat shards: sparse neighbourhoods
    _, new_xyz = ops.fps(xyz, 128)
    ball = ops.ball_query(xyz, new_xyz, 0.4, 64)
    sched = ops.sa_pad_schedule(ball).cpu().numpy()
    idx = ball.cpu().numpy().reshape(-1, 64)
    G = idx.shape[0]
    d2 = ((new_xyz.cpu().numpy().reshape(12, 128, 1, 3) - xyz.cpu().numpy().reshape(12, 1, 256, 3)) ** 2).sum(-1).reshape(G, 256)
    cnt = np.minimum((d2 < np.float32(0.4) ** 2).sum(1), 64)
    two = cnt > 32
    assert 0.1 < two.mean() < 0.9                     # both classes are exercised
    for g in range(G):
        assert (idx[g, cnt[g]:] == idx[g, 0]).all() and len(set(idx[g, :cnt[g]].tolist())) == cnt[g]
    n2 = int(sched[G])
    assert n2 == int(two.sum())
    assert np.array_equal(sched[:n2], np.nonzero(two)[0]) and np.array_equal(sched[n2:G], np.nonzero(~two)[0])
    # live-slot counts, by neighbourhood and in schedule order (stage 2 stores / stage 3 loads only those rows)
    assert np.array_equal(sched[G + 1:2 * G + 1], cnt) and np.array_equal(sched[2 * G + 1:3 * G + 1], cnt[sched[:G]])


@pytest.mark.parametrize("flag", ["SA_TRAIN_CHAIN", "SA_TRAIN_UTAB", "SA_TRAIN_WIDE", "SA_PAD_SKIP"])
def test_encoder_train_chain_equals_layerwise_batchnorm(weights_sd, dev, flag):
    """train-mode set abstraction by recomputation (csrc/sa_train.hip: per-layer chain launches that write only the batch sums,
    level 2's raw second-layer rows and level 1's pooled max / min) against the layer-wise fused-BatchNorm GEMMs on the same
    fragments (F = 12 x N = 1024): same sampling, pre-quantisation features within 2e-5 of their scale, running statistics and
    counters moved identically — the two differ only in the summation order of the fp64 batch sums.
    flag = the switch that is turned off for the comparison run: SA_TRAIN_CHAIN (everything layer-wise), SA_TRAIN_UTAB (first layer of
    levels 2-3 as a grouped convolution instead of the per-point table: U[p] - W_xyz . centroid), SA_TRAIN_WIDE (level 3 layer-wise),
    SA_PAD_SKIP (round 6: every neighbourhood walked as two halves instead of taking the ball query's padding — copies of row 0 — as
    32 y_0 / 32 y_0^2 in the sums)"""
    from pfpp_hip import config, encoder, ops
    from puzzlefusion_plusplus.vqvae.model.modules.vq_vae import VQVAE

    gen = torch.Generator().manual_seed(77)
    pts = (torch.rand(12, 1024, 3, generator=gen) * 2 - 1) * torch.rand(12, 1, 3, generator=gen)
    res = {}
    prev = getattr(encoder, flag)
    prev_min = encoder.SA_PAD_SKIP_MIN
    encoder.SA_PAD_SKIP_MIN = 0                   # 12 fragments are below the size from which the padding schedule is built by default
    try:
        for chain in (False, True):
            setattr(encoder, flag, chain)
            enc = VQVAE(config.denoiser_config())
            enc.load_state_dict(weights_sd("vqvae"), strict=True)
            enc = enc.to(dev).train()
            for p in enc.parameters():
                p.requires_grad = False
            cap = {}
            from pfpp_hip.encoder import pn2_encode

            pk = enc.packed_train()
            z_e, xyz = pn2_encode(pk, pts.to(dev), 25, cap)
            z_e2, _ = pn2_encode(pk, pts.to(dev) * 0.5, 25)              # second pass: the running buffers move again
            torch.cuda.synchronize()
            res[chain] = dict(z_e=z_e.cpu(), z_e2=z_e2.cpu(), xyz=xyz.cpu(), feats={k: v.cpu() for k, v in cap.items() if k.endswith("new_points")},
                              stats={k: v.detach().cpu().clone() for k, v in enc.state_dict().items() if "running" in k or "tracked" in k})
    finally:
        setattr(encoder, flag, prev)
        encoder.SA_PAD_SKIP_MIN = prev_min
    a, b = res[True], res[False]
    assert torch.equal(a["xyz"], b["xyz"])
    for k in b["feats"]:
        assert (a["feats"][k] - b["feats"][k]).abs().max() <= 2e-5 * b["feats"][k].abs().max(), k
    assert (a["z_e"] - b["z_e"]).abs().max() <= 2e-5 * b["z_e"].abs().max()
    assert (a["z_e2"] - b["z_e2"]).abs().max() <= 2e-5 * b["z_e2"].abs().max()
    for k in b["stats"]:
        if "tracked" in k:
            assert int(a["stats"][k]) == int(b["stats"][k]), k
        else:
            assert (a["stats"][k] - b["stats"][k]).abs().max() <= 1e-6 * max(1.0, float(b["stats"][k].abs().max())), k


def test_encoder_train_chain_full_size_properties(weights_sd, dev):
    """BASELINE configs[1] size (154 fragments x 1024 points, 1.26 M rows per level): the recomputing chain launches against the layer-wise
    fused-BatchNorm GEMMs — identical sampling, features within 2e-5 of their scale, the batch statistics the two paths hand to
    BatchNorm (running buffers after one pass) within 1e-6; a second run of the chain path reproduces its own features to 1e-6 (the only
    run-to-run freedom is the order of the fp64 atomics behind the batch sums)."""
    from pfpp_hip import config, encoder, synthetic
    from pfpp_hip.encoder import pn2_encode
    from puzzlefusion_plusplus.vqvae.model.modules.vq_vae import VQVAE

    data = synthetic.make_batch(0, 32, num_points=1024)
    v = data["part_valids"].bool()
    pts = data["part_pcs"][v].contiguous().to(dev)
    assert pts.shape[0] == 154
    res = {}
    prev = encoder.SA_TRAIN_CHAIN
    try:
        for tag, chain in (("layer", False), ("chain", True), ("chain2", True)):
            encoder.SA_TRAIN_CHAIN = chain
            enc = VQVAE(config.denoiser_config())
            enc.load_state_dict(weights_sd("vqvae"), strict=True)
            enc = enc.to(dev).train()
            for p in enc.parameters():
                p.requires_grad = False
            cap = {}
            z_e, xyz = pn2_encode(enc.packed_train(), pts, 25, cap)
            torch.cuda.synchronize()
            res[tag] = dict(z_e=z_e.cpu(), xyz=xyz.cpu(), f2=cap["sa2.new_points"].cpu(), f1=cap["sa1.new_points"].cpu(),
                            stats={k: t.detach().cpu().clone() for k, t in enc.state_dict().items() if "running" in k})
    finally:
        encoder.SA_TRAIN_CHAIN = prev
    a, b, c = res["chain"], res["layer"], res["chain2"]
    assert torch.equal(a["xyz"], b["xyz"])
    for k in ("f1", "f2", "z_e"):
        assert torch.isfinite(a[k]).all()
        assert (a[k] - b[k]).abs().max() <= 2e-5 * b[k].abs().max(), k
        assert (a[k] - c[k]).abs().max() <= 1e-6 * a[k].abs().max(), k
    for k in b["stats"]:
        assert (a["stats"][k] - b["stats"][k]).abs().max() <= 1e-6 * max(1.0, float(b["stats"][k].abs().max())), k


def _ddp_worker(rank, world, port, out_q):
    import os
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for p_ in (str(root), str(root / "puzzlefusion-plusplus_amd")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import torch.distributed as dist

    from oracle import weights
    from pfpp_hip.train import DenoiserTrainEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # gloo moves CUDA tensors through the host: same code path
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)

    class NS_:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer

    m = DenoiserTransformer(NS_(model=NS_(embed_dim=512, out_channels=7, num_layers=6, num_heads=8, num_dim=64, num_point=25)))
    m.load_state_dict(weights.denoiser_state_dict(), strict=True)
    eng = DenoiserTrainEngine(m.to(dev))
    g = np.load(root / "tests" / "golden" / "denoiser.npz")
    t = np.load(root / "tests" / "golden" / "train.npz")
    keys = ("x", "timesteps", "latent", "xyz", "part_valids", "scale", "ref_part")
    inp = [torch.from_numpy(g[k])[rank:rank + 1].to(dev) for k in keys]          # rank r trains on puzzle r
    noise = torch.from_numpy(t["noise"])[rank:rank + 1].to(dev)
    eng.loss_and_grads(*inp, noise, train=False)
    # a second backward on the already all-reduced buffer would reduce the first micro-batch twice: refused loudly
    raised = False
    try:
        eng.loss_and_grads(*inp, noise, train=False)
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    scale = eng.finish_grad_exchange()
    mean = (eng.flat.grads.cpu() * scale).numpy()
    eng.optimizer_step(lr=0.0, weight_decay=0.0)       # closes the step (lr 0: parameters unchanged)
    # gradient accumulation: two micro-batches, the first under no_sync -> one exchange of the accumulated sum
    eng.flat.zero_grad()
    with eng.no_sync():
        eng.loss_and_grads(*inp, noise, train=False)
    eng.loss_and_grads(*inp, noise, train=False)
    scale = eng.finish_grad_exchange()
    accum = (eng.flat.grads.cpu() * scale).numpy()
    if rank == 0:
        out_q.put(dict(mean=mean, accum=accum, raised=raised))
    dist.barrier()
    dist.destroy_process_group()


_LOAD_SCRIPT = """
import sys, time
sys.path[:0] = [{root!r}, {pkg!r}]
import torch
from pfpp_hip import planes as P
dev = torch.device('cuda:0')
a = P.split(torch.randn(4096, 4096, device=dev), 1.0)
w = P.split(torch.randn(4096, 4096, device=dev), 1.0)
out = torch.empty(4096, 4096, device=dev)
print('ready', flush=True)
t_end = time.time() + {seconds}
while time.time() < t_end:
    for _ in range(50):
        P.gemm(a, w, out, M=4096, N=4096, K=4096)          # LDS-bound tiles on every CU
    torch.cuda.synchronize()
"""


def _start_gpu_load(seconds: int = 240):
    """a second PROCESS keeping every CU's LDS pipeline busy (what a second rank / a pytest parent with live engines is to the GPU)"""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    proc = subprocess.Popen([sys.executable, "-c", _LOAD_SCRIPT.format(root=str(root), pkg=str(root / "puzzlefusion-plusplus_amd"), seconds=seconds)],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    assert proc.stdout.readline().strip() == "ready"
    return proc


def test_weight_gradient_gemm_is_exact_next_to_a_co_running_process(dev):
    """VERDICT r4 item 1, the root cause pinned as a test.  Round 4's two-rank tests failed about once in five runs: a 32 x 32 tile of
    ONE weight gradient (always a Linear with a bias: the column-sum instantiation of the dW plane GEMM) off by 1e-4 .. 2e-3 of the
    gradient's max.  Not the exchange, not stream ordering: compiler-inserted copies of fragment registers whose LDS reads (inline asm)
    were still in flight — wrong only when another process's waves keep the CU's LDS pipeline busy (csrc/gemm_pl.hip, 'WAIT
    PLACEMENT').  Here a second process saturates the LDS pipelines while the ff1-shaped weight gradient of the failing tests
    (4096 x 512 over 200 tokens, bias sums on) and the grouped launch run thousands of times: every result is bit-identical to the
    first one and agrees with float64.  Before the fix ~0.3 % of these launches differed."""
    from pfpp_hip import planes as P

    torch.manual_seed(3)
    K, M, N = 200, 4096, 512
    dy = torch.randn(K, M, device=dev) * 1e-3
    x = torch.randn(K, N, device=dev)
    dyp, xp = P.split(dy, 4096.0), P.split(x, 1.0)
    want = dy.double().t() @ x.double()

    def once(grouped):
        gw, gb = torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)
        if grouped:
            P.dw_group([(dyp, xp, gw, gb)], K)
        else:
            P.gemm(dyp, xp, gw, M=M, N=N, K=K, a_kmajor=True, w_kmajor=True, accumulate=True, colsum=gb)
        return gw, gb

    load = _start_gpu_load()
    try:
        for grouped in (False, True):
            gw0, gb0 = once(grouped)
            assert float((gw0.double() - want).abs().max() / want.abs().max()) < 2e-6
            assert float((gb0.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max()) < 2e-6
            for it in range(2500):
                gw, gb = once(grouped)
                assert torch.equal(gw, gw0) and torch.equal(gb, gb0), (grouped, it, float((gw - gw0).abs().max()))
    finally:
        load.kill()
        load.wait()


@pytest.mark.parametrize("armed", [False, True])
def test_per_layer_exchange_schedule_in_one_process_equals_the_six_layer_call(golden, weights_sd, dev, monkeypatch, armed):
    """VERDICT r4 'do this' 1(a): the multi-rank STEP without the collective.  The exchange is forced active in ONE process and replaced
    by a loop-back (every slice copied to pinned host memory and back on a pool stream, the way gloo's CUDA path moves it; 1 / world = 1;
    the table rows 'gathered' from this rank alone), so the code only N > 1 takes — six one-layer pfpp_tlayers_bwd(i, i + 1) calls with
    the AdaLN linears' gradients between them, the all-reduce issued from the third stream behind the main and the weight-gradient
    stream, (armed) the layer's AdamW queued behind it there, the tail's remaining slices — runs exactly as between ranks, next to a
    co-running process, and must reproduce the single-rank six-layer call: gradients to the order of the gradient atomics, and (armed,
    lr > 0) the same parameters after the step."""
    from pfpp_hip import parallel
    from pfpp_hip.train import DenoiserTrainEngine

    inp, noise, _ = golden_inputs(golden, dev)
    hp = dict(lr=1e-3, weight_decay=1e-2)

    def run(eng, n_iter):
        outs = []
        for it in range(n_iter):
            eng.flat.zero_grad()
            if armed:
                eng.arm_optimizer(**hp)
            eng.loss_and_grads(*inp, noise, seed=7, train=True)
            eng.finish_grad_exchange()
            torch.cuda.synchronize()
            outs.append(eng.flat.grads.clone() if not armed else eng.flat.exp_avg.clone())
            eng.optimizer_step(**(hp if armed else dict(lr=0.0, weight_decay=0.0)))
        torch.cuda.synchronize()
        return outs, eng.flat.params.clone()

    ref_out, ref_p = run(DenoiserTrainEngine(make_module(weights_sd, dev)), 3)

    class Handle:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    eng = DenoiserTrainEngine(make_module(weights_sd, dev))
    ex = eng._exchange
    monkeypatch.setattr(parallel.GradExchange, "active", staticmethod(lambda: True))
    monkeypatch.setattr(parallel.GradExchange, "mean_factor", staticmethod(lambda: 1.0))
    pool = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
    state = {"k": 0, "n": 0}

    def reduce_loopback(a, b):
        if b <= a:
            return
        ev = torch.cuda.Event()
        ev.record()
        st = pool[state["k"] % len(pool)]
        state["k"] += 1
        state["n"] += b - a
        st.wait_event(ev)
        with torch.cuda.stream(st):
            host = torch.empty(b - a, dtype=torch.float32, pin_memory=True)
            host.copy_(eng.flat.grads[a:b], non_blocking=True)
            st.synchronize()
            eng.flat.grads[a:b].copy_(host, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        ex._handles.append(Handle(done))

    ex._reduce = reduce_loopback
    ex.gather_rows = lambda rows, index, dim=1: (rows.contiguous(), index.contiguous())
    ex.finish = lambda: ([h.wait() for h in ex._handles], ex._handles.clear(), 1.0)[2]
    load = _start_gpu_load(120)
    try:
        got_out, got_p = run(eng, 3)
    finally:
        load.kill()
        load.wait()
    n_tab = eng.flat.offset["transformer_layers.0.norm1.linear.weight"]
    assert state["n"] == 3 * (eng.flat.numel - n_tab)          # every element outside the tables went through the exchange once per step
    for a, b in zip(got_out, ref_out):
        assert rel(a, b.cpu()) < (2e-5 if armed else 2e-6)
    if armed:
        assert float((got_p - ref_p).abs().max()) <= 3 * 1e-3 * 1.0001 * 2 and float(((got_p - ref_p).abs() > 1e-6).float().mean()) < 1e-3


def test_two_rank_data_parallel_gradients(golden, weights_sd, dev):
    """world_size 2 (both ranks on this GPU, gloo): the per-layer gradient exchange of the engine yields the mean of the two
    ranks' gradients — the N > 1 path of bench.py with the backend swapped"""
    import os

    import torch.multiprocessing as mp

    from pfpp_hip.train import DenoiserTrainEngine

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = q.get(timeout=600)
    got = torch.from_numpy(res["mean"])
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    inp, noise, _ = golden_inputs(golden, dev)
    want = None
    for r in range(2):
        eng = DenoiserTrainEngine(make_module(weights_sd, dev))
        eng.loss_and_grads(*[v[r:r + 1] for v in inp], noise[r:r + 1], train=False)
        want = eng.flat.grads.cpu() if want is None else want + eng.flat.grads.cpu()
        del eng
    assert rel(got, want / 2) < 1e-5
    # accumulation over two identical micro-batches per rank, exchanged once: twice the mean (in-place reduction of an
    # accumulating buffer), and the un-flagged second backward was refused
    assert res["raised"]
    assert rel(torch.from_numpy(res["accum"]), want) < 1e-5


def test_full_size_training_iteration_properties(weights_sd, dev):
    """BASELINE configs[1] size (32 puzzles x 20 slots x 1024 points, 154 valid fragments = 3,850 tokens, the size bench.py
    times): the gradient of the batch loss equals the selected-fragment-weighted sum of the two half-batch gradients (the
    weight-gradient contraction runs over a ragged 3,850 vs 1,900 / 1,950 tokens, different tiles and K splits), is finite,
    and repeats to rounding with the dropouts on"""
    from pfpp_hip import config, synthetic
    from pfpp_hip import train_ops as T
    from pfpp_hip.train import DenoiserTrainEngine
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    torch.manual_seed(0)
    model = Denoiser(config.denoiser_config())
    model.encoder.load_state_dict(weights_sd("vqvae")); model.denoiser.load_state_dict(weights_sd("denoiser"))
    model = model.to(dev)
    model.encoder.eval()
    data = {k: v.to(dev) for k, v in synthetic.make_batch(1234, 32, num_points=1024).items()}
    gt = torch.cat([data["part_trans"], data["part_rots"]], -1).float().contiguous()
    ref = data["ref_part"]
    gen = torch.Generator(device=dev).manual_seed(11)
    noise = torch.randn(gt.shape, device=dev, generator=gen)
    t = torch.randint(0, 1000, (32,), device=dev, generator=gen)
    noisy = model.noise_scheduler.add_noise(gt, noise, t)
    noisy = torch.where(ref.bool().unsqueeze(-1), gt, noisy)
    with torch.no_grad():
        latent, xyz = model._extract_features(data["part_pcs"], data["part_valids"], noisy)
    eng = DenoiserTrainEngine(model.denoiser)

    def grads_of(sl, weight):
        pv = data["part_valids"][sl].contiguous()
        pred, ctx = eng.forward(noisy[sl].contiguous(), t[sl].contiguous(), latent[sl].contiguous(), xyz[sl].contiguous(), pv,
                                data["part_scale"][sl].contiguous(), ref[sl].contiguous(), seed=0, train=False)
        n = pred.shape[0] * pred.shape[1]
        sel = (pv.reshape(n).bool() & ~ref[sl].reshape(n).bool()).to(torch.uint8).contiguous()
        loss, dpred = T.mse_loss(pred.reshape(n, 7), noise[sl].reshape(n, 7).contiguous().float(), sel)
        eng.backward(ctx, dpred * weight)
        return float(loss), int(sel.sum())

    eng.flat.zero_grad()
    loss_full, n_full = grads_of(slice(0, 32), 1.0)
    torch.cuda.synchronize()
    g_full = eng.flat.grads.clone()
    assert torch.isfinite(g_full).all() and g_full.abs().max() > 0
    eng.flat.zero_grad()
    n_a = int((data["part_valids"][:16].bool() & ~ref[:16].bool()).sum())
    n_b = n_full - n_a
    loss_a, _ = grads_of(slice(0, 16), n_a / n_full)
    loss_b, _ = grads_of(slice(16, 32), n_b / n_full)
    torch.cuda.synchronize()
    g_split = eng.flat.grads.clone()
    assert abs((n_a * loss_a + n_b * loss_b) / n_full - loss_full) < 1e-5 * loss_full
    for name in eng.flat.order:
        a, b = eng.flat.view(g_full, name), eng.flat.view(g_split, name)
        assert (a - b).abs().max() <= 2e-4 * max(a.abs().max().item(), 1e-6), name
    # dropouts on: the same seed repeats the iteration to rounding (counter-based masks; only atomic summation order may differ)
    outs = []
    for _ in range(2):
        eng.flat.zero_grad()
        outs.append(float(eng.loss_and_grads(noisy, t, latent, xyz, data["part_valids"], data["part_scale"], ref, noise, seed=5)))
        torch.cuda.synchronize()
        outs.append(eng.flat.grads.clone())
    assert outs[0] == pytest.approx(outs[2], rel=1e-6)
    assert (outs[1] - outs[3]).abs().max() <= 1e-5 * outs[1].abs().max()


def test_training_step_at_the_benchmarked_size_vs_oracle_autograd(weights_sd, dev):
    """VERDICT r5 item 2a.  The training step at the size bench.py times (BASELINE configs[1]: 32 puzzles x 20 slots, 154 valid
    fragments = 3,850 tokens, dropouts ON) against torch autograd through the CPU oracle on the same inputs: the oracle applies the
    very keep masks the kernels generate (counter-based generator read back per site), so the two computations differ by rounding
    only.  At this size the weight gradients are the six-problem grouped launch over 160 workgroups, the forward / input-gradient
    GEMMs run their 3,850-row tiles (the 700-token goldens take other tiles): loss 2e-5, prediction 1e-4 on the selected
    fragments, every parameter's gradient within 2e-4 of its max."""
    from pfpp_hip import config, synthetic
    from pfpp_hip import train_ops as TO
    from pfpp_hip.train import DenoiserTrainEngine
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    torch.manual_seed(0)
    model = Denoiser(config.denoiser_config())
    model.encoder.load_state_dict(weights_sd("vqvae")); model.denoiser.load_state_dict(weights_sd("denoiser"))
    model = model.to(dev)
    model.encoder.eval()
    B, P, L = 32, 20, 25
    data = {k: v.to(dev) for k, v in synthetic.make_batch(0, B, num_points=1024).items()}      # puzzles 0 .. 31: bench.py's batch on rank 0
    gt = torch.cat([data["part_trans"], data["part_rots"]], -1).float().contiguous()
    ref = data["ref_part"]
    gen = torch.Generator(device=dev).manual_seed(11)
    noise = torch.randn(gt.shape, device=dev, generator=gen)
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
    noisy = model.noise_scheduler.add_noise(gt, noise, t)
    noisy = torch.where(ref.bool().unsqueeze(-1), gt, noisy)
    with torch.no_grad():
        latent, xyz = model._extract_features(data["part_pcs"], data["part_valids"], noisy)
    inp = [noisy, t, latent, xyz, data["part_valids"], data["part_scale"], ref]
    inp_c = [v.cpu() for v in inp]
    eng = DenoiserTrainEngine(model.denoiser)
    seed = 20260930
    valid = inp_c[4].reshape(-1).bool()
    slot = torch.nonzero(valid).flatten()
    Fv = slot.numel()
    assert Fv * L == 3850, Fv

    def drop(site, tensor):
        width = tensor.shape[-1]
        p = eng.p_token if site == 0 else eng.p_layer
        keep_c = TO.dropout_mask(Fv * L * width, p, seed, site, dev).view(Fv, L, width).cpu().to(tensor.dtype)
        keep = torch.ones(B * P, L, width, dtype=tensor.dtype)
        keep[slot] = keep_c                                   # compact row order = ascending valid slots
        return tensor * keep.view(B, P * L, width) / (1 - p)

    sd, pred_o, loss_o = _oracle_grads(weights_sd, inp_c, noise.cpu(), drop)
    pred, ctx = eng.forward(*inp, seed=seed, train=True)
    sel = valid.view(B, P)
    assert (pred.cpu() - pred_o)[sel].abs().max() < 1e-4
    eng.flat.zero_grad()
    loss = eng.loss_and_grads(*inp, noise, seed=seed, train=True)
    assert abs(float(loss) - float(loss_o)) < 2e-5 * float(loss_o)
    named = dict(eng.module.named_parameters())
    errs = {n: rel(named[n].grad, sd[n].grad) for n in named}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-4, (worst, errs[worst])


@pytest.mark.parametrize("variant", [0, 3])      # 0 = the library's 256 x 128 tile, 3 = 128 x 128
def test_grouped_weight_gradient_launch_at_3850_rows_vs_float64(dev, variant):
    """VERDICT r5 item 2b.  pfpp_gemm_dw_group with a transformer block's six REAL problems over the benchmarked 3,850 token rows
    (qkv x 2: 1536 x 512, out-projection x 2: 512 x 512, ff1: 4096 x 512, ff2: 512 x 2048; 160 output tiles of 256 x 128 dealt to the
    XCDs as one concatenated list - where an indexing slip between problems would hide) against float64 (2e-6 of each gradient's
    max, bias sums too), accumulating onto a non-zero gradient, and against the six single launches (association of the partial
    sums only: 2e-6)."""
    from pfpp_hip import planes as P

    torch.manual_seed(17)
    K = 3850
    shapes = [(1536, 512, True), (512, 512, True), (1536, 512, True), (512, 512, True), (4096, 512, True), (512, 2048, True)]
    jobs, singles, wants = [], [], []
    for M, N, bias in shapes:
        dy = torch.randn(K, M, device=dev) * 1e-3
        x = torch.randn(K, N, device=dev)
        g0 = torch.randn(M, N, device=dev) * 1e-2            # the gradient buffer already holds something (accumulation)
        b0 = torch.randn(M, device=dev) * 1e-2
        dyp, xp = P.split(dy, 4096.0), P.split(x, 1.0)
        gw, gb = g0.clone(), b0.clone()
        jobs.append((dyp, xp, gw, gb if bias else None))
        singles.append((dyp, xp, g0.clone(), b0.clone(), M, N))
        wants.append((g0.double() + dy.double().t() @ x.double(), b0.double() + dy.double().sum(0)))
    P.dw_group(jobs, K, variant)
    for dyp, xp, gw1, gb1, M, N in singles:
        P.gemm(dyp, xp, gw1, M=M, N=N, K=K, a_kmajor=True, w_kmajor=True, accumulate=True, colsum=gb1)
    torch.cuda.synchronize()
    for i, ((_, _, gw, gb), (_, _, gw1, gb1, _, _), (ww, wb)) in enumerate(zip(jobs, singles, wants)):
        sw, sb = float(ww.abs().max()), float(wb.abs().max())
        assert float((gw.double() - ww).abs().max()) < 2e-6 * sw, (i, "dW vs float64")
        assert float((gb.double() - wb).abs().max()) < 2e-6 * sb, (i, "db vs float64")
        assert float((gw - gw1).abs().max()) < 2e-6 * sw, (i, "grouped vs single launch")
        assert float((gb - gb1).abs().max()) < 2e-6 * sb, (i, "grouped vs single launch (bias)")


def test_feature_pipeline_equals_inline_encoder(weights_sd, dev):
    """the encoder of iteration i+1 issued on its own stream during iteration i yields the same features, losses and
    encoder buffers as running it in line (same (noise, t) draws)"""
    from pfpp_hip import config, synthetic
    from pfpp_hip.train import DenoiserTrainEngine, FeaturePipeline
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    data = {k: v.to(dev) for k, v in synthetic.make_batch(70, 4, num_points=512).items()}
    gt = torch.cat([data["part_trans"], data["part_rots"]], -1).float().contiguous()
    ref = data["ref_part"]
    losses, stats = [], []
    for use_pipe in (False, True):
        torch.manual_seed(0)
        model = Denoiser(config.denoiser_config())
        model.encoder.load_state_dict(weights_sd("vqvae")); model.denoiser.load_state_dict(weights_sd("denoiser"))
        model = model.to(dev).train()
        for p_ in model.encoder.parameters():
            p_.requires_grad = False
        eng = DenoiserTrainEngine(model.denoiser)
        gen = torch.Generator(device=dev).manual_seed(3)

        def draw():
            return (torch.randn(gt.shape, device=dev, generator=gen), torch.randint(0, 1000, (4,), device=dev, generator=gen))

        pipe = FeaturePipeline(model, dev) if use_pipe else None
        ls = []
        for i in range(3):
            if pipe is not None:
                f = pipe.next(data, gt, ref, draw)
                noisy, t, latent, xyz, noise = f["noisy"], f["t"], f["latent"], f["xyz"], f["noise"]
            else:
                noise, t = draw()
                noisy = model.noise_scheduler.add_noise(gt, noise, t)
                noisy[ref] = gt[ref]
                with torch.no_grad():
                    latent, xyz = model._extract_features(data["part_pcs"], data["part_valids"], noisy)
            eng.flat.zero_grad()
            ls.append(float(eng.loss_and_grads(noisy, t, latent, xyz, data["part_valids"], data["part_scale"], ref, noise, seed=i)))
            eng.optimizer_step()
        torch.cuda.synchronize()
        losses.append(ls)
        stats.append(model.encoder.state_dict()["pn2.sa2.mlp_bns.1.running_var"].clone())
    assert losses[0] == pytest.approx(losses[1], rel=1e-5)
    # the pipelined run has issued one extra encoder pass (the prefetched 4th batch): compare after the same number of passes is
    # not possible from outside, so only require the buffers to have moved consistently (same direction, similar size)
    assert torch.isfinite(stats[1]).all() and (stats[0] - stats[1]).abs().max() < 0.5 * stats[0].abs().max()


@pytest.mark.gpu
def test_host_side_layout_and_no_device_read_in_the_step(weights_sd, dev):
    """the valid-fragment layout built on the host by the batch-transfer hooks equals the one read back from the
    device, a fresh part_valids tensor gets a fresh layout, and a training iteration on a batch whose layout is attached
    performs no synchronising device->host read (the step can be enqueued ahead of the GPU)"""
    import warnings

    from pfpp_hip import config, synthetic
    from pfpp_hip.denoiser import CompactLayout, layout_of
    from pfpp_hip.train import DenoiserTrainEngine
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    torch.manual_seed(0)
    model = Denoiser(config.denoiser_config())
    model.encoder.load_state_dict(weights_sd("vqvae")); model.denoiser.load_state_dict(weights_sd("denoiser"))
    model = model.to(dev).train()
    for p_ in model.encoder.parameters():
        p_.requires_grad = False
    host = synthetic.make_batch(71, 3, num_points=512)
    batch = model.on_before_batch_transfer(dict(host))
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}      # what Lightning's transfer does
    batch = model.on_after_batch_transfer(batch)
    assert "_pfpp_valids_host" not in batch
    lay = layout_of(batch["part_valids"], 25)
    want = CompactLayout(batch["part_valids"].clone(), 25)
    for name in ("slot", "slot32", "frag_b", "frag_p", "seq_len", "seq_off"):
        assert torch.equal(getattr(lay, name), getattr(want, name)), name
    assert (lay.Fv, lay.max_len) == (want.Fv, want.max_len)
    # a different batch -> a different layout; an in-place write to the same tensor invalidates the remembered one
    other = synthetic.make_batch(72, 3, num_points=512, num_parts=5)["part_valids"].to(dev)
    assert layout_of(other, 25).Fv == 15
    other[0, 4] = 0
    assert layout_of(other, 25).Fv == 14

    eng = DenoiserTrainEngine(model.denoiser)
    gt = torch.cat([batch["part_trans"], batch["part_rots"]], -1).float().contiguous()
    noise = torch.randn_like(gt)
    t = torch.randint(0, 1000, (3,), device=dev)

    def load(seed):
        b_ = model.on_before_batch_transfer(dict(synthetic.make_batch(seed, 3, num_points=512)))
        b_ = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b_.items()}
        return model.on_after_batch_transfer(b_)

    def iteration(b_, seed):
        gt_ = torch.cat([b_["part_trans"], b_["part_rots"]], -1).float().contiguous()
        noisy = torch.where(b_["ref_part"].unsqueeze(-1), gt_, model.noise_scheduler.add_noise(gt_, noise, t))
        with torch.no_grad():
            latent, xyz = model._extract_features(b_["part_pcs"], b_["part_valids"], noisy)
        eng.flat.zero_grad()
        loss_ = eng.loss_and_grads(noisy, t, latent, xyz, b_["part_valids"], b_["part_scale"], b_["ref_part"], noise, seed=seed)
        eng.optimizer_step()
        return loss_

    iteration(batch, 0)                       # first call: packing, workspaces
    fresh = load(73)                          # a NEW batch through the hooks (the H2D copies are outside the checked region)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            loss = iteration(fresh, 1)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    syncs = [f"{x.filename}:{x.lineno} {x.message}" for x in w if "synchroniz" in str(x.message)]
    assert not syncs, syncs
    assert torch.isfinite(loss).all()


@pytest.mark.gpu
def test_module_surface_runs_the_benchmarked_schedule(dev):
    """the schedule bench.py times (loop body on a high-priority stream, next batch's frozen encoder one iteration
    ahead on the CU-masked stream, AdamW per layer under the backward with the gradient clear) is what a plain loop over the
    MODULE surface gets — `for batch in model.training_schedule(loader): training_step / backward / optimizer.step / zero_grad`,
    called from the default stream — within 12 % / 0.7 ms of the engine-level iteration of bench.TrainWorkload at BASELINE configs[1]'s
    size (best of three / five windows).  The module loop issues the same kernels for the same kernel time (profiles/r03o_*); what separates
    the two is host time (autograd hop, optimizer wrapper, logging) and the encoder's stream (a loop on the default stream cannot use
    the CU-masked one): 2 to 6 % in a fresh process.
    Both loops are timed in a FRESH process (tools/diag/module_vs_engine.py), like a training script: inside this suite's process the
    same measurement depends on how many HIP streams the ~250 tests before it have created — ROCm spreads streams over
    GPU_MAX_HW_QUEUES hardware queues, with the default 4 two of the schedule's streams shared one (10.8 ms per module iteration),
    with 8 the suite's leftovers still cost the module loop ~0.5 ms, with 24 both loops ran at 18 - 21 ms (round 6,
    profiles/r06e_ab_tables_active_and_hw_queues.txt).
    In this process: the features the pipeline hands over are the ones the in-line path computes (same draw -> same loss)."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    run = subprocess.run([sys.executable, str(root / "tools" / "diag" / "module_vs_engine.py")], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    t_engine, t_module = rec["engine_ms"] * 1e-3, rec["module_ms"] * 1e-3
    print(f"engine-level iteration {t_engine * 1e3:.3f} ms, module-surface iteration {t_module * 1e3:.3f} ms")
    assert rec["losses_finite"] and rec["loss_last5"] < rec["loss_first5"]          # it trains
    # the module loop pays a fixed 0.1 - 0.55 ms of host work per iteration on top of the engine loop (autograd hop, optimizer wrapper,
    # logging, two host threads); against round 5's 6.2 ms engine iteration that was 4.8 - 5.5 %, against round 6's 5.75 - 5.9 ms it
    # is 1.5 - 9.4 % depending on the box (five fresh-process runs, profiles/r06e_ab_tables_active_and_hw_queues.txt): the bar follows
    # the faster engine iteration, the absolute gap is bounded next to it
    assert t_module <= 1.12 * t_engine and t_module - t_engine <= 0.7e-3, (t_module, t_engine)

    from pfpp_hip import config, synthetic
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    torch.manual_seed(1234)
    model = Denoiser(config.denoiser_config()).to(dev)
    with torch.no_grad():
        model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
    for p_ in model.encoder.parameters():
        p_.requires_grad = False
    model.train()
    data = {k: v.to(dev) for k, v in synthetic.make_batch(0, 32, num_points=1024).items()}
    # same draw -> the prefetched features equal the in-line ones: forward with injected (noise, t) == forward through the schedule
    from pfpp_hip.train import FeaturePipeline

    torch.manual_seed(7)
    gt = torch.cat([data["part_trans"], data["part_rots"]], dim=-1).float().contiguous()
    noise = torch.randn(gt.shape, device=dev)
    t = torch.randint(0, 1000, (32,), device=dev).long()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        pipe = FeaturePipeline(model, dev)
        f = pipe._issue(data, gt, data["ref_part"], noise, t)
    torch.cuda.synchronize()
    with torch.no_grad():
        noisy = model.noise_scheduler.add_noise(gt, noise, t)
        noisy = torch.where(data["ref_part"].bool().unsqueeze(-1), gt, noisy)
        lat, xyz = model._extract_features(data["part_pcs"], data["part_valids"], noisy)
    torch.cuda.synchronize()
    # train-mode BatchNorm uses the batch's own statistics: the two passes differ only by the order of the fp64 atomics behind them,
    # which may flip a VQ code at a near-tie
    assert torch.equal(f["xyz"], xyz) and float((f["latent"] != lat).float().mean()) < 0.01


def test_bench_multi_rank_path_with_gloo(dev):
    """bench.py under torch.distributed.run with 2 ranks (both on this GPU, PFPP_BENCH_BACKEND=gloo): the N > 1 branch —
    barriers, max-over-ranks clock, per-layer gradient exchange inside the timed step, one JSON line from rank 0"""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PFPP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--points", "256", "--no-cpu-baseline", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["puzzles_per_gpu"] == 4


@pytest.mark.gpu
def test_bench_gpus_flag_spawns_its_own_ranks(dev):
    """`python bench.py --gpus 2` with no launcher and WORLD_SIZE unset must start 2 ranks itself (scripts/train_denoiser.sh:6-7
    `+trainer.devices=N +trainer.strategy=ddp`) and print ONE line with n_gpus == 2 whose rank count comes out of a collective;
    with the RCCL backend and fewer than N devices it must refuse instead of silently running one rank."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--points", "256", "--no-cpu-baseline", "--no-roofline"]
    out = subprocess.run([sys.executable, str(root / "bench.py"), *args], capture_output=True, text=True, timeout=600,
                         env=dict(env, PFPP_BENCH_BACKEND="gloo"), cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["extra"]["gloo_ranks"] == 2 and len(d["extra"]["rank_devices"]) == 2
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, str(root / "bench.py"), *args], capture_output=True, text=True, timeout=600,
                             env=dict(env, PFPP_BENCH_BACKEND="nccl"), cwd=str(root))
        assert out.returncode != 0 and "GPU(s) visible" in out.stderr and not out.stdout.strip()


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism THROUGH THE MODULE SURFACE (VERDICT r3 star row; scripts/train_denoiser.sh:6-7: +trainer.devices=4 +trainer.strategy=ddp)
# ---------------------------------------------------------------------------------------------------------------------
def _surface_paths():
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for p_ in (str(root), str(root / "puzzlefusion-plusplus_amd")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    return root


class _PuzzleList(torch.utils.data.Dataset):
    """synthetic puzzles as a map-style dataset (what GeometryLatentDataset is to the loader)"""

    def __init__(self, ids, num_points=512, num_parts=4):
        self.ids, self.num_points, self.num_parts = list(ids), num_points, num_parts

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        from pfpp_hip import synthetic

        return synthetic.make_puzzle(self.ids[i], self.num_points, 20, self.num_parts, None)


def _surface_model(dev, seed):
    """a Denoiser whose frozen encoder stays in eval mode (batch-statistics BatchNorm sums with fp64 atomics: run-to-run VQ code flips
    would blur the comparison of two runs; everything else is the default training path)"""
    from oracle import weights
    from pfpp_hip import config
    from puzzlefusion_plusplus.denoiser.model.denoiser import Denoiser

    class DetDenoiser(Denoiser):
        def train(self, mode=True):
            super().train(mode)
            self.encoder.eval()
            return self

    torch.manual_seed(seed)
    model = DetDenoiser(config.denoiser_config())
    model.encoder.load_state_dict(weights.vqvae_state_dict())
    model.denoiser.load_state_dict(weights.denoiser_state_dict())
    for p_ in model.encoder.parameters():
        p_.requires_grad = False
    return model.to(dev)


def _surface_fit(model, ids, devices, out_dir, **kw):
    from torch.utils.data import DataLoader

    from pfpp_hip.launch import Trainer

    loader = DataLoader(_PuzzleList(ids), batch_size=2, shuffle=False, drop_last=True)
    tr = Trainer(devices=devices, strategy="ddp" if devices > 1 else "auto", max_epochs=1, check_val_every_n_epoch=0,
                 default_root_dir=str(out_dir), **kw)
    tr.fit(model, loader)
    return tr


def _surface_worker(rank, world, port, out_dir, steps, accumulate):
    import os

    _surface_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), RANK=str(rank), WORLD_SIZE=str(world),
                      PFPP_DDP_BACKEND="gloo")        # both ranks on this GPU: gloo moves the CUDA tensors through the host, same code path
    dev = torch.device("cuda:0")
    model = _surface_model(dev, 100 + rank)
    tr = _surface_fit(model, range(8), world, f"{out_dir}/rank{rank}", max_steps=steps, accumulate_grad_batches=accumulate)
    f = model.denoiser.train_engine().flat
    torch.cuda.synchronize()
    torch.save(dict(exp_avg=f.exp_avg.cpu(), params=f.params.cpu(), steps=tr.global_step, world=tr.world_size), f"{out_dir}/rank{rank}.pt")


@pytest.mark.parametrize("accumulate", [1, 2])
def test_two_rank_training_through_the_module_surface(dev, tmp_path, accumulate):
    """`Trainer(devices=2, strategy="ddp").fit(model, loader)` — one process per rank, DistributedSampler, training_schedule ->
    training_step -> backward -> FusedAdamW.step, gradients exchanged per layer under the backward with the layer's AdamW queued behind
    its all-reduce (accumulate 1), or accumulated under engine.no_sync() and exchanged once (accumulate 2): after one optimizer step
    the first Adam moment (0.05 x the averaged gradient) equals the mean of what the two ranks' shards give alone, and both replicas
    hold bit-identical parameters."""
    import torch.multiprocessing as mp

    _surface_paths()
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_surface_worker, args=(r, 2, port, str(tmp_path), 1, accumulate)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(timeout=900)
        assert p_.exitcode == 0
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    assert got[0]["steps"] == 1 and got[0]["world"] == 2
    assert torch.equal(got[0]["params"], got[1]["params"]) and torch.equal(got[0]["exp_avg"], got[1]["exp_avg"])
    assert torch.isfinite(got[0]["params"]).all()
    ck = torch.load(tmp_path / "rank0" / "last.ckpt", weights_only=False)           # Lightning's layout, written by rank 0 only
    assert {"state_dict", "optimizer_states", "epoch", "global_step"} <= set(ck) and not (tmp_path / "rank1" / "last.ckpt").exists()
    assert any(k.startswith("denoiser.transformer_layers.0.") for k in ck["state_dict"]) and any(k.startswith("encoder.") for k in ck["state_dict"])
    # what each rank's shard gives alone (same seed -> same noise / timestep / dropout draws; rank r reads puzzles r, r + 2, ...)
    want = None
    for r in range(2):
        model = _surface_model(dev, 100 + r)
        ids = list(range(r, 8, 2))
        _surface_fit(model, ids, 1, tmp_path / f"alone{r}", max_steps=1, accumulate_grad_batches=accumulate)
        torch.cuda.synchronize()
        m = model.denoiser.train_engine().flat.exp_avg.cpu()
        want = m if want is None else want + m
        del model
    assert rel(got[0]["exp_avg"], want / 2) < 2e-5


def _poison_worker(rank, world, port, out_q):
    import os

    import torch.distributed as dist

    _surface_paths()
    from oracle import weights
    from pfpp_hip.train import DenoiserTrainEngine
    from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    root = _surface_paths()
    m = DenoiserTransformer(NS(model=NS(embed_dim=512, out_channels=7, num_layers=6, num_heads=8, num_dim=64, num_point=25)))
    m.load_state_dict(weights.denoiser_state_dict(), strict=True)
    eng = DenoiserTrainEngine(m.to(dev))
    f = eng.flat
    g, t = np.load(root / "tests" / "golden" / "denoiser.npz"), np.load(root / "tests" / "golden" / "train.npz")
    keys = ("x", "timesteps", "latent", "xyz", "part_valids", "scale", "ref_part")
    inp = [torch.from_numpy(g[k])[rank:rank + 1].to(dev) for k in keys]
    noise = torch.from_numpy(t["noise"])[rank:rank + 1].to(dev)
    hp = dict(lr=1e-3, weight_decay=1e-2)
    a, b = f.layer_ranges[3]
    tail = f.offset["mlp_out_rot.2.bias"] + 3         # a dense (all-reduced) element outside the layers; the AdaLN tables travel as rows
    snap = None
    for step in range(3):
        eng.arm_optimizer(zero_grad=True, **hp)        # per-layer AdamW behind each layer's all-reduce
        if step == 1:
            torch.cuda.synchronize()
            snap = (f.params[a + 7].item(), f.exp_avg[a + 7].item(), f.params[tail].item())
            if rank == 0:                              # ONE rank overflows: the all-reduce hands inf / NaN to both
                f.grads[a + 7] = float("inf")
                f.grads[tail] = float("nan")
        eng.loss_and_grads(*inp, noise, train=False)
        eng.optimizer_step(zero_grad=True, **hp)
        if step == 1:
            torch.cuda.synchronize()
            kept = (f.params[a + 7].item(), f.exp_avg[a + 7].item(), f.params[tail].item()) == snap
            flagged = int(eng._overflow.sum().item()) == 0          # cleared for the next step after the copy to the host was queued
    torch.cuda.synchronize()
    # numpy: pickled by value (a torch tensor travels as a file descriptor that dies with this process)
    out_q.put(dict(rank=rank, params=f.params.cpu().numpy(), m=f.exp_avg.cpu().numpy(), v=f.exp_avg_sq.cpu().numpy(),
                   hi=f.hi.float().cpu().numpy(), kept=kept, flagged=flagged, overflow_steps=eng.overflow_steps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_stay_equal_after_an_overflow(dev):
    """the guarded AdamW decides PER ELEMENT from that element's own all-reduced gradient, never from a flag other
    workgroups of the launch are still writing — so when one rank's backward overflows, both replicas skip exactly the same elements
    (parameters and moments there untouched) and remain bit-identical; with the optimizer armed, i.e. on the multi-rank form of the
    benchmarked schedule (layer updates queued behind the layer's all-reduce)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_poison_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda d: d["rank"])
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    for k in ("params", "m", "v", "hi"):
        assert np.array_equal(res[0][k], res[1][k]), k
        assert np.isfinite(res[0][k]).all(), k
    assert res[0]["kept"] and res[1]["kept"] and res[0]["flagged"]


@pytest.mark.parametrize("wd", ["0", "1"])
@pytest.mark.parametrize("train,armed", [(True, False), (True, True), (False, False)])
def test_blocks_sequenced_from_c_equal_the_python_sequence(golden, weights_sd, dev, train, armed, wd, monkeypatch):
    """VERDICT r3 'do this' 3: pfpp_tlayers_fwd / pfpp_tlayers_bwd (csrc/tlayer.hip) enqueue the six blocks' launches from C — the same
    launches with the same arguments as the Python sequence (PFPP_TRAIN_CSEQ=0 / engine._cseq = False), so the prediction is
    bit-identical, the gradients agree to the order of the LayerNorm / AdaLN gradient atomics, and an armed step (AdamW per layer on the
    side stream, queued by the C sequencer) leaves the same parameters."""
    from pfpp_hip.train import DenoiserTrainEngine

    # wd = "1" (default): the C sequencer runs the qkv / out / second feed-forward linears and their input gradients through pfpp_gemm_wd
    # (weights blocked per layer where they are read).  One k-ordered chain per output there; at this test's few tokens the tiled kernel
    # splits K over workgroups, so the two agree to fp32 rounding — bit-identity at the sizes where the tiled kernel does not split is
    # test_gemm_wd_bit_identical_to_the_tiled_gemm's.  wd = "0": the same launches as the Python sequence, bit-identical.
    # the block's weight gradients: wd = "0" keeps one pfpp_gemm_planes launch per weight (what the Python sequence issues), wd = "1"
    # takes the default — ONE pfpp_gemm_dw_group launch per block, each output tile over the whole contraction (round 5)
    monkeypatch.setenv("PFPP_TRAIN_WD", wd)
    monkeypatch.setenv("PFPP_TRAIN_DW_GROUP", "0" if wd == "0" else "1")
    inp, noise, _ = golden_inputs(golden, dev)
    hp = dict(lr=1e-3, weight_decay=1e-2)
    out = []
    for cseq in (False, True):
        eng = DenoiserTrainEngine(make_module(weights_sd, dev))
        eng._cseq = cseq
        for step in range(2):
            eng.flat.zero_grad()
            if armed:
                eng.arm_optimizer(**hp)
            pred, ctx = eng.forward(*inp, seed=11 + step, train=train)
            assert (ctx.t.get("cseq") is not None) == cseq
            if step == 0:
                first = pred.clone()                 # before any update: no gradient atomics behind it yet
            n = pred.shape[0] * pred.shape[1]
            dpred = ((pred - noise).reshape(n, 7).float() * (2.0 / n)).contiguous()
            eng.backward(ctx, dpred)
            torch.cuda.synchronize()
            if step == 0:
                grads0 = eng.flat.grads.clone()          # same weights on both sides: only the gradient atomics' order differs
            grads = eng.flat.grads.clone()
            eng.optimizer_step(**hp)
        torch.cuda.synchronize()
        out.append((first, pred.clone(), grads, eng.flat.params.clone(), eng.flat.exp_avg.clone(), grads0))
        del eng
    (f0, p0, g0, w0, m0, s0), (f1, p1, g1, w1, m1, s1) = out
    assert torch.equal(f0, f1) if wd == "0" else rel(f1, f0.cpu()) < 2e-6
    assert rel(p1, p0.cpu()) < 1e-4
    tol = 2e-6 if wd == "0" else 2e-5          # wd: one chain per output against the tiled kernel's K-split partial sums at these few tokens
    assert rel(s1, s0.cpu()) < tol
    # the second step runs on weights that already differ where Adam's sign-like first update met a noise-level gradient element: its
    # gradients agree to a few 1e-6 (2.0 - 2.5e-6 seen on one box in 4 of 22 whole-suite runs, profiles/r05x_suite_loop_after_fix_2.txt)
    assert rel(g1, g0.cpu()) < 5 * tol and rel(m1, m0.cpu()) < 5 * tol
    assert float((w1 - w0).abs().max()) <= 2e-3 * 1.0001 * 2        # Adam moves an element by at most ~lr per step; sign flips only at noise-level gradients
    moved = float(((w1 - w0).abs() > 1e-6).float().mean())
    print(f"parameters that differ by more than 1e-6 after two steps: {moved:.3e}")
    assert moved < 1e-3


def test_second_step_gradient_slack_is_adam_sign_flips_at_noise_level_gradients(golden, weights_sd, dev, monkeypatch):
    """VERDICT r5 weak 2, the measurement behind the 5 x tolerance of the test above.  Python-sequenced and C-sequenced engines side by
    side: after ONE optimizer step their parameters differ only where the first step's gradient element is noise (Adam's first update
    is lr * sign(g): an element whose gradient is a rounding residue around zero can take opposite signs on the two sides) — measured
    here as: every parameter that differs lies where |g| < 1e-4 of its tensor's largest gradient.  And when the second step starts
    from IDENTICAL weights and moments (copied across), its gradients agree to the first step's tolerance again — so the slack of
    the trajectory test is the weight difference, not the sequencer."""
    from pfpp_hip.train import DenoiserTrainEngine

    monkeypatch.setenv("PFPP_TRAIN_WD", "0")
    monkeypatch.setenv("PFPP_TRAIN_DW_GROUP", "0")
    inp, noise, _ = golden_inputs(golden, dev)
    hp = dict(lr=1e-3, weight_decay=1e-2)
    engs = []
    for cseq in (False, True):
        e = DenoiserTrainEngine(make_module(weights_sd, dev))
        e._cseq = cseq
        engs.append(e)

    def step(e, seed):
        e.flat.zero_grad()
        pred, ctx = e.forward(*inp, seed=seed, train=True)
        n = pred.shape[0] * pred.shape[1]
        e.backward(ctx, ((pred - noise).reshape(n, 7).float() * (2.0 / n)).contiguous())
        torch.cuda.synchronize()
        return e.flat.grads.clone()

    g_py, g_c = step(engs[0], 11), step(engs[1], 11)
    assert rel(g_c, g_py.cpu()) < 2e-6
    for e in engs:
        e.optimizer_step(**hp)
    torch.cuda.synchronize()
    w_py, w_c = engs[0].flat.params, engs[1].flat.params
    differs = (w_py - w_c).abs() > 1e-6
    n_diff = int(differs.sum())
    worst = 0.0
    f = engs[0].flat
    for name in f.order:                                        # gradient of the differing elements relative to their tensor's largest
        d = f.view(differs, name)
        if bool(d.any()):
            g = f.view(g_py, name)
            worst = max(worst, float(g[d].abs().max() / g.abs().max()))
    print(f"{n_diff} of {w_py.numel()} parameters differ after one step; largest |g| / max|g| among them: {worst:.2e}")
    assert n_diff < 1e-3 * w_py.numel() and worst < 1e-4
    # the second step from identical state: the C sequencer's gradients are back inside the first step's tolerance
    with torch.no_grad():
        for a, b in ((engs[1].flat.params, engs[0].flat.params), (engs[1].flat.exp_avg, engs[0].flat.exp_avg),
                     (engs[1].flat.exp_avg_sq, engs[0].flat.exp_avg_sq)):
            a.copy_(b)
    engs[1].flat.refresh_planes()
    engs[0].flat.refresh_planes()
    g_py2, g_c2 = step(engs[0], 12), step(engs[1], 12)
    assert rel(g_c2, g_py2.cpu()) < 2e-6


@pytest.mark.parametrize("train", [True, False])
def test_fused_embedding_and_adaln_ends_equal_the_layerwise_path(golden, weights_sd, dev, train, monkeypatch):
    """VERDICT r4 'do this' 5: the training step's skinny ends — token embedding forward in one launch on the packed [W_shape | W_param]
    planes, its five gradients from one contraction over the tokens (csrc/embed_train.hip), the AdaLN linears' gradients and
    d/d(embedded timestep) in two fp32 launches (csrc/ada_bwd.hip) — against the launches they replace (PFPP_TRAIN_EMBED_FWD_FUSED /
    _BWD_FUSED / PFPP_TRAIN_ADA_BWD_FUSED = 0: features + two skinny GEMMs + combine; two weight-gradient GEMMs, two adds, two column
    sums, the per-fragment token sum; a column sum + two tiled gradient GEMMs): two steps, the second on updated weights (re-packed)."""
    from pfpp_hip.train import DenoiserTrainEngine

    inp, noise, _ = golden_inputs(golden, dev)
    hp = dict(lr=1e-3, weight_decay=1e-2)
    out = []
    for fused in ("0", "1"):
        for k in ("PFPP_TRAIN_EMBED_FWD_FUSED", "PFPP_TRAIN_EMBED_BWD_FUSED", "PFPP_TRAIN_ADA_BWD_FUSED"):
            monkeypatch.setenv(k, fused)
        eng = DenoiserTrainEngine(make_module(weights_sd, dev))
        assert eng._embed_fwd_fused == (fused == "1") and eng._embed_bwd_fused == (fused == "1") and eng._ada_bwd_fused == (fused == "1")
        for step in range(2):
            eng.flat.zero_grad()
            pred, ctx = eng.forward(*inp, seed=3 + step, train=train)
            assert (ctx.t["ft"] is not None) == (fused == "1") and (ctx.t["sf"] is None) == (fused == "1")
            if step == 0:
                first = pred.clone()
            n = pred.shape[0] * pred.shape[1]
            dpred = ((pred - noise).reshape(n, 7).float() * (2.0 / n)).contiguous()
            eng.backward(ctx, dpred)
            torch.cuda.synchronize()
            if step == 0:
                grads = eng.flat.grads.clone()           # same weights on both sides: only the kernels differ
            last = eng.flat.grads.clone()
            eng.optimizer_step(**hp)
        torch.cuda.synchronize()
        out.append((first, pred.clone(), grads, eng.flat, last))
    (f0, p0, g0, fl0, l0), (f1, p1, g1, fl1, l1) = out
    assert rel(f1, f0.cpu()) < 2e-6 and rel(p1, p0.cpu()) < 1e-4
    assert rel(g1, g0.cpu()) < 2e-5 and rel(l1, l0.cpu()) < 2e-4       # second step: on weights that already differ by Adam's rounding
    # the tensors the new kernels write, one by one (each against its own magnitude)
    for name in ("shape_embedding.weight", "shape_embedding.bias", "param_fc.weight", "param_fc.bias", "ref_part_emb.weight",
                 "transformer_layers.0.norm1.linear.weight", "transformer_layers.5.norm2.linear.bias", "transformer_layers.3.norm1.emb.weight"):
        a, b = fl1.view(g1, name), fl0.view(g0, name)
        assert rel(a, b.cpu()) < 3e-5, name
