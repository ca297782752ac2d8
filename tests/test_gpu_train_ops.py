"""GPU (-m gpu): the training kernels (include/pfpp.h section a17), each against torch autograd of the
same operation evaluated in fp32 (fp64 where noted) on the CPU.  Tolerances are relative to the
magnitude of the expected tensor: gradients are sums of thousands of fp32 products."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    return float((got.double().cpu() - want.double()).abs().max() / (want.double().abs().max() + 1e-30))


# ----------------------------------------------------------------------------- GEMMs
@pytest.mark.parametrize("M,N,K", [(1000, 512, 512), (3850, 2048, 512), (777 * 4, 512, 4096), (100, 148, 512), (640, 256, 512)])
def test_grad_input_gemm(dev, M, N, K):
    """dX = dY . W  (A row-major, W k-major), gradient-sized values with operand scaling"""
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, K, generator=g) * 1e-4          # K = out features (contraction)
    W = torch.randn(K, N, generator=g) / math.sqrt(K)   # [out, in]
    want = dY.double() @ W.double()
    got = T.grad_input(dY.to(dev), W.to(dev), g_scale=2.0 ** 12)
    assert rel_err(got, want) < 2e-6
    # without scaling the f16 split of 1e-4-sized values loses bits: the scaled path must be no worse
    got0 = T.grad_input(dY.to(dev), W.to(dev))
    assert rel_err(got, want) <= rel_err(got0, want) * 1.5 + 1e-9


@pytest.mark.parametrize("rows,n_out,n_in", [(3850, 512, 512), (3850, 1536, 512), (3850, 4096, 512), (3850, 512, 2048), (777, 512, 512),
                                             (154, 256, 512)])
@pytest.mark.parametrize("splits,variant", [(0, 0), (1, 3), (4, 6), (3, 2)])
def test_plane_gemm_layouts_and_fused_bias_gradient(dev, rows, n_out, n_in, splits, variant):
    """pfpp_gemm_planes in its three forms on one linear layer of the training step (include/pfpp.h): y = x.W^T, dX = dY.W (W read in
    place as the k-major operand), dW += dY^T.X (both k-major, ragged contraction = token count) with the bias gradient
    db += colsum(dY) computed by the same kernel — against fp64 of the values the planes stand for"""
    from pfpp_hip import planes as P

    g = torch.Generator().manual_seed(rows + n_out + n_in)
    G = 4096.0
    x = torch.randn(rows, n_in, generator=g)
    W = torch.randn(n_out, n_in, generator=g) / math.sqrt(n_in)
    dY = torch.randn(rows, n_out, generator=g) * 1e-4
    xp, wp, dyp = P.split(x.to(dev)), P.split(W.to(dev)), P.split(dY.to(dev), G)
    xv, wv, dyv = xp.float().double().cpu(), wp.float().double().cpu(), dyp.float().double().cpu()
    # forward and dX (non-accumulating: the split goes through the slab workspace)
    y = torch.empty(rows, n_out, device=dev)
    P.gemm(xp, wp, y, M=rows, N=n_out, K=n_in, splits=min(splits, 2), variant=variant)
    assert rel_err(y, xv @ wv.t()) < 2e-6
    dx = torch.empty(rows, n_in, device=dev)
    P.gemm(dyp, wp, dx, M=rows, N=n_in, K=n_out, w_kmajor=True, splits=splits, variant=variant)
    assert rel_err(dx, dyv @ wv) < 2e-6
    # dW and db accumulate onto what is there
    dW0 = torch.randn(n_out, n_in, generator=g) * 1e-3
    db0 = torch.randn(n_out, generator=g) * 1e-3
    dW, db = dW0.to(dev), db0.to(dev)
    P.gemm(dyp, xp, dW, M=n_out, N=n_in, K=rows, a_kmajor=True, w_kmajor=True, accumulate=True, splits=splits, variant=variant, colsum=db)
    want_dW = dW0.double() + dyv.t() @ xv
    want_db = db0.double() + dyv.sum(0)
    assert rel_err(dW, want_dW) < 2e-6
    assert rel_err(db, want_db) < 2e-6
    # the stand-alone column-sum kernel agrees; and the fused form is deterministic
    db2 = db0.to(dev)
    P.colsum(dyp, db2)
    assert rel_err(db2, want_db) < 2e-6
    dW_b, db_b = dW0.to(dev), db0.to(dev)
    P.gemm(dyp, xp, dW_b, M=n_out, N=n_in, K=rows, a_kmajor=True, w_kmajor=True, accumulate=True, splits=splits, variant=variant, colsum=db_b)
    assert torch.equal(dW_b, dW) and torch.equal(db_b, db)


def test_deferred_k_split_reductions_in_one_launch_equal_the_immediate_ones(dev):
    """the six weight gradients of a transformer block (dW += dY^T . X, db += colsum dY) with their K-split reductions handed back
    (pfpp_gemm_planes_args.defer) and run by ONE pfpp_slab_reduce_group launch == each GEMM reducing at once, bit for bit; an
    unsplit GEMM leaves an empty job"""
    from pfpp_hip import planes as P

    g = torch.Generator().manual_seed(11)
    rows, G = 3850, 4096.0
    shapes = [(512, 2048, True), (4096, 512, True), (512, 512, True), (1536, 512, False), (512, 512, True), (1536, 512, False)]
    ws = torch.empty(64 * 1024 * 1024, device=dev)
    jobs, used, outs, refs = [], 0, [], []
    for n_out, n_in, has_bias in shapes:
        x = P.split(torch.randn(rows, n_in, generator=g).to(dev))
        dy = P.split((torch.randn(rows, n_out, generator=g) * 1e-4).to(dev), G)
        dW0, db0 = torch.randn(n_out, n_in, generator=g).to(dev) * 1e-3, torch.randn(n_out, generator=g).to(dev) * 1e-3
        dW_r, db_r = dW0.clone(), db0.clone()
        P.gemm(dy, x, dW_r, M=n_out, N=n_in, K=rows, a_kmajor=True, w_kmajor=True, accumulate=True, colsum=db_r if has_bias else None)
        dW, db = dW0.clone(), db0.clone()
        job = P.SlabJob()
        P.gemm(dy, x, dW, M=n_out, N=n_in, K=rows, a_kmajor=True, w_kmajor=True, accumulate=True, colsum=db if has_bias else None,
               ws=ws[used // 4:], defer=job)
        assert job.splits >= 2 and job.M == n_out and job.N == n_in
        assert torch.equal(dW, dW0)                    # nothing has been added yet
        used += (job.splits * job.M * (job.N + 1) * 4 + 255) // 256 * 256
        jobs.append(job); outs.append((dW, db)); refs.append((dW_r, db_r))
    P.slab_reduce_group(jobs)
    for (dW, db), (dW_r, db_r) in zip(outs, refs):
        assert torch.equal(dW, dW_r) and torch.equal(db, db_r)
    # an unsplit launch writes C itself and leaves nothing behind
    x, dy = P.split(torch.randn(64, 512, generator=g).to(dev)), P.split(torch.randn(64, 512, generator=g).to(dev))
    dW = torch.zeros(512, 512, device=dev)
    job = P.SlabJob()
    P.gemm(dy, x, dW, M=512, N=512, K=64, a_kmajor=True, w_kmajor=True, accumulate=True, splits=1, ws=ws, defer=job)
    assert job.splits == 0 and float(dW.abs().max()) > 0
    P.slab_reduce_group([job])                          # a no-op
    with pytest.raises(P._lib.PfppError, match="defer"):
        P.gemm(dy, x, dW, M=512, N=512, K=64, a_kmajor=True, w_kmajor=True, bias=torch.zeros(512, device=dev), ws=ws, defer=job)


@pytest.mark.parametrize("M,N,K", [(3850, 512, 512), (16000, 1536, 512), (1234 * 4, 512, 2048), (640, 512, 148), (32, 1024, 512)])
def test_grad_weight_gemm(dev, M, N, K):
    """dW [N,K] = dY[M,N]^T . X[M,K]  (both operands k-major, split-K atomics)"""
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(M + K)
    dY = torch.randn(M, N, generator=g) * 1e-4
    X = torch.randn(M, K, generator=g)
    want = dY.double().t() @ X.double()
    dW = torch.zeros(N, K, device=dev)
    T.grad_weight(dY.to(dev), X.to(dev), dW, g_scale=2.0 ** 12)
    assert rel_err(dW, want) < 5e-6
    T.grad_weight(dY.to(dev), X.to(dev), dW, g_scale=2.0 ** 12)      # accumulates
    assert rel_err(dW, 2 * want) < 5e-6
    # the bias gradient from the same launch: db += column sums of the (unscaled) dY, taken from the staged dY tiles
    dW2 = torch.zeros(N, K, device=dev)
    db = torch.full((N,), 0.5, device=dev)
    T.grad_weight(dY.to(dev), X.to(dev), dW2, g_scale=2.0 ** 12, db=db)
    assert rel_err(dW2, want) < 5e-6
    assert rel_err(db - 0.5, dY.double().sum(0)) < 2e-5
    db2 = torch.zeros(N, device=dev)
    T.colsum(dY.to(dev), db2)
    assert rel_err(db - 0.5, db2.cpu()) < 2e-5


def test_grad_weight_group(dev):
    """the six weight gradients of a transformer layer (and ragged small ones) in one grouped launch = one launch each"""
    from pfpp_hip import _lib, train_ops as T

    g = torch.Generator().manual_seed(11)
    M = 3850
    shapes = [(512, 2048), (4096, 512), (512, 512), (1536, 512), (512, 512), (1536, 512), (256, 148), (4, 260)]
    probs, wants = [], []
    for n_out, k_in in shapes:
        dY = torch.randn(M if n_out > 4 else 154, n_out, generator=g) * 1e-4
        X = torch.randn(dY.shape[0], k_in, generator=g)
        dW = torch.randn(n_out, k_in, generator=g) * 1e-3            # accumulates into what is there
        wants.append(dW.double() + dY.double().t() @ X.double())
        probs.append((dY.to(dev), X.to(dev), dW.to(dev)))
    T.grad_weight_group(probs, g_scale=2.0 ** 12)
    for (_, _, dW), want in zip(probs, wants):
        assert rel_err(dW, want) < 5e-6
    # same through the per-problem entry point, bitwise equal when neither splits K... (different split factors: tolerance)
    single = torch.zeros(512, 512, device=dev)
    T.grad_weight(probs[2][0], probs[2][1], single, g_scale=2.0 ** 12)
    grouped = torch.zeros(512, 512, device=dev)
    T.grad_weight_group([(probs[2][0], probs[2][1], grouped)], g_scale=2.0 ** 12)
    assert rel_err(grouped, single.double().cpu()) < 2e-6
    with pytest.raises(ValueError):
        T.grad_weight_group([probs[0]] * 9)
    with pytest.raises(ValueError):
        T.grad_weight_group([])


def test_gemm_grad_batched_and_bad_args(dev):
    from pfpp_hip import _lib, train_ops as T

    g = torch.Generator().manual_seed(3)
    nb, B, C = 12, 32, 512
    dmods = torch.randn(nb, B, 2 * C, generator=g) * 1e-3
    se = torch.randn(nb, B, C, generator=g)
    want = torch.einsum("zbn,zbk->znk", dmods.double(), se.double())
    dW = torch.zeros(nb, 2 * C, C, device=dev)
    T.gemm_grad(dmods.to(dev), se.to(dev), dW, M=2 * C, N=C, K=B, lda=2 * C, ldw=C, ldc=C, a_kmajor=True, w_kmajor=True,
                accumulate=True, batch=nb, sA=B * 2 * C, sW=B * C, sC=2 * C * C, a_scale=1024.0)
    assert rel_err(dW, want) < 5e-6
    a = torch.randn(256, 64).to(dev)
    o = torch.zeros(64, 64, device=dev)
    with pytest.raises(_lib.PfppError, match="split_k"):
        T.gemm_grad(a, a, o, M=64, N=64, K=256, lda=64, ldw=64, ldc=64, a_kmajor=True, w_kmajor=True, accumulate=False,
                    split_k=2)
    with pytest.raises(_lib.PfppError, match="k-major A"):
        T.gemm_grad(dmods.to(dev), se.to(dev), dW, M=3, N=C, K=B, lda=2 * C, ldw=C, ldc=C, a_kmajor=True, w_kmajor=True,
                    accumulate=True)


@pytest.mark.parametrize("n,B,C", [(12, 32, 512), (4, 7, 512), (2, 45, 128), (3, 1, 64)])
def test_ada_linear_backward_vs_float64(dev, n, B, C):
    """pfpp_ada_linear_bwd (csrc/ada_bwd.hip): weight / bias gradients (accumulated into what is there) and d(embedded timestep) of the
    AdaLN modulation linears (attention.py:21-25) against float64; B below, at and above the 32 puzzles of one pass; twice = bit-identical"""
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(n * 1000 + B)
    N2 = 2 * C
    dmods = torch.randn(n, B, N2, generator=g) * 1e-3
    se = torch.randn(n, B, C, generator=g)
    w = torch.randn(n, N2, C, generator=g) / math.sqrt(C)
    gw0 = torch.randn(n, N2, C, generator=g) * 1e-3
    gb0 = torch.randn(n, N2, generator=g) * 1e-3
    want_gw = gw0.double() + torch.einsum("jbn,jbk->jnk", dmods.double(), se.double())
    want_gb = gb0.double() + dmods.double().sum(1)
    want_dse = torch.einsum("jbn,jnk->jbk", dmods.double(), w.double())
    outs = []
    for _ in range(2):
        gw, gb = gw0.to(dev), gb0.to(dev)
        dse = T.ada_linear_bwd(dmods.to(dev), se.to(dev), w.to(dev), gw, gb)
        outs.append((gw.clone(), gb.clone(), dse.clone()))
    gw, gb, dse = outs[0]
    assert rel_err(gw, want_gw) < 1e-6 and rel_err(gb, want_gb) < 1e-6 and rel_err(dse, want_dse) < 2e-6
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    with pytest.raises(Exception):
        T.ada_linear_bwd(dmods.to(dev), se.to(dev), w.to(dev)[:, :, : C - 1].contiguous(), gw, gb)


def test_colsum(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(4)
    x = torch.randn(3851, 1536, generator=g)
    out = torch.zeros(1536, device=dev)
    T.colsum(x.to(dev), out)
    assert rel_err(out, x.double().sum(0)) < 1e-6
    # narrow, strided (output head: 3 of 7 columns)
    y = torch.randn(154, 7, generator=g)
    o3 = torch.zeros(3, device=dev)
    T.colsum(y.to(dev), o3, rows=154, cols=3, ld=7, accumulate=False)
    assert rel_err(o3, y[:, :3].double().sum(0)) < 1e-6
    o4 = torch.zeros(4, device=dev)
    T.colsum(y.to(dev), o4, rows=154, cols=4, ld=7, x_off=3)
    assert rel_err(o4, y[:, 3:].double().sum(0)) < 1e-6


# ----------------------------------------------------------------------------- dropout / GEGLU / activations
def test_dropout_mask_statistics_and_consistency(dev):
    from pfpp_hip import train_ops as T

    n = 1 << 20
    for p in (0.1, 0.2):
        keep = T.dropout_mask(n, p, seed=1234, site=7, device=dev).cpu().float()
        assert abs(float(keep.mean()) - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n)
        # neighbouring elements / other sites are uncorrelated
        k2 = T.dropout_mask(n, p, seed=1234, site=8, device=dev).cpu().float()
        assert abs(float(((keep - keep.mean()) * (k2 - k2.mean())).mean())) < 5e-3 * p
        assert abs(float(((keep[1:] - keep.mean()) * (keep[:-1] - keep.mean())).mean())) < 5e-3 * p
    x = torch.randn(1000, 512)
    res = torch.randn(1000, 512)
    keep = T.dropout_mask(x.numel(), 0.2, 99, 3, dev).view(1000, 512).cpu().float()
    got = T.dropout(x.to(dev), 0.2, 99, 3, res=res.to(dev))
    assert torch.allclose(got.cpu(), res + x * keep / 0.8, rtol=1e-6, atol=1e-6)
    assert torch.equal(T.dropout(x.to(dev), 0.0, 1, 1).cpu(), x)


def test_geglu_fwd_bwd(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(5)
    rows, inner = 777, 2048
    z = torch.randn(rows, 2 * inner, generator=g, dtype=torch.float64, requires_grad=True)
    du = torch.randn(rows, inner, generator=g, dtype=torch.float64)
    p, seed, site = 0.1, 42, 11
    keep = T.dropout_mask(rows * inner, p, seed, site, dev).view(rows, inner).cpu().double()
    v, gate = z.chunk(2, dim=-1)
    u = v * F.gelu(gate) * keep / (1 - p)
    u.backward(du)
    zf = z.detach().float().to(dev)
    got_u = T.geglu(zf, p, seed, site)
    assert rel_err(got_u, u.detach()) < 1e-6
    got_dz = T.geglu_bwd(zf, du.float().to(dev), p, seed, site)
    assert rel_err(got_dz, z.grad) < 1e-6


def test_act_fwd_bwd(dev):
    from pfpp_hip import train_ops as T

    x = torch.randn(154, 512, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(154, 512, dtype=torch.float64)
    y = F.silu(x)
    y.backward(dy)
    xf = x.detach().float().to(dev)
    assert rel_err(T.act(xf, "silu"), y.detach()) < 1e-6
    assert rel_err(T.act_bwd(xf, dy.float().to(dev), "silu"), x.grad) < 1e-6


# ----------------------------------------------------------------------------- LayerNorm backward
def test_layernorm_bwd_adaln_grouped(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(6)
    B, L, C = 5, 25, 512
    counts = [3, 1, 7, 2, 4]
    frag_b = torch.tensor(sum(([b] * c for b, c in enumerate(counts)), []), dtype=torch.int32)
    n = frag_b.numel()
    x = torch.randn(n * L, C, generator=g, dtype=torch.float64, requires_grad=True)
    mod = (torch.randn(B, 2 * C, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    dy = torch.randn(n * L, C, generator=g, dtype=torch.float64) * 1e-3
    rb = frag_b.long().repeat_interleave(L)
    y = F.layer_norm(x, (C,)) * (1 + mod[rb, :C]) + mod[rb, C:]
    y.backward(dy)
    dx0 = torch.randn(n * L, C, generator=g) * 1e-3          # the running residual gradient the kernel adds to
    dx = dx0.clone().to(dev)
    dmod = torch.zeros(B, 2 * C, device=dev)
    T.layernorm_bwd(x.detach().float().to(dev), dy.float().to(dev), dx, mod=mod.detach().float().to(dev),
                    group_batch=frag_b.to(dev), group_rows=L, dmult=dmod, dadd=dmod[:, C:], ld_d=2 * C)
    assert rel_err(dx.cpu() - dx0, x.grad) < 2e-5
    assert rel_err(dmod, mod.grad) < 2e-5


def test_layernorm_bwd_affine_and_uniform_batches(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(7)
    rows, C = 1003, 512
    x = torch.randn(rows, C, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)
    beta = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    F.layer_norm(x, (C,), gamma, beta).backward(dy)
    dx = torch.zeros(rows, C, device=dev)
    dg = torch.zeros(2, C, device=dev)
    T.layernorm_bwd(x.detach().float().to(dev), dy.float().to(dev), dx, gamma=gamma.detach().float().to(dev),
                    group_rows=32, dmult=dg[0], dadd=dg[1], ld_d=0)
    assert rel_err(dx, x.grad) < 2e-5
    assert rel_err(dg[0], gamma.grad) < 2e-5 and rel_err(dg[1], beta.grad) < 2e-5
    # AdaLN with uniform batches (padded layout): 2 puzzles x 100 rows, groups of 25 rows
    x2 = torch.randn(200, 256, generator=g, dtype=torch.float64, requires_grad=True)
    mod = torch.randn(2, 512, generator=g, dtype=torch.float64, requires_grad=True)
    dy2 = torch.randn(200, 256, generator=g, dtype=torch.float64)
    rb = torch.arange(200) // 100
    (F.layer_norm(x2, (256,)) * (1 + mod[rb, :256]) + mod[rb, 256:]).backward(dy2)
    dx2 = torch.zeros(200, 256, device=dev)
    dm = torch.zeros(2, 512, device=dev)
    T.layernorm_bwd(x2.detach().float().to(dev), dy2.float().to(dev), dx2, mod=mod.detach().float().to(dev), group_rows=25,
                    rows_per_batch=100, dmult=dm, dadd=dm[:, 256:], ld_d=512)
    assert rel_err(dx2, x2.grad) < 2e-5 and rel_err(dm, mod.grad) < 2e-5


def test_dropout_fused_into_layernorm_forward_and_backward(dev):
    """pfpp_dropout_layernorm == pfpp_dropout then pfpp_layernorm, pfpp_layernorm_bwd_dropout == pfpp_layernorm_bwd then
    pfpp_dropout: same masks, same values (the fused kernels repeat the arithmetic of the two they replace), all LayerNorm forms"""
    from pfpp_hip import ops
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(16)
    L, C = 25, 512
    counts = [3, 1, 7, 2, 4]
    frag_b = torch.tensor(sum(([b] * c for b, c in enumerate(counts)), []), dtype=torch.int32).to(dev)
    rows = frag_b.numel() * L
    y = torch.randn(rows, C, generator=g).to(dev)
    res = torch.randn(rows, C, generator=g).to(dev)
    mod = (torch.randn(len(counts), 2 * C, generator=g) * 0.3).to(dev)
    gamma, beta = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    p, seed = 0.2, 77
    # forward: AdaLN over the grouped token list with a residual, affine form with a residual, no residual (token dropout)
    for site, r, kw, ln in ((4, res, dict(mod=mod, group_batch=frag_b, group_rows=L), lambda h: ops.layernorm_grouped(h, mod, frag_b, L)),
                            (5, res, dict(gamma=gamma, beta=beta), lambda h: ops.layernorm(h, gamma=gamma, beta=beta)),
                            (0, None, dict(mod=mod, group_batch=frag_b, group_rows=L), lambda h: ops.layernorm_grouped(h, mod, frag_b, L))):
        h_ref = T.dropout(y, p, seed, site, res=r)
        n_ref = ln(h_ref)
        h, n = T.dropout_layernorm(y.clone(), r, p, seed, site, **kw)
        assert torch.equal(h, h_ref) and rel_err(n, n_ref.cpu()) < 1e-6      # the affine step may contract differently: an ulp
    kept = (T.dropout(torch.ones_like(y), p, seed, 4) != 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01
    # backward
    x = torch.randn(rows, C, generator=g).to(dev)
    dy = (torch.randn(rows, C, generator=g) * 1e-3).to(dev)
    dx0 = (torch.randn(rows, C, generator=g) * 1e-3).to(dev)
    for kw in (dict(mod=mod, group_batch=frag_b, group_rows=L, ld_d=2 * C), dict(gamma=gamma, group_rows=32, ld_d=0)):
        outs = []
        for fused in (False, True):
            dx = dx0.clone()
            dm = torch.zeros(len(counts), 2 * C, device=dev)
            d = T.layernorm_bwd(x, dy, dx, dmult=dm if "mod" in kw else dm[0, :C], dadd=dm[:, C:] if "mod" in kw else dm[0, C:],
                                drop=(p, seed, 9) if fused else None, **kw)
            if not fused:
                assert d is dx
                d = T.dropout(dx, p, seed, 9)
            outs.append((dx, d, dm))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert rel_err(outs[1][2], outs[0][2].cpu()) < 1e-6          # atomics: order of the column sums differs between runs


# ----------------------------------------------------------------------------- attention backward
def _attn_ref(qkv, H, dh, scale, groups, key_valid=None):
    """per-sequence softmax attention on a packed [rows, 3*H*dh] projection; groups = list of (start, len)"""
    C = H * dh
    outs = []
    for gi, (s, n) in enumerate(groups):
        q, k, v = (qkv[s:s + n, i * C:(i + 1) * C].view(n, H, dh).transpose(0, 1) for i in range(3))
        sc = q @ k.transpose(1, 2) * scale
        if key_valid is not None:
            sc = sc.masked_fill(~key_valid[gi, :n].bool()[None, None, :], float("-inf"))
        outs.append((sc.softmax(-1) @ v).transpose(0, 1).reshape(n, C))
    return torch.cat(outs, 0)


def test_attn_blockdiag_bwd(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(8)
    n_frag, L, H, dh = 37, 25, 8, 64
    qkv = torch.randn(n_frag * L, 3 * H * dh, generator=g, dtype=torch.float64, requires_grad=True)
    dO = torch.randn(n_frag * L, H * dh, generator=g, dtype=torch.float64) * 1e-3
    scale = 1 / math.sqrt(dh)
    _attn_ref(qkv, H, dh, scale, [(f * L, L) for f in range(n_frag)]).backward(dO)
    got = T.attn_blockdiag_bwd(qkv.detach().float().to(dev), dO.float().to(dev), n_frag, L, H, dh, scale)
    assert rel_err(got, qkv.grad) < 2e-5


@pytest.mark.parametrize("lens,dh,H,masked", [([50, 500, 125, 25], 64, 8, False), ([190, 190], 32, 8, True),
                                              ([500, 500, 500], 64, 8, True)])
def test_attn_dense_train_and_bwd(dev, lens, dh, H, masked):
    from pfpp_hip import ops, train_ops as T

    g = torch.Generator().manual_seed(sum(lens))
    rows = sum(lens)
    C = H * dh
    qkv = torch.randn(rows, 3 * C, generator=g, dtype=torch.float64, requires_grad=True)
    dO = torch.randn(rows, C, generator=g, dtype=torch.float64) * 1e-3
    offs = [sum(lens[:i]) for i in range(len(lens))]
    kvm = None
    if masked:
        kvm = (torch.rand(len(lens), max(lens), generator=g) < 0.6)
        kvm[:, 0] = True
    scale = 1 / math.sqrt(dh)
    want = _attn_ref(qkv, H, dh, scale, list(zip(offs, lens)), kvm)
    want.backward(dO)
    so = torch.tensor(offs, dtype=torch.int32, device=dev)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    kv8 = kvm.to(torch.uint8).contiguous().to(dev) if masked else None
    qf = qkv.detach().float().to(dev)
    out, lse = T.attn_dense_train(qf, so, sl, max(lens), H, dh, scale, key_valid=kv8)
    assert rel_err(out, want.detach()) < 1e-5
    assert torch.equal(out, ops.attn_dense(qf, so, sl, max(lens), H, dh, scale, kv8))
    got = T.attn_dense_bwd(qf, out, dO.float().to(dev), lse, so, sl, max(lens), H, dh, scale, key_valid=kv8)
    assert rel_err(got, qkv.grad) < 2e-5
    # the same in parts with dk/dv on a second stream (pfpp_attn_dense_bwd_parts): identical bits
    aux = torch.cuda.Stream()
    got2 = T.attn_dense_bwd(qf, out, dO.float().to(dev), lse, so, sl, max(lens), H, dh, scale, key_valid=kv8, aux_stream=aux)
    torch.cuda.synchronize()
    assert torch.equal(got2, got)


# ----------------------------------------------------------------------------- small pieces
def test_pool_token_embed_backward(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(9)
    n, L, C = 154, 25, 512
    dp = torch.randn(n, C, generator=g)
    got = T.mean_pool_bwd(dp.to(dev), L)
    assert torch.allclose(got.cpu(), (dp / L).repeat_interleave(L, 0), rtol=1e-6, atol=1e-9)
    dtok = torch.randn(n * L, C, generator=g)
    ref = (torch.rand(n, generator=g) < 0.3).to(torch.uint8)
    dref = torch.zeros(2, C, device=dev)
    dx = T.token_combine_bwd(dtok.to(dev), ref.to(dev), dref, L)
    want_dx = dtok.double().view(n, L, C).sum(1)
    assert rel_err(dx, want_dx) < 1e-6
    assert rel_err(dref[1], want_dx[ref.bool()].sum(0)) < 1e-5 and rel_err(dref[0], want_dx[~ref.bool()].sum(0)) < 1e-5
    n_tab, n_emb, B = 12, 1000, 32
    tables = torch.randn(n_tab, n_emb, C, generator=g, dtype=torch.float64, requires_grad=True)
    t = torch.randint(0, n_emb, (B,), generator=g)
    t[1] = t[0]                                                  # duplicate timestep: gradients add
    dse = torch.randn(n_tab, B, C, generator=g, dtype=torch.float64)
    F.silu(tables[:, t]).backward(dse)
    dt = torch.zeros(n_tab, n_emb, C, device=dev)
    T.silu_embed_bwd(tables.detach().float().to(dev), t.to(dev), dse.float().to(dev), dt)
    assert rel_err(dt, tables.grad) < 1e-6


@pytest.mark.parametrize("n,L,slots", [(154, 25, 640), (3, 25, 3), (37, 11, 60)])
def test_token_embedding_backward_in_one_launch_vs_float64(dev, n, L, slots):
    """pfpp_token_features_t + pfpp_token_embed_bwd (csrc/embed_train.hip): the transposed feature planes hold exactly the split of
    pfpp_token_features' values (+ the ref_part indicators and the ones column); the five gradients of shape_embedding / param_fc /
    ref_part_emb (denoiser_transformer.py:117-135, 150-156, 173-185) accumulate into what is there and equal float64 of
    dtok^T . features; run twice = bit-identical (no atomics)"""
    from pfpp_hip import ops, train_ops as T

    g = torch.Generator().manual_seed(n * 7 + L)
    C = 512
    latent = torch.randn(slots, L, 64, generator=g)
    xyz = torch.rand(slots, L, 3, generator=g) * 2 - 1
    scale = torch.rand(slots, generator=g) + 0.5
    x = torch.randn(slots, 7, generator=g)
    ref = (torch.rand(slots, generator=g) < 0.3).to(torch.uint8)
    slot = torch.randperm(slots, generator=g)[:n].sort().values.to(torch.int32)
    dtok = torch.randn(n * L, C, generator=g) * 1e-4
    d = lambda t_: t_.to(dev)
    sf, pf = ops.token_features(d(latent), d(xyz), d(scale), d(x), slot=d(slot))
    fh, fl = T.token_features_t(d(latent), d(xyz), d(scale), d(x), d(slot), d(ref), n, L)
    M = n * L
    assert fh.shape[1] % 16 == 0 and fh.shape[1] >= M
    feat = (fh.float() + fl.float()).cpu()                                   # [320, Mp]
    sfc, pfc = sf.cpu(), pf.cpu()
    assert (feat[:148, :M].t() - sfc[:, :148]).abs().max() <= 2.0 ** -22 * sfc.abs().max() + 3e-8
    assert (feat[148:295, :M].t() - pfc[:, :147].repeat_interleave(L, 0)).abs().max() <= 2.0 ** -22 * pfc.abs().max() + 3e-8
    rl = ref[slot.long()].repeat_interleave(L)
    assert torch.equal(feat[295, :M], (rl == 0).float()) and torch.equal(feat[296, :M], (rl == 1).float())
    assert torch.equal(feat[297, :M], torch.ones(M)) and float(feat[298:].abs().max()) == 0.0 and float(feat[:, M:].abs().sum()) == 0.0
    g0 = [torch.randn(*shape, generator=g) * 1e-3 for shape in ((C, 148), (C,), (C, 147), (C,), (2, C))]
    D = dtok.double()
    dx = D.view(n, L, C).sum(1)
    rs = ref[slot.long()].bool()
    want = [g0[0].double() + D.t() @ sfc[:, :148].double(), g0[1].double() + D.sum(0), g0[2].double() + dx.t() @ pfc[:, :147].double(),
            g0[3].double() + D.sum(0), g0[4].double() + torch.stack([dx[~rs].sum(0), dx[rs].sum(0)])]
    runs = []
    for _ in range(2):
        gs = [d(t_.clone()) for t_ in g0]
        T.token_embed_bwd(d(dtok), fh, fl, *gs, n, L, g_scale=4096.0)
        runs.append(gs)
    for got, w_, base in zip(runs[0], want, g0):
        assert float((got.double().cpu() - w_).abs().max() / ((w_ - base.double()).abs().max() + 1e-30)) < 2e-6
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    with pytest.raises(Exception):
        T.token_embed_bwd(d(dtok), fh, fl, *runs[0], n + 1, L)


def test_mse_loss(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(10)
    n = 640
    pred = torch.randn(n, 7, generator=g, dtype=torch.float64, requires_grad=True)
    tgt = torch.randn(n, 7, generator=g, dtype=torch.float64)
    sel = torch.rand(n, generator=g) < 0.2
    loss = F.mse_loss(pred[sel], tgt[sel])
    loss.backward()
    got_l, got_d = T.mse_loss(pred.detach().float().to(dev), tgt.float().to(dev), sel.to(torch.uint8).to(dev))
    assert abs(float(got_l) - float(loss)) < 1e-6 * float(loss)
    assert rel_err(got_d, pred.grad) < 1e-6


def test_adamw_matches_torch(dev):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(11)
    n = 100003
    p0 = torch.randn(n, generator=g) * 0.05
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    hi = torch.empty(n, dtype=torch.float16, device=dev)
    lo = torch.empty(n, dtype=torch.float16, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 1e-3
        p_ref.grad = grad.clone()
        opt.step()
        T.adamw(p, grad.to(dev), m, v, lr=2e-4, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=1e-6, step=step, hi=hi, lo=lo)
    assert (p.cpu() - p_ref.detach()).abs().max() < 2e-7
    assert (hi.float() + lo.float() - p).abs().max() < 1e-7


def test_adamw_by_rows_equals_one_pass_over_the_tables(dev):
    """round 5: the AdaLN timestep tables (MyAdaLayerNorm.emb, attention.py:18-25; a third of all parameters) get gradients in the
    batch's rows only, so the armed single-rank step updates every OTHER row at the start of the backward (pfpp_adamw_rows mode 0, off
    the iteration's exposed tail) and the batch's rows at the end (mode 1, each once whatever the duplicates among the timesteps).  The
    two passes together are bit-identical to one guarded AdamW over the stack — parameters, both moments, planes, cleared gradients —
    including a non-finite gradient in a listed row (skipped and flagged)."""
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(5)
    n_tab, rows, C = 12, 3072, 64
    t = torch.tensor([7, 3071, 0, 7, 1500, 3, 1500, 2999], dtype=torch.int64)          # duplicates on purpose
    p0 = torch.randn(n_tab, rows, C, generator=g) * 0.05
    m0, v0 = torch.randn(n_tab, rows, C, generator=g) * 1e-3, torch.rand(n_tab, rows, C, generator=g) * 1e-6
    grad = torch.zeros(n_tab, rows, C)
    grad[:, t.unique()] = torch.randn(n_tab, t.unique().numel(), C, generator=g) * 1e-3
    grad[3, 1500, 5] = float("inf")
    hp = dict(lr=2e-4, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=1e-6, step=3)
    out = []
    for by_rows in (False, True):
        p, m, v, gr = (x.clone().to(dev) for x in (p0, m0, v0, grad))
        hi = torch.zeros(p.shape, dtype=torch.float16, device=dev)
        lo = torch.zeros(p.shape, dtype=torch.float16, device=dev)
        ovf = torch.zeros(2, dtype=torch.int32, device=dev)
        if by_rows:
            T.adamw_rows(p, gr, m, v, t.to(dev), mode=0, hi=hi, lo=lo, zero_grad=True, overflow=ovf, **hp)
            T.adamw_rows(p, gr, m, v, t.to(dev), mode=1, hi=hi, lo=lo, zero_grad=True, overflow=ovf, **hp)
        else:
            T.adamw(p.view(-1), gr.view(-1), m.view(-1), v.view(-1), hi=hi.view(-1), lo=lo.view(-1), zero_grad=True, overflow=ovf, **hp)
        torch.cuda.synchronize()
        out.append((p.cpu(), m.cpu(), v.cpu(), hi.cpu(), lo.cpu(), gr.cpu(), ovf.cpu()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert out[1][6].tolist() == [1, 1] and float(out[1][5].abs().max()) == 0.0
    assert torch.equal(out[1][0][3, 1500, 5], p0[3, 1500, 5])                        # the overflowed element was left alone
    assert not torch.equal(out[1][0][0, 10], p0[0, 10])                              # an unlisted row did move (decay, first-moment step)


@pytest.mark.parametrize("weight_decay", [1e-6, 1e-2])
def test_adamw_over_the_active_rows_equals_one_pass_over_the_tables(dev, weight_decay):
    """round 6: a row of the AdaLN timestep tables that has never received a gradient has zero moments, and with the reference's
    hyper-parameters (lr 2e-4, weight_decay 1e-6: fl32(1 - lr wd) = 1) its AdamW update is the identity — pfpp_adamw_rows_active leaves
    such rows alone (two thirds of the tables' parameters are never read or written), the rows silu_embed_bwd marked are updated.  Four
    steps with fresh timestep draws (rows beyond the 1,000 training timesteps included, duplicates included) against the guarded
    one-pass AdamW, bit for bit: parameters, both moments, planes, gradients; with a weight decay that does not round away every row
    is taken and the two still agree."""
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(11)
    n_tab, rows, C, B = 12, 3072, 64, 8
    p0 = torch.randn(n_tab, rows, C, generator=g) * 0.05
    out = []
    for active_path in (False, True):
        gen = torch.Generator().manual_seed(12)
        p = p0.clone().to(dev)
        m, v, gr = (torch.zeros_like(p) for _ in range(3))
        hi = torch.zeros(p.shape, dtype=torch.float16, device=dev)
        lo = torch.zeros(p.shape, dtype=torch.float16, device=dev)
        act = torch.zeros(128, dtype=torch.int32, device=dev)
        for step in range(1, 5):
            t = torch.randint(0, 1000, (B,), generator=gen)
            t[1] = t[0]
            if step == 3:
                t[5] = 3071                      # a row outside the training range still counts once it is indexed
            dse = (torch.randn(n_tab, B, C, generator=gen) * 1e-2).to(dev)
            T.silu_embed_bwd(p, t.to(dev), dse, gr, active=act if active_path else None)
            hp = dict(lr=2e-4, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=weight_decay, step=step)
            if active_path:
                T.adamw_rows_active(p, gr, m, v, act, hi=hi, lo=lo, zero_grad=True, **hp)
            else:
                T.adamw(p.view(-1), gr.view(-1), m.view(-1), v.view(-1), hi=hi.view(-1), lo=lo.view(-1), zero_grad=True, **hp)
        torch.cuda.synchronize()
        out.append((p.cpu(), m.cpu(), v.cpu(), gr.cpu(), hi.cpu(), lo.cpu(), act.cpu()))
    for a, b in zip(out[0][:4], out[1][:4]):
        assert torch.equal(a, b)
    touched = out[1][2].abs().sum(dim=(0, 2)) > 0                                   # rows with a second moment
    assert 20 <= int(touched.sum()) <= 4 * B and bool(touched[3071])
    bits = out[1][6].numpy().view("uint32")
    marked = torch.tensor([(int(bits[r >> 5]) >> (r & 31)) & 1 for r in range(rows)], dtype=torch.bool)
    assert torch.equal(marked, touched)
    # planes: identical where a row was ever written; an untouched row keeps whatever the caller put there (zeros here), the dense pass
    # wrote split(p) — the engine's planes are initialised from the parameters, so there the two agree everywhere
    assert torch.equal(out[0][4][:, touched], out[1][4][:, touched]) and torch.equal(out[0][5][:, touched], out[1][5][:, touched])
    if weight_decay > 1e-4:
        assert torch.equal(out[0][4], out[1][4]) and not torch.equal(out[1][0][0, 2000], p0[0, 2000])      # every row decayed


# ----------------------------------------------------------------------------- train-mode BatchNorm
@pytest.mark.parametrize("rows,C,pool", [(154 * 256 * 32 // 16, 64, 0), (8192 * 3 + 64, 128, 64), (1600 * 4, 512, 64)])
def test_bn_stats_and_apply(dev, rows, C, pool):
    from pfpp_hip import train_ops as T

    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2.0 + torch.randn(C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    want = F.relu(F.batch_norm(x.t().reshape(1, C, rows, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)).reshape(C, rows).t()
    if pool:
        want = want.reshape(rows // pool, pool, C).amax(1)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    mean, var = T.bn_stats(x.to(dev), rm_d, rv_d, momentum=0.1)
    assert rel_err(mean, x.double().mean(0)) < 1e-6 and rel_err(var, x.double().var(0, unbiased=False)) < 1e-6
    assert (rm_d.cpu() - rm_ref).abs().max() < 1e-6 and (rv_d.cpu() - rv_ref).abs().max() < 1e-6
    got = T.bn_apply(x.to(dev), mean, var, gamma.to(dev), beta.to(dev), pool=pool)
    assert (got.cpu() - want).abs().max() < 1e-5


# ----------------------------------------------------------------------------- hi + lo == x for every plane producer
def _assert_planes_carry(x32: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, scale: float, what: str):
    """the split's contract (csrc/pfpp_common.h): hi + lo == scale * x to 22 bits (absolute floor: the fp16 subnormal spacing).
    The failure this guards against is a stored hi that is NOT the hi the low half was computed against (two different roundings of
    one value): that leaves hi + lo off by a whole fp16 ulp of hi — 2^-11 relative, five hundred times the bound."""
    x = x32.double().cpu().reshape(-1) * scale
    got = hi.double().cpu().reshape(-1) + lo.double().cpu().reshape(-1)
    ok = torch.isfinite(x)
    err = (got - x).abs()[ok]
    bound = x.abs()[ok] * 2.0 ** -20 + 1.3e-7
    bad = err > bound
    assert int(ok.sum()) > 0.99 * x.numel(), what
    assert not bool(bad.any()), (what, int(bad.sum()), float((err / bound).max()),
                                 float(x[ok][bad][0]), float(got[ok][bad][0]))
    # and hi is a nearest-or-tie neighbour of the value: |x - hi| <= half an fp16 ulp of hi (what keeps lo in its 11 bits)
    h = hi.double().cpu().reshape(-1)[ok]
    ulp = torch.maximum(2.0 ** (torch.floor(torch.log2(h.abs().clamp_min(2.0 ** -14))) - 10), torch.tensor(2.0 ** -24, dtype=torch.float64))
    assert bool(((x[ok] - h).abs() <= 0.5 * ulp * (1 + 2.0 ** -10) + 1e-12).all()), what


def test_every_plane_producer_writes_hi_plus_lo_equal_to_its_fp32_value(dev):
    """VERDICT r3 'do this' 1b.  Every kernel that hands a GEMM operand over as split-f16 planes is run so that the SAME launch (or, for
    the kernels with one output form per call, a second deterministic launch on the same inputs) also yields the fp32 value, on inputs
    large enough that thousands of elements sit at double-rounding positions of their products (random mantissas: ~1e-4 of all
    elements), with gradient-sized values lifted by the power-of-two scale the backward uses.  Build-side half of the guarantee: the
    library is compiled with the mixed-precision fused conversions off (pfpp_hip.build.NO_MIX; test_abi_and_host checks the
    disassembly), so a split can only ever see one fp16 rounding of its argument."""
    import ctypes as C

    from pfpp_hip import _lib, ops, planes as P, train_ops as T
    from pfpp_hip.packing import PW

    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    M, Cc, L, H, dh = 3850, 512, 25, 8, 64
    Fv = M // L
    st = ops._stream
    ptr = ops._ptr
    G = 4096.0

    def planes(rows, cols, scale=1.0):
        return P.Planes.empty(rows, cols, dev, scale)

    # 1. stand-alone split kernel (pfpp_split_planes), unscaled and scaled
    x = (torch.randn(M, Cc, generator=g) * torch.logspace(-3, 3, Cc)).to(dev)
    for s in (1.0, G):
        p_ = P.split(x, s)
        _assert_planes_carry(x, p_.hi, p_.lo, s, f"split_planes x{s}")
    # 2. LayerNorm forward: AdaLN (grouped) and affine, plane form against the fp32 form of the same kernel family
    h = torch.randn(M, Cc, generator=g).to(dev) * 3.0
    mods = (torch.randn(32, 2 * Cc, generator=g) * 0.5).to(dev)
    frag_b = torch.randint(0, 32, (Fv,), generator=g).to(torch.int32).to(dev)
    n32 = ops.layernorm_grouped(h, mods, frag_b, L)
    nsp = ops.SplitAct.empty(M, Cc, dev)
    ops.layernorm_grouped(h, mods, frag_b, L, out=nsp)
    _assert_planes_carry(n32, nsp.hi, nsp.lo, 1.0, "layernorm_grouped")
    gamma, beta = torch.randn(Cc, generator=g).to(dev), torch.randn(Cc, generator=g).to(dev)
    n32 = ops.layernorm(h, gamma=gamma, beta=beta)
    ops.layernorm(h, gamma=gamma, beta=beta, out=nsp)
    _assert_planes_carry(n32, nsp.hi, nsp.lo, 1.0, "layernorm affine")
    # 3. dropout + residual + LayerNorm forward: fp32 n and planes from ONE launch
    y = torch.randn(M, Cc, generator=g).to(dev)
    n32 = torch.empty(M, Cc, device=dev)
    np_ = planes(M, Cc)
    y2 = y.clone()
    _lib.check(lib.pfpp_dropout_layernorm_p(ptr(y2), ptr(h), ptr(y2), ptr(n32), ptr(mods), mods.stride(0), None, None, ptr(frag_b), L, 1,
                                            M, Cc, 1e-5, 0.2, 77, 3, P._pl(np_), st()), "pfpp_dropout_layernorm_p")
    _assert_planes_carry(n32, np_.hi, np_.lo, 1.0, "dropout_layernorm")
    # 4. GEGLU forward / backward: fp32 and planes from one launch
    inner = 2048
    z = torch.randn(M, 2 * inner, generator=g).to(dev) * 2.0
    u32 = torch.empty(M, inner, device=dev)
    up = planes(M, inner)
    _lib.check(lib.pfpp_geglu_p(ptr(z), ptr(u32), M, inner, 0.2, 5, 9, P._pl(up), st()), "pfpp_geglu_p")
    _assert_planes_carry(u32, up.hi, up.lo, 1.0, "geglu")
    du = (torch.randn(M, inner, generator=g) * 1e-4).to(dev)
    dz32 = torch.empty(M, 2 * inner, device=dev)
    dzp = planes(M, 2 * inner, G)
    _lib.check(lib.pfpp_geglu_bwd_p(ptr(z), ptr(du), ptr(dz32), M, inner, 0.2, 5, 9, P._pl(dzp), st()), "pfpp_geglu_bwd_p")
    _assert_planes_carry(dz32, dzp.hi, dzp.lo, G, "geglu_bwd")
    # 5. LayerNorm backward (+ the dropout that follows it in the chain): dx planes and the continued-chain planes from one launch
    dy = (torch.randn(M, Cc, generator=g) * 1e-4).to(dev)
    dx = (torch.randn(M, Cc, generator=g) * 1e-4).to(dev)
    dmods = torch.zeros(32, 2 * Cc, device=dev)
    ret, dxp, ret32 = T.layernorm_bwd_planes(h, dy, dx, G, mod=mods, group_batch=frag_b, group_rows=L, dmult=dmods, dadd=dmods[:, Cc:],
                                             ld_d=2 * Cc, drop=(0.2, 5, 4), want_ret=True, want_dx=True, ret_fp32=True)
    _assert_planes_carry(dx, dxp.hi, dxp.lo, G, "layernorm_bwd dx")
    _assert_planes_carry(ret32, ret.hi, ret.lo, G, "layernorm_bwd dropout(dx)")
    # 6. attention forward: per-fragment (split-f16 kernel, two output forms) and dense (both forms from one launch)
    qkv = torch.randn(M, 3 * Cc, generator=g).to(dev)
    att_scale = 1.0 / math.sqrt(dh)
    a32 = ops.attn_blockdiag(qkv, Fv, L, H, dh, att_scale)
    asp = ops.SplitAct.empty(M, Cc, dev)
    ops.attn_blockdiag(qkv, Fv, L, H, dh, att_scale, out=asp)
    _assert_planes_carry(a32, asp.hi, asp.lo, 1.0, "attn_blockdiag")
    lens = [125, 500, 25, 350, 200, 75, 300, 475, 150, 400, 50, 450, 250, 500]
    rows = sum(lens)
    seq_len = torch.tensor(lens, dtype=torch.int32)
    seq_off = (torch.cumsum(seq_len, 0) - seq_len).to(torch.int32).to(dev)
    seq_len = seq_len.to(dev)
    qkv2 = qkv[:rows].contiguous()
    o32, op_, lse = T.attn_dense_train_planes(qkv2, seq_off, seq_len, 500, H, dh, att_scale)
    _assert_planes_carry(o32, op_.hi, op_.lo, 1.0, "attn_dense")
    # 7. attention backward: dqkv as fp32 and as planes of G * dqkv from one launch
    do = (torch.randn(rows, Cc, generator=g) * 1e-4).to(dev)
    dq32 = torch.empty(rows, 3 * Cc, device=dev)
    dqp = planes(rows, 3 * Cc, G)
    dvec = torch.empty_like(lse)
    _lib.check(lib.pfpp_attn_dense_bwd_p(ptr(qkv2), ptr(o32), ptr(do), ptr(lse), ptr(dvec), ptr(dq32), ptr(seq_off), ptr(seq_len), None, 0,
                                         len(lens), 500, H, dh, att_scale, P._pl(dqp), st()), "pfpp_attn_dense_bwd_p")
    _assert_planes_carry(dq32, dqp.hi, dqp.lo, G, "attn_dense_bwd")
    do = (torch.randn(M, Cc, generator=g) * 1e-4).to(dev)
    dq32 = torch.empty(M, 3 * Cc, device=dev)
    dqp = planes(M, 3 * Cc, G)
    _lib.check(lib.pfpp_attn_blockdiag_bwd_p(ptr(qkv), ptr(do), ptr(dq32), Fv, L, H, dh, att_scale, P._pl(dqp), st()),
               "pfpp_attn_blockdiag_bwd_p")
    _assert_planes_carry(dq32, dqp.hi, dqp.lo, G, "attn_blockdiag_bwd")
    # 8. GEMM epilogue with plane output (bias + activation applied to the accumulator, then split): against its fp32 form
    W = PW(torch.randn(Cc, Cc, generator=g).to(dev) / math.sqrt(Cc))
    bias = torch.randn(Cc, generator=g).to(dev)
    for act in ("none", "relu", "silu"):
        c32 = ops.gemm(nsp, W, M=M, N=Cc, K=Cc, lda=Cc, bias=bias, act=act)
        csp = ops.SplitAct.empty(M, Cc, dev)
        ops.gemm(nsp, W, M=M, N=Cc, K=Cc, lda=Cc, bias=bias, act=act, out=csp)
        _assert_planes_carry(c32, csp.hi, csp.lo, 1.0, f"gemm epilogue {act}")
    # 9. AdamW: the refreshed weight planes against the parameters it just wrote
    n = 1 << 20
    p = torch.randn(n, generator=g).to(dev) * 0.05
    gr = (torch.randn(n, generator=g) * 1e-3).to(dev)
    m_, v_ = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    hi, lo = torch.empty(n, dtype=torch.float16, device=dev), torch.empty(n, dtype=torch.float16, device=dev)
    for step in (1, 2, 3):
        T.adamw(p, gr, m_, v_, lr=2e-4, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=1e-6, step=step, hi=hi, lo=lo)
        _assert_planes_carry(p, hi, lo, 1.0, "adamw planes")


@pytest.mark.parametrize("R", [154, 33, 640])
def test_fused_output_heads_forward_and_backward_vs_float64(dev, R):
    """csrc/heads.hip (pool -> both 3-layer heads as one kernel): mlp_out_trans / mlp_out_rot of
    DenoiserTransformer._out (denoiser_transformer.py:138-147, :58-61) and their autograd, against float64 torch on the same
    weights — prediction rows scattered to their slots, saved activations, the chain's gradients (da0, da1, d pooled broadcast
    over the L latent points), the last layer's weight gradient and all three bias gradients of both heads."""
    from pfpp_hip import train_ops as T
    from pfpp_hip._lib import HeadGrads
    from pfpp_hip.packing import PW

    g = torch.Generator().manual_seed(R)
    C, C2, L, G = 512, 256, 25, 4096.0
    pooled = torch.randn(R, C, generator=g)
    heads, structs, grads, gstructs = [], [], [], []
    for n_out in (3, 4):
        W0 = torch.randn(C, C, generator=g) / math.sqrt(C)
        W2 = torch.randn(C2, C, generator=g) / math.sqrt(C)
        W4 = torch.randn(n_out, C2, generator=g) / math.sqrt(C2)
        b0, b2, b4 = (torch.randn(n, generator=g) * 0.1 for n in (C, C2, n_out))
        heads.append((W0, b0, W2, b2, W4, b4))
        dev_t = [t.to(dev).contiguous() for t in (W0, W2, W4, b0, b2, b4)]
        pw0, pw2 = PW(dev_t[0]), PW(dev_t[1])
        structs.append((T.head_params(pw0, pw2, *dev_t[2:]), pw0, pw2, dev_t))
        gb = [torch.zeros(n_out, C2, device=dev), torch.zeros(n_out, device=dev), torch.zeros(C2, device=dev), torch.zeros(C, device=dev)]
        grads.append(gb)
        gstructs.append(HeadGrads(*(t.data_ptr() for t in gb)))
    perm = torch.randperm(R + 7, generator=g)[:R].to(torch.int32)
    out = torch.zeros(R + 7, 7, device=dev)
    saved = T.heads_fwd(pooled.to(dev), structs[0][0], structs[1][0], out, slot=perm.to(dev), save=True)
    # float64 reference with autograd (weights as the planes carry them: hi + lo of scale * W)
    x = pooled.double().requires_grad_(True)
    ref_out, inter, leaves = [], [], []
    for (W0, b0, W2, b2, W4, b4), (_, pw0, pw2, _) in zip(heads, structs):
        W0e = ((pw0.hi.double() + pw0.lo.double()) / pw0.scale).cpu()
        W2e = ((pw2.hi.double() + pw2.lo.double()) / pw2.scale).cpu()
        W4d, b0d, b2d, b4d = (t.double().requires_grad_(True) for t in (W4, b0, b2, b4))
        a0 = x @ W0e.t() + b0d
        a0.retain_grad()
        v0 = torch.nn.functional.silu(a0)
        a1 = v0 @ W2e.t() + b2d
        a1.retain_grad()
        v1 = torch.nn.functional.silu(a1)
        ref_out.append(v1 @ W4d.t() + b4d)
        inter.append((a0, v0, a1, v1))
        leaves.append((W4d, b4d, b2d, b0d))
    want = torch.cat(ref_out, 1)
    got = out.cpu()[perm.long()]
    assert rel_err(got, want.detach()) < 2e-6
    untouched = torch.ones(R + 7, dtype=torch.bool)
    untouched[perm.long()] = False
    assert float(out.cpu()[untouched].abs().max()) == 0.0
    for k in range(4):
        for hd in range(2):
            assert rel_err(saved[k][hd], inter[hd][k].detach()) < 2e-6, (k, hd)
    # eval form: weights static -> read from their fragment-blocked planes (same products, another order of the k-steps)
    st_structs = [T.head_params(pw0, pw2, *dev_t[2:], static=True) for (_, pw0, pw2, dev_t) in structs]
    out_s = torch.zeros(R + 7, 7, device=dev)
    T.heads_fwd(pooled.to(dev), st_structs[0], st_structs[1], out_s, slot=perm.to(dev))
    assert rel_err(out_s.cpu()[perm.long()], want.detach()) < 2e-6 and float(out_s.cpu()[untouched].abs().max()) == 0.0
    out_s2 = torch.zeros(R + 7, 7, device=dev)
    T.heads_fwd(pooled.to(dev), st_structs[0], st_structs[1], out_s2, slot=perm.to(dev))
    assert torch.equal(out_s2, out_s)
    dout = torch.randn(R, 7, generator=g) * 1e-3
    want.backward(dout.double())
    da0, da1, dx = T.heads_bwd(dout.to(dev), structs[0][0], structs[1][0], saved, gstructs[0], gstructs[1], G, L)
    for hd in range(2):
        assert rel_err(da0[hd], inter[hd][0].grad) < 4e-6 and rel_err(da1[hd], inter[hd][2].grad) < 4e-6, hd
        for got_g, leaf in zip(grads[hd], leaves[hd]):
            assert rel_err(got_g, leaf.grad) < 4e-6, hd
    want_dx = (x.grad / L).repeat_interleave(L, dim=0)
    assert dx.shape == (R * L, C) and rel_err(dx, want_dx) < 4e-6
