"""Fused MLP-denoiser chain kernel (csrc/chain.hip) against a float64 restatement of the same math with the kernel's
bf16 storage points (weights, x_t, u, h, pred, dz are bf16 tensors).  Tolerance: 3e-2 of each tensor's max magnitude
(bf16 has 8 significant bits; the chain accumulates a few roundings per layer).  GPU only."""
import pytest
import torch

from inferbiomechanics_amd._tuning import tuning as TU

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd import hip as h
    h.lib()
    return h


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def close(actual, expected, rtol, what):
    a = actual.detach().to("cpu", torch.float64)
    e = expected.detach().to("cpu", torch.float64)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - e).abs().max().item()
    ref = max(e.abs().max().item(), 1e-30)
    assert err <= rtol * ref, f"{what}: max err {err:.3e} > {rtol:.1e} * {ref:.3e}"


def bf(x):
    return x.to(BF).to(torch.float64)


from tests.chain_restatement import chain_restatement  # noqa: E402


CASES = [(256, 50, 300, 512, 2),     # BASELINE configs[1]: 12800 tokens -> 256 workgroups of 50
         (7, 50, 300, 512, 2),       # 350 tokens: ragged last workgroup
         (300, 50, 300, 512, 3),     # 15000 tokens: 64-token panels, three blocks
         (5, 13, 48, 128, 1),        # small widths, one block
         (3, 50, 100, 128, 2),
         (9, 20, 300, 256, 2),
         (6, 50, 40, 512, 2), (6, 50, 100, 512, 1), (6, 50, 200, 512, 2), (6, 50, 384, 512, 2), (6, 50, 512, 512, 2)]


@pytest.mark.parametrize("B,T,D,H,L", CASES)
def test_chain_matches_float64_restatement(hip, B, T, D, H, L):
    assert hip.mlp_chain_supported(D, H, L)
    M = B * T
    x0 = rnd((B, T, D), 1).to(BF)
    eps = rnd((B, T, D), 2).to(BF)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3))
    t[0] = 0
    t[-1] = 999
    tabs = R.schedule_tables()
    sab, s1m = tabs["sqrt_ab"].to(torch.float32), tabs["sqrt_1mab"].to(torch.float32)
    e = rnd((B, L * H), 4, 0.5).to(BF)
    dims = [D] + [H] * L
    W = [rnd((dims[i + 1], dims[i]), 10 + i, dims[i] ** -0.5).to(BF) for i in range(L)] + [rnd((D, H), 30, H ** -0.5).to(BF)]
    bias = [rnd((H,), 40 + i, 0.1).to(torch.float32) for i in range(L)] + [rnd((D,), 49, 0.1).to(torch.float32)]
    gamma = [(1.0 + rnd((H,), 50 + i, 0.2)).to(torch.float32) for i in range(L)]
    beta = [rnd((H,), 60 + i, 0.1).to(torch.float32) for i in range(L)]

    # ---- float64 restatement (autograd) with the kernel's storage points; the SAME function with the rounding switched off is
    # held to oracle/ref_cpu.denoiser_mlp_forward + autograd on CPU (tests/test_oracle_golden.py::test_chain_restatement_...)
    xt_ref, us, hs, pred, loss, g64, b64 = chain_restatement(x0, eps, t, sab, s1m, e, W, bias, gamma, beta, bf)
    # ---- kernel
    d = lambda x: x.to(DEV)
    packed = torch.zeros(hip.mlp_chain_packed_elems(D, H, L), dtype=BF, device=DEV)
    hip.mlp_chain_pack([d(w) for w in W], packed, D, H)
    Dp = (D + 7) // 8 * 8
    xt = torch.zeros(M, Dp, dtype=BF, device=DEV)[:, :D]
    dpred = torch.zeros(M, Dp, dtype=BF, device=DEV)[:, :D]
    u_k = [torch.zeros(M, H, dtype=BF, device=DEV) for _ in range(L)]
    h_k = [torch.zeros(M, H, dtype=BF, device=DEV) for _ in range(L)]
    dz_k = [torch.zeros(M, H, dtype=BF, device=DEV) for _ in range(L)]
    nwg = hip.mlp_chain_workgroups(M)
    Wd = hip.mlp_chain_partial_width(D, H, L)
    part = torch.zeros(nwg, Wd, dtype=torch.float32, device=DEV)
    window_panels = hip.mlp_chain_rows_per_workgroup(M) == T and nwg == B
    de_lp = torch.zeros(B, L * H, dtype=BF, device=DEV) if window_panels else None
    hip.mlp_chain_train(d(x0), d(eps), d(t), d(sab), d(s1m), d(e), packed, [d(b) for b in bias], [d(g) for g in gamma],
                        [d(b) for b in beta], xt, u_k, h_k, dz_k, dpred, part, T, de_lp=de_lp)
    # every small gradient + the loss in one multi-segment reduction
    out = torch.zeros(1, dtype=torch.float32, device=DEV)
    dg_k = [torch.zeros(H, device=DEV) for _ in range(L)]
    db_k = [torch.zeros(H, device=DEV) for _ in range(L)]
    dbl_k = [torch.zeros(H, device=DEV) for _ in range(L)]
    dbl2 = torch.zeros(L * H, device=DEV)
    dbh_k = torch.zeros(D, device=DEV)
    segs = []
    for i in range(L):
        segs += [(3 * i * H, H, dg_k[i], None, 1.0), (3 * i * H + H, H, db_k[i], None, 1.0),
                 (3 * i * H + 2 * H, H, dbl_k[i], dbl2[i * H:(i + 1) * H], 1.0)]
    segs += [(3 * L * H, D, dbh_k, None, 1.0), (Wd - 4, 1, out, None, 1.0 / (M * D))]
    hip.colsum_segments(part, nwg, segs)
    torch.cuda.synchronize()

    tol = 3e-2
    close(xt, xt_ref, 1e-2, "x_t")
    for i in range(L):
        close(u_k[i], us[i], tol, f"u{i}")
        close(h_k[i], hs[i], tol, f"h{i}")
    if L <= 2:
        # the product path hands over NO u buffers (the pre-activations never leave the kernel's registers): every
        # output must be bitwise what the launch with u buffers produced
        outs = [xt.clone(), dpred.clone(), part.clone()] + [a.clone() for a in h_k + dz_k]
        for a in [xt, dpred, part] + h_k + dz_k:
            a.zero_()
        hip.mlp_chain_train(d(x0), d(eps), d(t), d(sab), d(s1m), d(e), packed, [d(b) for b in bias], [d(g) for g in gamma],
                            [d(b) for b in beta], xt, None, h_k, dz_k, dpred, part, T, de_lp=de_lp)
        torch.cuda.synchronize()
        for a, b_ in zip([xt, dpred, part] + h_k + dz_k, outs):
            assert torch.equal(a, b_)
    assert abs(out.item() - loss.item()) <= 1e-2 * abs(loss.item()), (out.item(), loss.item())
    close(dpred, pred.grad, tol, "dpred")
    for i in range(L - 1, -1, -1):
        close(dz_k[i], us[i].grad, tol, f"dz{i}")
        close(dg_k[i], g64[i].grad, tol, f"dgamma{i}")
        close(db_k[i], b64[i].grad, tol, f"dbeta{i}")
        close(dbl_k[i], us[i].grad.sum(0), tol, f"dbias{i}")
        close(dbl2[i * H:(i + 1) * H], us[i].grad.sum(0), tol, f"dbias{i} (second destination)")
        if de_lp is not None:
            close(de_lp[:, i * H:(i + 1) * H], us[i].grad.reshape(B, T, H).sum(1), tol, f"de{i}")
    close(dbh_k, pred.grad.sum(0), tol, "dbias head")


@pytest.mark.parametrize("B,temb,hid,out", [(256, 128, 512, 1024), (70, 32, 128, 256), (3, 128, 512, 512), (1000, 128, 512, 1024), (37, 128, 512, 1024)])
def test_time_mlp_fwd_matches_float64(hip, B, temb, hid, out):
    assert hip.time_mlp_fwd_supported(temb, hid, out)
    table = R.timestep_embedding(torch.arange(1000), temb).to(torch.float32)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(1))
    t[0], t[-1] = 0, 999
    w1 = rnd((hid, temb), 2, temb ** -0.5).to(BF)
    w2 = rnd((out, hid), 3, hid ** -0.5).to(BF)
    b1 = rnd((hid,), 4, 0.1).to(torch.float32)
    b2 = rnd((out,), 5, 0.1).to(torch.float32)
    d = lambda x: x.to(DEV)
    s = torch.zeros(B, temb, dtype=BF, device=DEV)
    zu = torch.zeros(B, hid, dtype=BF, device=DEV)
    u = torch.zeros(B, hid, dtype=BF, device=DEV)
    e = torch.zeros(B, out, dtype=BF, device=DEV)
    hip.time_mlp_fwd(d(table), d(t), d(w1), d(b1), d(w2), d(b2), s, zu, u, e)
    torch.cuda.synchronize()
    s_ref = table[t].to(BF)
    assert torch.equal(s.cpu(), s_ref)                       # a gather + one rounding: bit-exact
    z_ref = bf(s_ref.to(torch.float64) @ w1.to(torch.float64).T + b1.to(torch.float64))
    u_ref = bf(z_ref * torch.sigmoid(z_ref))
    e_ref = u_ref @ w2.to(torch.float64).T + b2.to(torch.float64)
    close(zu, z_ref, 1e-2, "zu")
    close(u, u_ref, 1e-2, "u")
    close(e, e_ref, 1e-2, "e")


def test_deferred_wgrad_slabs_match_fused_wgrad(hip):
    """ib_linear_wgrad_slabs + ONE ib_slab_reduce_multi == ib_linear_wgrad: bitwise where both take the split-M kernel (same
    slabs, same order); for short reductions (<= 1024 rows) ib_linear_wgrad is the one-pass kernel, which adds the same
    products in another order -- equal within fp32 rounding"""
    shapes = [(12800, 512, 300), (12800, 300, 512), (256, 1024, 512), (256, 512, 128), (100, 64, 32)]
    items, refs = [], []
    for j, (M, N, K) in enumerate(shapes):
        dz = rnd((M, N), 70 + j).to(BF).to(DEV)
        x = rnd((M, K), 80 + j).to(BF).to(DEV)
        ws = torch.zeros(max(hip.linear_wgrad_workspace_bytes(M, N, K), 4), dtype=torch.uint8, device=DEV)
        ref = torch.zeros(N, K, device=DEV)
        hip.linear_wgrad(dz, x, ref, ws)
        ws2 = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=DEV)
        n = hip.linear_wgrad_slabs(dz, x, ws2)
        items.append((ws2, n, torch.full((N, K), 7.0, device=DEV)))
        refs.append(ref)
    hip.slab_reduce_multi(items)
    torch.cuda.synchronize()
    for (M, N, K), (ws2, n, dw), ref in zip(shapes, items, refs):
        if M > 1024:
            assert torch.equal(dw, ref)
        else:
            close(dw, ref.double(), 2e-6, f"short reduction {M}x{N}x{K}")
    before = [dw.clone() for _, _, dw in items]
    hip.slab_reduce_multi(items, accumulate=True)
    torch.cuda.synchronize()
    for (ws2, n, dw), b in zip(items, before):
        assert torch.equal(dw, b + b)


def test_grouped_wgrad_launch_matches_single_launches(hip):
    """several dW problems in one launch: each gets a share of the launch's workgroup budget (fewer, longer split-M slices
    than a single launch would take), so the slab COUNT differs but the reduced gradient is the same sum -- equal to
    the single launches within fp32 rounding, and bitwise reproducible run to run"""
    shapes = [(12800, 304, 512), (12800, 512, 512), (12800, 512, 304), (256, 1024, 512)]
    probs, probs2, singles = [], [], []
    for j, (M, N, K) in enumerate(shapes):
        dz = rnd((M, N), 170 + j).to(BF).to(DEV)
        x = rnd((M, K), 180 + j).to(BF).to(DEV)
        nb = int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K))
        ws1, ws2, ws3 = (torch.zeros(nb, dtype=torch.uint8, device=DEV) for _ in range(3))
        singles.append((ws1, hip.linear_wgrad_slabs(dz, x, ws1)))
        probs.append((dz, x, ws2))
        probs2.append((dz, x, ws3))
    ns = hip.linear_wgrad_slabs_multi(probs)
    ns2 = hip.linear_wgrad_slabs_multi(probs2)
    torch.cuda.synchronize()
    assert ns is not None and ns == ns2 and all(a <= b for (M, _, _), a, (_, b) in zip(shapes, ns, singles) if M > 1024)
    for (M, N, K), (ws1, n1), (dz, x, ws2), (_, _, ws3), n in zip(shapes, singles, probs, probs2, ns):
        assert torch.equal(ws2, ws3)
        one = torch.empty(N, K, device=DEV)
        grp = torch.empty(N, K, device=DEV)
        hip.slab_reduce_multi([(ws1, n1, one), (ws2, n, grp)])
        close(grp, one.double(), 2e-6, f"grouped vs single {M}x{N}x{K}")
        close(grp, dz.double().T @ x.double(), 1e-5, "vs float64")
    # a ragged reduction length does not qualify for the ring kernel: the caller is told to fall back
    dz = rnd((100, 64), 1).to(BF).to(DEV)
    x = rnd((100, 32), 2).to(BF).to(DEV)
    ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(100, 64, 32)), dtype=torch.uint8, device=DEV)
    assert hip.linear_wgrad_slabs_multi([(dz, x, ws), (dz, x, ws.clone())]) is None


def test_colsum_segments_ragged(hip):
    part = rnd((37, 132), 90).to(torch.float32).to(DEV)
    a, b, c = torch.zeros(70, device=DEV), torch.ones(3, device=DEV), torch.zeros(1, device=DEV)
    a2 = torch.zeros(70, device=DEV)
    hip.colsum_segments(part, 37, [(0, 70, a, a2, 1.0), (72, 3, b, None, 2.0), (128, 1, c, None, 0.5)])
    torch.cuda.synchronize()
    p64 = part.double().cpu()
    close(a, p64[:, :70].sum(0), 1e-5, "seg0")
    close(a2, p64[:, :70].sum(0), 1e-5, "seg0 copy")
    close(b, 2.0 * p64[:, 72:75].sum(0), 1e-5, "seg1")
    close(c, 0.5 * p64[:, 128:129].sum(0), 1e-5, "seg2")
    hip.colsum_segments(part, 37, [(72, 3, b, None, 2.0)], accumulate=True)
    torch.cuda.synchronize()
    close(b, 4.0 * p64[:, 72:75].sum(0), 1e-5, "seg1 accumulated")


def test_chain_trainer_tracks_per_op_trainer():
    """HipTrainer with the chain kernel vs the per-op launch plan (IB_NO_CHAIN=1): same model, same batches; the two
    bf16 trajectories may differ by rounding only."""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP

    def run(no_chain):
        if no_chain:
            TU.no_chain = True
        else:
            TU.no_chain = False
        try:
            torch.manual_seed(0)
            m = DiffusionMLP(300, [512, 512], compute_dtype=BF).to(DEV)
            sd = R.det_params(R.denoiser_mlp_param_shapes(300, [512, 512]), seed0=5.0)
            m.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()})
            p0 = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
            tr = HipTrainer(m, "diffusion", "sgd", 5e-2, use_graph=not no_chain)
            losses = []
            for s in range(12):
                g = torch.Generator().manual_seed(100 + s)
                x0 = torch.randn(32, 50, 300, generator=g).to(DEV)
                eps = torch.randn(32, 50, 300, generator=g).to(DEV)
                t = torch.randint(0, 1000, (32,), generator=g).to(DEV)
                tr.step((x0, t, eps))
                losses.append(tr.loss_value())
            return losses, {k: v.detach().float().cpu() - p0[k] for k, v in m.state_dict().items()}
        finally:
            TU.no_chain = False

    la, pa = run(False)
    lb, pb = run(True)
    assert la[-1] < la[0]
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-2 * abs(y), (la, lb)
    for k in pa:                       # pa / pb: parameter MOVEMENT over the 12 SGD steps
        dlt = (pa[k] - pb[k]).abs().max().item()
        assert dlt <= 0.1 * max(pb[k].abs().max().item(), 1e-6), (k, dlt, pb[k].abs().max().item())


def test_optimizer_sources_bitwise_equal_to_step_reduce():
    """single-GPU steps let the optimizer sum the weight-gradient slabs and the partial rows itself
    (ib_optim_step_sources); the result must be BITWISE the parameters of the separate ib_step_reduce + ib_optim_step"""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP

    def run(fused):
        if fused:
            TU.no_opt_fuse = False
        else:
            TU.no_opt_fuse = True
        try:
            m = DiffusionMLP(300, [512, 512], compute_dtype=BF).to(DEV)
            sd = R.det_params(R.denoiser_mlp_param_shapes(300, [512, 512]), seed0=7.0)
            m.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()})
            tr = HipTrainer(m, "diffusion", "adam", 1e-3, use_graph=True)
            losses = []
            for s_ in range(6):
                g = torch.Generator().manual_seed(300 + s_)
                x0 = torch.randn(64, 50, 300, generator=g).to(DEV)
                eps = torch.randn(64, 50, 300, generator=g).to(DEV)
                t = torch.randint(0, 1000, (64,), generator=g).to(DEV)
                tr.step((x0, t, eps))
                losses.append(tr.loss_value())
            return losses, {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        finally:
            TU.no_opt_fuse = False

    la, pa = run(True)
    lb, pb = run(False)
    assert la == lb, (la, lb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


def test_zero_copy_batches_equal_staged_batches():
    """device-resident bf16 batches are consumed in place through the input pointer slots (ib_set_ptrs); the result must
    be exactly what the staging-copy path produces, including after the graph has been captured"""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP

    def run(zero_copy):
        if zero_copy:
            TU.no_zero_copy = False
        else:
            TU.no_zero_copy = True
        try:
            m = DiffusionMLP(300, [512, 512], compute_dtype=BF).to(DEV)
            sd = R.det_params(R.denoiser_mlp_param_shapes(300, [512, 512]), seed0=9.0)
            m.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()})
            tr = HipTrainer(m, "diffusion", "rmsprop", 1e-4, use_graph=True)
            batches = []
            for s_ in range(4):
                g = torch.Generator().manual_seed(500 + s_)
                batches.append((torch.randn(32, 50, 300, generator=g).to(DEV, BF),
                                torch.randint(0, 1000, (32,), generator=g).to(DEV),
                                torch.randn(32, 50, 300, generator=g).to(DEV, BF)))
            losses = []
            for s_ in range(8):
                tr.step(batches[s_ % 4])
                losses.append(tr.loss_value())
            assert tr._zero_copy                      # the chain path is active in both runs
            return losses
        finally:
            TU.no_zero_copy = False

    la, lb = run(True), run(False)
    assert la == lb, (la, lb)
    assert len(set(la)) > 4                           # different batches really were consumed


@pytest.mark.parametrize("M,N,K,act", [(256, 1024, 512, "silu"), (100, 512, 128, "silu"), (16, 128, 16, "none"),
                                       (256, 256, 64, "relu"), (7, 1024, 512, "elu")])
def test_linear_dgrad_skinny_with_bias_sums(hip, M, N, K, act):
    """few-row dgrad (the time-embedding MLP's hidden layer): dx = (dz w) * act'(aux) and dbias = column sums of the STORED
    dx from one launch; ragged row counts (partial / missing row tiles), accumulate, strided operands"""
    dz_w = rnd((M, N + 8), 1).to(BF).to(DEV)
    dz = dz_w[:, :N]                                           # row pitch N + 8
    w = rnd((N, K), 2, N ** -0.5).to(BF).to(DEV)
    aux = rnd((M, K), 3).to(BF).to(DEV)
    dx = torch.full((M, K), 7.0, dtype=BF, device=DEV)
    db = torch.full((K,), 3.0, dtype=torch.float32, device=DEV)
    assert hip.linear_dgrad_skinny(dz, w, dx, act_below=act, aux=aux if act != "none" else None, dbias=db)
    a = aux.double().cpu()
    fac = {"none": torch.ones_like(a), "relu": (a > 0).double(), "elu": torch.where(a > 0, torch.ones_like(a), a + 1),
           "silu": torch.sigmoid(a) * (1 + a * (1 - torch.sigmoid(a)))}[act]
    exp = (dz.double().cpu() @ w.double().cpu()) * fac
    close(dx, exp, 1e-2, "dx")
    close(db, dx.double().sum(0), 1e-5, "dbias = column sums of the stored dx")
    # the tiled kernel gives the same tensor within bf16 rounding
    ref = torch.empty_like(dx)
    hip.linear_dgrad(dz, w, ref, act_below=act, aux=aux if act != "none" else None)
    close(dx, ref.double(), 1e-2, "vs tiled dgrad")
    hip.linear_dgrad_skinny(dz, w, dx, act_below=act, aux=aux if act != "none" else None, dbias=db, accumulate=True)
    close(db, 2 * dx.double().sum(0), 1e-5, "accumulate")
    # shapes outside the kernel's range are refused (the caller falls back), never mis-computed
    assert not hip.linear_dgrad_skinny(torch.zeros(300, N, dtype=BF, device=DEV), w, torch.zeros(300, K, dtype=BF, device=DEV))
    assert not hip.linear_dgrad_skinny(torch.zeros(M, 96, dtype=BF, device=DEV), torch.zeros(96, K, dtype=BF, device=DEV), dx)


@pytest.mark.parametrize("B,temb,hid,out", [(256, 128, 512, 1024), (70, 32, 128, 256), (3, 128, 512, 512), (130, 128, 512, 2048)])
def test_time_mlp_bwd_matches_float64(hip, B, temb, hid, out):
    """ib_time_mlp_bwd (one launch: few-row dgrad through time_mlp.2 * silu'(zu), the weight gradient of time_mlp.0 and
    its bias gradient as per-64-window slabs) against float64 on the same bf16 operands; the slabs are summed here in slab
    order, as the optimizer / ib_step_reduce do.  Ragged batches (rows beyond B contribute nothing)."""
    assert hip.time_mlp_bwd_supported(temb, hid, out)
    de = rnd((B, out), 1, 0.3).to(BF)
    w2 = rnd((out, hid), 2, hid ** -0.5).to(BF)
    zu = rnd((B, hid), 3, 1.5).to(BF)
    s = rnd((B, temb), 4, 0.7).to(BF)
    n = hip.time_mlp_bwd_slab_count(B)
    assert n == (B + 63) // 64
    sw = torch.full((n, hid, temb), float("nan"), device=DEV)
    sb = torch.full((n, hid), float("nan"), device=DEV)
    d = lambda x: x.to(DEV)
    hip.time_mlp_bwd(d(de), d(w2), d(zu), d(s), sw, sb)
    torch.cuda.synchronize()
    z = zu.to(torch.float64)
    sg = torch.sigmoid(z)
    dzu = (de.to(torch.float64) @ w2.to(torch.float64)) * (sg * (1 + z * (1 - sg)))
    dzu_b = bf(dzu)                                   # the kernel rounds dzu to bf16 before both products
    close(sw.sum(0), dzu_b.T @ s.to(torch.float64), 2e-2, "dW1")
    close(sb.sum(0), dzu_b.sum(0), 2e-2, "db1")
    # deterministic: a second launch reproduces every slab bit for bit
    sw2, sb2 = torch.zeros_like(sw), torch.zeros_like(sb)
    hip.time_mlp_bwd(d(de), d(w2), d(zu), d(s), sw2, sb2)
    torch.cuda.synchronize()
    assert torch.equal(sw, sw2) and torch.equal(sb, sb2)


def test_time_mlp_bwd_rides_in_the_grouped_weight_gradient_launch(hip):
    """ib_linear_wgrad_slabs_multi_tb: the MLP denoiser's four weight-gradient problems (M = 12800 token rows; time_mlp.2 with
    M = 256) and, as extra workgroups of the SAME launch, the time-MLP hidden layer's backward (8 waves per workgroup there:
    the reduction split eight ways).  Every output against float64; the GEMM slabs bitwise equal to the launch without rider."""
    M, T, D, H, B, temb, hid = 12800, 50, 300, 512, 256, 128, 512
    out = 2 * H
    mk = lambda shape, seed, sc=1.0: rnd(shape, seed, sc).to(BF).to(DEV)
    dpred, h1, dz1, h0, dz0 = mk((M, 304), 1)[:, :D], mk((M, H), 2), mk((M, H), 3), mk((M, H), 4), mk((M, H), 5)
    xt = mk((M, 304), 6)[:, :D]
    de, u_t = mk((B, out), 7, 0.3), mk((B, hid), 8)
    w2, zu, s = mk((out, hid), 9, hid ** -0.5), mk((B, hid), 10, 1.5), mk((B, temb), 11, 0.7)
    probs = [(dpred, h1), (dz1, h0), (dz0, xt), (de, u_t)]
    ws = lambda: [torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(a.shape[0], a.shape[1], b.shape[1])),
                              dtype=torch.uint8, device=DEV) for a, b in probs]
    n = hip.time_mlp_bwd_slab_count(B)
    sw = torch.full((n, hid, temb), float("nan"), device=DEV)
    sb = torch.full((n, hid), float("nan"), device=DEV)
    w_a, w_b = ws(), ws()
    ns_a = hip.linear_wgrad_slabs_multi([(a, b, w) for (a, b), w in zip(probs, w_a)], time_bwd=(de, w2, zu, s, sw, sb))
    assert ns_a is not None, "the combined launch must take the headline shape"
    ns_b = hip.linear_wgrad_slabs_multi([(a, b, w) for (a, b), w in zip(probs, w_b)])
    torch.cuda.synchronize()
    assert ns_a == ns_b
    for (a, b), wa, wb, k in zip(probs, w_a, w_b, ns_a):
        N, K = a.shape[1], b.shape[1]
        sa = wa[:k * N * K * 4].view(torch.float32).view(k, N, K)
        assert torch.equal(sa, wb[:k * N * K * 4].view(torch.float32).view(k, N, K))
        close(sa.sum(0), a.double().T @ b.double(), 2e-2, f"dW [{N},{K}]")
    z = zu.double().cpu()
    sg = torch.sigmoid(z)
    dzu_b = bf((de.double().cpu() @ w2.double().cpu()) * (sg * (1 + z * (1 - sg))))
    close(sw.sum(0), dzu_b.T @ s.double().cpu(), 2e-2, "rider dW1")
    close(sb.sum(0), dzu_b.sum(0), 2e-2, "rider db1")
