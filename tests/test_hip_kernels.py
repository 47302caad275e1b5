"""Kernel-level parity: every C-ABI entry point against float64 restatements (oracle / plain math)
on seeded inputs.  Tolerances: fp32 mode <= 1e-3 relative (north_star; in practice ~1e-6), bf16 mode
<= 3e-2 relative to the tensor's max magnitude (bf16 storage has 8 significant bits).  Index / table
ops (gathers, schedule tables) are bit-exact.  GPU only."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd import hip as h
    h.lib()
    return h


DEV = "cuda"
TOL = {torch.float32: 1e-3, torch.bfloat16: 3e-2}
TIGHT = {torch.float32: 2e-5, torch.bfloat16: 3e-2}


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def close(actual, expected, rtol, what=""):
    a = actual.detach().to("cpu", torch.float64)
    e = expected.detach().to("cpu", torch.float64)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - e).abs().max().item()
    ref = max(e.abs().max().item(), 1e-30)
    assert err <= rtol * ref, f"{what}: max err {err:.3e} > {rtol:.1e} * {ref:.3e}"


def test_tr16_transposing_read_layout(hip):
    """ds_read_b64_tr_b16 must deliver in[8*(lane/16)+q][lane%16] as element q (what gemm.hip assumes)."""
    img = torch.arange(64 * 16, dtype=torch.int16).reshape(64, 16)
    out = torch.zeros(64, 4, dtype=torch.int16, device=DEV)
    hip.selftest_tr16(img.to(DEV), out)
    got = out.cpu()
    exp = torch.empty(64, 4, dtype=torch.int16)
    for lane in range(64):
        for q in range(4):
            exp[lane, q] = img[8 * (lane // 16) + q, lane % 16]
    assert torch.equal(got, exp), f"tr16 layout differs:\n{got[:20]}\nvs\n{exp[:20]}"


SHAPES = [(4, 512, 1470), (37, 300, 512), (256, 512, 128), (640, 512, 300), (130, 129, 67), (1, 5, 3),
          (300, 1536, 512), (2560, 512, 300)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_fwd(hip, dtype, M, N, K):
    x = rnd((M, K), 1, 1.0, dtype)
    w = rnd((N, K), 2, 1.0 / math.sqrt(K), dtype)
    b = rnd((N,), 3, 0.1)
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    hip.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y)
    exp = x.double() @ w.double().T + b.double()
    close(y, exp, TIGHT[dtype], "linear_fwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid", "silu"])
def test_linear_fwd_epilogue(hip, dtype, act):
    B, T, K, N = 5, 7, 44, 72
    M = B * T
    x = rnd((M, K), 1, 1.0, dtype)
    w = rnd((N, K), 2, 1.0 / math.sqrt(K), dtype)
    b = rnd((N,), 3, 0.1)
    e_wide = rnd((B, N + 8), 4, 0.5, dtype)
    e = e_wide[:, 4:4 + N]                               # strided row-broadcast operand
    pm = rnd((T, N), 5, 0.5, dtype)
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    z = torch.empty(M, N, dtype=dtype, device=DEV)
    hip.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y, act=act, z=z, add_div=e_wide.to(DEV)[:, 4:4 + N], add_mod=pm.to(DEV), seg=T)
    zz = x.double() @ w.double().T + b.double()
    zz = (zz.reshape(B, T, N) + e.double()[:, None, :] + pm.double()[None, :, :]).reshape(M, N)
    close(z, zz, TIGHT[dtype], "pre-activation")
    close(y, R.act(act, zz), TIGHT[dtype], "activation " + act)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4, 300, 512), (37, 512, 1470), (640, 300, 512), (130, 129, 67), (1, 5, 3),
                                   (2560, 512, 300)])
@pytest.mark.parametrize("act", ["none", "sigmoid", "silu", "relu", "tanh"])
def test_linear_dgrad(hip, dtype, M, N, K, act):
    dz = rnd((M, N), 1, 1.0, dtype)
    w = rnd((N, K), 2, 1.0 / math.sqrt(N), dtype)
    aux = rnd((M, K), 3, 1.0, dtype)
    if act in ("sigmoid",):
        aux = torch.sigmoid(aux.float()).to(dtype)
    if act == "tanh":
        aux = torch.tanh(aux.float()).to(dtype)
    add = rnd((M, K), 4, 1.0, dtype)
    dx = torch.empty(M, K, dtype=dtype, device=DEV)
    hip.linear_dgrad(dz.to(DEV), w.to(DEV), dx, act_below=act, aux=aux.to(DEV) if act != "none" else None,
                     addend=add.to(DEV))
    exp = dz.double() @ w.double()
    a = aux.double()
    if act == "relu":
        exp = exp * (a > 0)
    elif act == "tanh":
        exp = exp * (1 - a * a)
    elif act == "sigmoid":
        exp = exp * a * (1 - a)
    elif act == "silu":
        s = torch.sigmoid(a)
        exp = exp * (s * (1 + a * (1 - s)))
    exp = exp + add.double()
    close(dx, exp, TIGHT[dtype], "dgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4, 300, 512), (64, 512, 1470), (12800, 512, 300), (130, 129, 67), (1, 5, 3),
                                   (2500, 300, 512), (700, 1536, 512)])
def test_linear_wgrad(hip, dtype, M, N, K):
    dz = rnd((M, N), 1, 1.0, dtype)
    x = rnd((M, K), 2, 1.0, dtype)
    dw = torch.full((N, K), 7.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(max(hip.linear_wgrad_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV)
    hip.linear_wgrad(dz.to(DEV), x.to(DEV), dw, ws)
    exp = dz.double().T @ x.double()
    close(dw, exp, 2e-5 if dtype == torch.float32 else 1e-5 + 2e-2, "wgrad")
    hip.linear_wgrad(dz.to(DEV), x.to(DEV), dw, ws, accumulate=True)
    close(dw, 2 * exp, 2e-5 if dtype == torch.float32 else 2e-2, "wgrad accumulate")
    # strided destination (a column block of a wider gradient matrix)
    wide = torch.zeros(N, K + 30, dtype=torch.float32, device=DEV)
    hip.linear_wgrad(dz.to(DEV), x.to(DEV), wide[:, :K], ws)
    close(wide[:, :K], exp, 2e-5 if dtype == torch.float32 else 2e-2, "wgrad strided")
    assert float(wide[:, K:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_is_bitwise_reproducible(hip, dtype):
    M, N, K = 3000, 300, 512
    dz, x = rnd((M, N), 1, 1.0, dtype).to(DEV), rnd((M, K), 2, 1.0, dtype).to(DEV)
    ws = torch.empty(hip.linear_wgrad_workspace_bytes(M, N, K), dtype=torch.uint8, device=DEV)
    a = torch.empty(N, K, dtype=torch.float32, device=DEV)
    b = torch.empty(N, K, dtype=torch.float32, device=DEV)
    hip.linear_wgrad(dz, x, a, ws)
    hip.linear_wgrad(dz, x, b, ws)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(10, 512), (257, 108), (33, 2048), (5, 30), (100, 300)])
@pytest.mark.parametrize("act,use_res", [("none", True), ("silu", False), ("none", False)])
def test_layernorm_fwd_bwd(hip, dtype, M, N, act, use_res):
    x = rnd((M, N), 1, 1.5, dtype)
    res = rnd((M, N), 2, 1.0, dtype) if use_res else None
    gamma = (1 + rnd((N,), 3, 0.1)).float()
    beta = rnd((N,), 4, 0.1).float()
    dy = rnd((M, N), 5, 1.0, dtype)
    xd = x.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if use_res else None
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v = R.act(act, xd) + (rd if use_res else 0)
    yd = R.layer_norm(v, gd, bd)
    yd.backward(dy.double())
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    mean = torch.empty(M, dtype=torch.float32, device=DEV)
    rstd = torch.empty(M, dtype=torch.float32, device=DEV)
    hip.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), y, mean, rstd, res=res.to(DEV) if use_res else None, act=act)
    close(y, yd, TIGHT[dtype], "ln fwd")
    close(mean, v.mean(-1), 1e-5, "ln mean")
    dx = torch.empty(M, N, dtype=dtype, device=DEV)
    dres = torch.empty(M, N, dtype=dtype, device=DEV) if (use_res or act != "none") else None
    dg = torch.empty(N, dtype=torch.float32, device=DEV)
    db = torch.empty(N, dtype=torch.float32, device=DEV)
    ws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    hip.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dx, dg, db, ws,
                      res=res.to(DEV) if use_res else None, dres=dres, act=act)
    close(dx, xd.grad, TOL[dtype] if dtype == torch.bfloat16 else 1e-4, "ln dx")
    if use_res:
        close(dres, rd.grad, TOL[dtype] if dtype == torch.bfloat16 else 1e-4, "ln dres")
    close(dg, gd.grad, 1e-4 if dtype == torch.float32 else 2e-2, "ln dgamma")
    close(db, bd.grad, 1e-4 if dtype == torch.float32 else 2e-2, "ln dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,H,dh", [(2, 50, 8, 64), (1, 200, 8, 64), (3, 10, 3, 36), (2, 37, 4, 32), (1, 1, 1, 8)])
def test_attention_fwd_bwd(hip, dtype, B, T, H, dh):
    d = H * dh
    qkv = rnd((B, T, 3 * d), 1, 1.0, dtype)
    dout = rnd((B, T, d), 2, 1.0, dtype)
    q64 = qkv.double().requires_grad_(True)
    q, k, v = q64[..., :d], q64[..., d:2 * d], q64[..., 2 * d:]
    sp = lambda t: t.reshape(B, T, H, dh).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    o = (p @ sp(v)).transpose(1, 2).reshape(B, T, d)
    o.backward(dout.double())
    out = torch.empty(B, T, d, dtype=dtype, device=DEV)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=DEV)
    hip.attention_fwd(qkv.to(DEV), out, lse, H)
    close(out, o, TIGHT[dtype], "attention out")
    close(lse, torch.logsumexp(s, dim=-1), 1e-4 if dtype == torch.float32 else 2e-2, "lse")
    dqkv = torch.empty(B, T, 3 * d, dtype=dtype, device=DEV)
    hip.attention_bwd(qkv.to(DEV), out, dout.to(DEV), lse, dqkv, H)
    close(dqkv, q64.grad, 1e-4 if dtype == torch.float32 else 4e-2, "attention dqkv")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_colsum(hip, dtype):
    B, T, N = 6, 50, 300
    x = rnd((B * T, N + 4), 1, 1.0, dtype).to(DEV)[:, 2:2 + N]
    out = torch.zeros(B, N, dtype=torch.float32, device=DEV)
    hip.segment_colsum(x, out, seg=T, mode=0)
    close(out, x.double().reshape(B, T, N).sum(1), 1e-5 if dtype == torch.float32 else 1e-5, "div sum")
    out2 = torch.ones(T, N, dtype=torch.float32, device=DEV)
    hip.segment_colsum(x, out2, seg=T, mode=1, accumulate=True)
    close(out2, x.double().reshape(B, T, N).sum(0) + 1, 1e-5, "mod sum accumulate")
    out3 = torch.zeros(1, N, dtype=torch.float32, device=DEV)
    hip.segment_colsum(x, out3, seg=B * T, mode=0)
    close(out3, x.double().sum(0, keepdim=True), 1e-5, "full colsum")
    # ragged last segment
    out4 = torch.zeros(3, N, dtype=torch.float32, device=DEV)
    hip.segment_colsum(x, out4, seg=128, mode=0)
    xs = x.double()
    close(out4, torch.stack([xs[:128].sum(0), xs[128:256].sum(0), xs[256:].sum(0)]), 1e-5, "ragged")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_concat_keys_and_casts(hip, dtype):
    B, F = 3, 10
    ws = R.input_widths(23, 5)
    ts = [rnd((B, F, w), 10 + i) for i, w in enumerate(ws)]
    out = torch.empty(B, F * sum(ws), dtype=dtype, device=DEV)
    hip.concat_keys([t.to(DEV) for t in ts], out)
    exp = torch.cat(ts, dim=-1).reshape(B, -1).to(dtype)
    assert torch.equal(out.cpu(), exp)                       # pure data movement + RNE cast: bit-exact
    src = rnd((1000,), 5)
    dst = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    hip.cast(src.to(DEV), dst)
    assert torch.equal(dst.cpu(), src.to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_diffusion_elementwise(hip, dtype):
    from inferbiomechanics_amd.diffusion.schedule import DiffusionTables
    tabs = DiffusionTables(torch.device(DEV), 1000, 100, 128)
    otab = R.schedule_tables(1000)
    # tables: float64 host math cast once -> bit-exact vs the oracle's float64 tables cast to fp32
    assert torch.equal(tabs.sqrt_ab.cpu(), otab["sqrt_ab"].to(torch.float32))
    assert torch.equal(tabs.sqrt_1mab.cpu(), otab["sqrt_1mab"].to(torch.float32))
    assert torch.equal(tabs.ddim_t.cpu(), R.ddim_timesteps(1000, 100))
    assert torch.equal(tabs.ddim_coef.cpu(), R.ddim_coeffs(1000, 100).to(torch.float32))
    B, T, D = 5, 7, 12
    t = torch.tensor([0, 1, 499, 999, 250], dtype=torch.int64)
    emb = torch.empty(B, 128, dtype=dtype, device=DEV)
    hip.gather_rows(tabs.temb, t.to(DEV), emb)
    assert torch.equal(emb.cpu(), R.timestep_embedding(t, 128).to(torch.float32).to(dtype))   # bit-exact gather
    x0, eps = rnd((B, T, D), 1, 1.0, dtype), rnd((B, T, D), 2, 1.0, dtype)
    xt = torch.empty(B, T, D, dtype=dtype, device=DEV)
    hip.q_sample(x0.to(DEV), eps.to(DEV), t.to(DEV), tabs.sqrt_ab, tabs.sqrt_1mab, xt)
    close(xt, R.q_sample(x0.double(), t, eps.double(), otab), TIGHT[dtype], "q_sample")
    x = rnd((B, T, D), 3, 1.0, dtype).to(DEV)
    x_ref = x.double().cpu()
    co = R.ddim_coeffs(1000, 100)
    t_out = torch.zeros(B, dtype=torch.int64, device=DEV)
    ctr = torch.tensor([17], dtype=torch.int32, device=DEV)
    hip.ddim_step(x, eps.to(DEV), tabs.ddim_coef, tabs.ddim_t, step_dev=ctr, t_out=t_out)
    close(x, co[17, 0] * x_ref + co[17, 1] * eps.double(), TIGHT[dtype], "ddim step")
    assert torch.equal(t_out.cpu(), torch.full((B,), int(R.ddim_timesteps(1000, 100)[18]), dtype=torch.int64))
    hip.counter_add(ctr, 1)
    assert int(ctr.cpu()) == 18


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mse_loss(hip, dtype):
    n = 12345
    p, t = rnd((n,), 1, 1.0, dtype), rnd((n,), 2, 1.0, dtype)
    res = torch.zeros(1, dtype=torch.float32, device=DEV)
    dp = torch.empty(n, dtype=dtype, device=DEV)
    ws = torch.empty(hip.mse_loss_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    hip.mse_loss(p.to(DEV), t.to(DEV), res, ws, dpred=dp)
    d = p.double() - t.double()
    close(res, (d * d).mean().reshape(1), 1e-5, "mse")
    close(dp, 2 * d / n, TIGHT[dtype], "dmse")


@pytest.mark.parametrize("opt", ["sgd", "adam", "rmsprop", "adagrad", "adadelta", "adamax"])
def test_optimizers_follow_torch_optim_trajectories(hip, golden_dir, opt):
    import os
    g = np.load(os.path.join(golden_dir, "optim_traj.npz"))
    p = R.det_fill((257,), 3, 0.5).float().to(DEV)
    s1 = torch.zeros_like(p)
    s2 = torch.zeros_like(p)
    shadow = torch.zeros(257, dtype=torch.bfloat16, device=DEV)
    for s in range(4):
        grad = R.det_fill((257,), 20 + s, 0.3 * (s + 1)).float().to(DEV)
        hip.optim_step(opt, p, grad * 4.0, s1, s2, lr=1e-2, step=s + 1, grad_scale=0.25, shadow=shadow)
        close(p, torch.from_numpy(g[opt][s]), 2e-5, f"{opt} step {s + 1}")
        assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))
    # device-resident step counter (graph-replay form)
    p2 = R.det_fill((257,), 3, 0.5).float().to(DEV)
    s1.zero_(); s2.zero_()
    ctr = torch.zeros(1, dtype=torch.int32, device=DEV)
    for s in range(4):
        grad = R.det_fill((257,), 20 + s, 0.3 * (s + 1)).float().to(DEV)
        hip.counter_add(ctr, 1)
        hip.optim_step(opt, p2, grad, s1, s2, lr=1e-2, step=0, step_dev=ctr)
    close(p2, torch.from_numpy(g[opt][3]), 2e-5, f"{opt} via device counter")


def test_hipgraph_capture_and_replay(hip):
    x = rnd((64, 32), 1).to(DEV)
    w = rnd((48, 32), 2, 0.2).to(DEV)
    y = torch.zeros(64, 48, device=DEV)
    hip.linear_fwd(x, w, None, y)                         # warm-up outside capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = hip.Graph()
        g.begin()
        hip.linear_fwd(x, w, None, y, act="relu")
        g.end()
        y.zero_()
        e0, e1 = hip.Event(), hip.Event()
        e0.record()
        g.launch()
        e1.record()
        assert e0.elapsed_ms(e1) >= 0.0
    torch.cuda.synchronize()
    close(y, torch.relu(x.double() @ w.double().T), 2e-5, "graph replay")


def test_bad_arguments_fail_loudly(hip):
    x = torch.zeros(4, 8, device=DEV)
    w = torch.zeros(5, 9, device=DEV)
    y = torch.zeros(4, 5, device=DEV)
    with pytest.raises(hip.HipError):
        hip.linear_fwd(x, w, None, y)
    with pytest.raises(hip.HipError):
        hip.linear_fwd(x.cpu(), w, None, y)
    with pytest.raises(hip.HipError):
        hip.linear_fwd(x.to(torch.float16), w.to(torch.float16), None, y.to(torch.float16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_with_per_window_add(hip, dtype):
    """v = silu(x + e[window]) -> LN: the time-embedding add fused into the LayerNorm prologue (fwd + bwd)"""
    B, T, N = 5, 7, 512
    M = B * T
    x = rnd((M, N), 1, 1.5, dtype)
    e_wide = rnd((B, 2 * N), 2, 1.0, dtype)
    e = e_wide[:, N:]                                       # strided slice, as the plans pass it
    gamma = (1 + rnd((N,), 3, 0.1)).float()
    beta = rnd((N,), 4, 0.1).float()
    dy = rnd((M, N), 5, 1.0, dtype)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v = R.act("silu", (xd.reshape(B, T, N) + e.double()[:, None, :]).reshape(M, N))
    yd = R.layer_norm(v, gd, bd)
    yd.backward(dy.double())
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    mean = torch.empty(M, dtype=torch.float32, device=DEV)
    rstd = torch.empty(M, dtype=torch.float32, device=DEV)
    ed = e_wide.to(DEV)[:, N:]
    hip.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), y, mean, rstd, act="silu", add_div=ed, seg=T)
    close(y, yd, TIGHT[dtype], "ln(+e) fwd")
    dx = torch.empty(M, N, dtype=dtype, device=DEV)
    dg = torch.empty(N, dtype=torch.float32, device=DEV)
    db = torch.empty(N, dtype=torch.float32, device=DEV)
    ws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    hip.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dx, dg, db, ws, act="silu", add_div=ed, seg=T)
    close(dx, xd.grad, 1e-4 if dtype == torch.float32 else 3e-2, "ln(+e) dx")
    close(dg, gd.grad, 1e-4 if dtype == torch.float32 else 2e-2, "ln(+e) dgamma")
    close(db, bd.grad, 1e-4 if dtype == torch.float32 else 2e-2, "ln(+e) dbeta")


@pytest.mark.parametrize("opt", ["adam", "rmsprop"])
def test_optimizer_self_counting_mode(hip, golden_dir, opt):
    """ticket mode: the kernel uses *step_dev + 1 and its last-exiting block publishes it (graph-replay form
    without a separate counter launch); must follow the same torch.optim trajectory"""
    import os
    g = np.load(os.path.join(golden_dir, "optim_traj.npz"))
    n = 257
    p = R.det_fill((n,), 3, 0.5).float().to(DEV)
    s1, s2 = torch.zeros_like(p), torch.zeros_like(p)
    ctr = torch.zeros(1, dtype=torch.int32, device=DEV)
    tick = torch.zeros(hip.optim_ticket_words(), dtype=torch.int32, device=DEV)
    for s in range(4):
        grad = R.det_fill((n,), 20 + s, 0.3 * (s + 1)).float().to(DEV)
        hip.optim_step(opt, p, grad, s1, s2, lr=1e-2, step=0, step_dev=ctr, ticket=tick)
        close(p, torch.from_numpy(g[opt][s]), 2e-5, f"{opt} self-counting step {s + 1}")
        assert int(ctr.cpu()) == s + 1 and int(tick.abs().sum().cpu()) == 0
    # many blocks: the ticket logic must hold with a multi-block grid
    big = torch.zeros(300_000, device=DEV)
    gb = torch.ones_like(big)
    sb = torch.zeros_like(big)
    ctr.zero_()
    for s in range(3):
        hip.optim_step("adagrad", big, gb, sb, None, lr=1.0, step=0, step_dev=ctr, ticket=tick)
    assert int(ctr.cpu()) == 3 and int(tick.abs().sum().cpu()) == 0
    # grids below / at / just above the number of sub-counters (blocks of 256 float4): every block count publishes once
    for nblk in (1, 2, 31, 32, 33, 65):
        nn = nblk * 1024 - 4
        bb, g2, s2 = torch.zeros(nn, device=DEV), torch.ones(nn, device=DEV), torch.zeros(nn, device=DEV)
        ctr.zero_()
        for s in range(2):
            hip.optim_step("adagrad", bb, g2, s2, None, lr=1.0, step=0, step_dev=ctr, ticket=tick)
        assert int(ctr.cpu()) == 2 and int(tick.abs().sum().cpu()) == 0, nblk
    with pytest.raises(hip.HipError):
        hip.optim_step("adagrad", big, gb, sb, None, lr=1.0, step=0, step_dev=ctr,
                       ticket=torch.zeros(1, dtype=torch.int32, device=DEV))
    close(big[:5], -torch.tensor([1.0 + 2 ** -0.5 + 3 ** -0.5] * 5), 1e-5, "adagrad 3 steps")


def test_segment_colsum_bf16_side_output_and_deferred_ln_reduce(hip):
    B, T, N = 4, 9, 64
    x = rnd((B * T, N), 1, 1.0, torch.bfloat16).to(DEV)
    out = torch.zeros(B, N, dtype=torch.float32, device=DEV)
    lp = torch.zeros(B, 2 * N, dtype=torch.bfloat16, device=DEV)
    hip.segment_colsum(x, out, seg=T, mode=0, out_bf16=lp[:, N:])
    assert torch.equal(lp[:, N:].cpu(), out.cpu().to(torch.bfloat16)) and float(lp[:, :N].abs().max()) == 0.0
    # deferred LayerNorm parameter-gradient reduction == immediate one
    M = 40
    xx, dy = rnd((M, N), 2, 1.0, torch.float32).to(DEV), rnd((M, N), 3, 1.0, torch.float32).to(DEV)
    gm, bt = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    y = torch.empty_like(xx)
    mu, rs = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.layernorm_fwd(xx, gm, bt, y, mu, rs)
    ws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    dx1, dx2 = torch.empty_like(xx), torch.empty_like(xx)
    dg1, db1, dg2, db2 = (torch.empty(N, device=DEV) for _ in range(4))
    hip.layernorm_bwd(dy, xx, gm, mu, rs, dx1, dg1, db1, ws)
    hip.layernorm_bwd(dy, xx, gm, mu, rs, dx2, None, None, ws)
    hip.layernorm_bwd_reduce(ws, dg2, db2, M, N)
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)


@pytest.mark.parametrize("M,N,K", [(3200, 512, 2048), (3200, 512, 512), (100, 128, 256), (37, 64, 64), (640, 1024, 512)])
def test_linear_ln_fwd_matches_unfused(hip, M, N, K):
    """K-split GEMM + fused (bias + residual + LayerNorm) reduction == linear_fwd + layernorm_fwd within bf16 rounding
    (the fused form keeps the GEMM output in fp32 up to the LayerNorm, so it is compared with a float64 restatement)"""
    bf = torch.bfloat16
    x = rnd((M, K), 1, 1.0, bf).to(DEV)
    w = rnd((N, K), 2, K ** -0.5, bf).to(DEV)
    b = rnd((N,), 3, 0.1).to(DEV)
    res = rnd((M, N), 4, 1.0, bf).to(DEV)
    g = (1.0 + rnd((N,), 5, 0.2)).to(torch.float32).to(DEV)
    be = rnd((N,), 6, 0.1).to(DEV)
    y = torch.zeros(M, N, dtype=bf, device=DEV)
    ws = torch.zeros(int(hip.lib().ib_linear_ln_fwd_workspace(M, N, K)), dtype=torch.uint8, device=DEV)
    assert hip.linear_ln_fwd(x, w, b, res, g, be, y, ws)
    torch.cuda.synchronize()
    z = x.double().cpu() @ w.double().cpu().T + b.double().cpu() + res.double().cpu()
    mu = z.mean(-1, keepdim=True)
    exp = (z - mu) / torch.sqrt(((z - mu) ** 2).mean(-1, keepdim=True) + 1e-5) * g.double().cpu() + be.double().cpu()
    close(y, exp, 2e-2, "linear_ln_fwd")
    # shapes outside the fused kernel's domain are reported, not mis-computed
    y2 = torch.zeros(M, 96, dtype=bf, device=DEV)
    ws2 = torch.zeros(int(hip.lib().ib_linear_ln_fwd_workspace(M, 96, K)), dtype=torch.uint8, device=DEV)
    assert not hip.linear_ln_fwd(x, rnd((96, K), 7, 0.1, bf).to(DEV), None, None, torch.ones(96, device=DEV),
                                 torch.zeros(96, device=DEV), y2, ws2)


@pytest.mark.parametrize("M,opts", [(3200, "br"), (200, "br"), (3201, "br"), (1, "br"), (5000, "br"), (8192, "b"), (777, "r"),
                                    (300, "")])
def test_linear_ln_panel_fwd_matches_float64(hip, M, opts):
    """csrc/linln_panel.hip: y = LayerNorm(res + x W^T + bias) for a [512, 512] weight from its packed image, one launch over
    panels of ceil(M / 256) rows (ragged last panel, one and two MFMA row tiles per panel, bias / residual optional, pitched
    input / residual / output rows) against a float64 restatement; 8193 rows and other widths are reported as unsupported"""
    bf = torch.bfloat16
    d, ffn = 512, 1024
    x_p = rnd((M, d + 8), 1, 1.0, bf).to(DEV)                  # row pitch 520 elements
    x = x_p[:, :d]
    w = rnd((d, d), 2, d ** -0.5, bf).to(DEV)
    b = rnd((d,), 3, 0.1).to(DEV) if "b" in opts else None
    res_p = rnd((M, d + 4), 4, 1.0, bf).to(DEV)
    res = res_p[:, :d] if "r" in opts else None
    g = (1.0 + rnd((d,), 5, 0.2)).to(torch.float32).to(DEV)
    be = rnd((d,), 6, 0.1).to(DEV)
    y_p = torch.full((M, d + 16), 7.0, dtype=bf, device=DEV)
    y = y_p[:, :d]
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=DEV)
    hip.ffn_chain_pack([(rnd((ffn, d), 8, 0.05, bf).to(DEV), rnd((d, ffn), 9, 0.05, bf).to(DEV), packed, w)])
    nc = ffn // 512
    wo_img = packed[4 * nc * d * d:(4 * nc + 1) * d * d]
    assert hip.linear_ln_panel_ok(M, d, d)
    assert hip.linear_ln_panel_fwd(x, wo_img, b, res, g, be, y)
    assert hip.lib().ib_debug_last_path() == 15
    torch.cuda.synchronize()
    # the oracle's functions (oracle/ref_cpu.py: nn.Linear, nn.LayerNorm of TransformerBaseline.py:12-13,29-31) in float64
    z = R.linear(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu())
    if res is not None:
        z = z + res.double().cpu()
    exp = R.layer_norm(z, g.double().cpu(), be.double().cpu())
    close(y, exp, 2e-2, "linear_ln_panel_fwd")
    assert torch.all(y_p[:, d:] == 7.0)                          # nothing written beside the rows
    # the same numbers as the split-K form it replaces in the sampler, up to one bf16 rounding of the output
    if M <= 4096:
        y2 = torch.zeros(M, d, dtype=bf, device=DEV)
        ws = torch.zeros(int(hip.lib().ib_linear_ln_fwd_workspace(M, d, d)), dtype=torch.uint8, device=DEV)
        assert hip.linear_ln_fwd(x, w, b, res, g, be, y2, ws)
        assert (y.float() - y2.float()).abs().max().item() <= 2 ** -6 * max(1.0, exp.abs().max().item())
    assert not hip.linear_ln_panel_ok(8193, d, d) and not hip.linear_ln_panel_ok(M, 256, d) and not hip.linear_ln_panel_ok(M, d, 1024)


@pytest.mark.parametrize("M", [3200, 200, 1, 17, 800, 2000, 8192])
def test_linear_panel_fwd_matches_float64(hip, M):
    """csrc/linln_panel.hip::linear_panel_kernel: the frozen-weight in-projection [M, 512] -> [M, 1536] over (row panel,
    512-column chunk) workgroups from the layer's packed image (128-column blocks at a few hundred rows; one / two / four
    MFMA row tiles per panel), pitched input rows, against float64; other shapes are reported as unsupported"""
    bf = torch.bfloat16
    d, ffn = 512, 1024
    x = rnd((M, d + 8), 1, 1.0, bf).to(DEV)[:, :d]
    wq = rnd((3 * d, d), 2, d ** -0.5, bf).to(DEV)
    b = rnd((3 * d,), 3, 0.1).to(DEV)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=DEV)
    hip.ffn_chain_pack([(rnd((ffn, d), 8, 0.05, bf).to(DEV), rnd((d, ffn), 9, 0.05, bf).to(DEV), packed,
                         rnd((d, d), 10, 0.05, bf).to(DEV), wq)])
    nc = ffn // 512
    img = packed[(4 * nc + 2) * d * d:(4 * nc + 5) * d * d]
    y = torch.full((M, 3 * d + 8), 7.0, dtype=bf, device=DEV)
    assert hip.linear_panel_ok(M, 3 * d, d)
    assert hip.linear_panel_fwd(x, img, b, y[:, :3 * d])
    assert hip.lib().ib_debug_last_path() == 17
    torch.cuda.synchronize()
    exp = R.linear(x.double().cpu(), wq.double().cpu(), b.double().cpu())          # oracle/ref_cpu.py
    close(y[:, :3 * d], exp, 1e-2, "linear_panel_fwd")
    assert torch.all(y[:, 3 * d:] == 7.0)
    assert not hip.linear_panel_ok(8193, 3 * d, d) and not hip.linear_panel_ok(M, 3 * d, 256) and not hip.linear_panel_ok(M, 300, d)


@pytest.mark.parametrize("M,ffn", [(3200, 2048), (200, 2048), (400, 2048), (4096, 2048), (3201, 1024), (1, 2048), (50, 512), (100, 1024), (777, 4096), (6400, 2048)])
def test_ffn_infer_fwd_matches_float64_and_is_bitwise_repeatable(hip, M, ffn):
    """csrc/linln_panel.hip::ffn_coop_kernel: the frozen-weight feed-forward sublayer -- a panel of rows shared by the
    workgroups of its hidden chunks, fp32 partial products finished by the slab-reduction LayerNorm launch.  Against a
    float64 restatement with the hidden activation rounded to bf16 (what the kernel keeps in LDS and what the per-op path
    stores); repeated launches give bitwise the same rows (6400 rows: more than one round of workgroups)."""
    bf = torch.bfloat16
    d = 512
    x1 = rnd((M, d), 1, 1.0, bf).to(DEV)
    w1 = rnd((ffn, d), 2, d ** -0.5, bf).to(DEV)
    w2 = rnd((d, ffn), 3, ffn ** -0.5, bf).to(DEV)
    b1, b2 = rnd((ffn,), 4, 0.1).to(DEV), rnd((d,), 5, 0.1).to(DEV)
    g = (1.0 + rnd((d,), 6, 0.2)).to(torch.float32).to(DEV)
    be = rnd((d,), 7, 0.1).to(DEV)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=DEV)
    hip.ffn_chain_pack([(w1, w2, packed)])
    panels = hip.ffn_infer_panels(M, d, ffn)
    assert panels > 0
    ws = torch.empty(int(hip.lib().ib_ffn_infer_workspace(M, d, ffn)), dtype=torch.uint8, device=DEV)
    y = torch.zeros(M, d, dtype=bf, device=DEV)
    hip.ffn_infer_fwd(x1, packed, b1, b2, g, be, y, ws)
    assert hip.lib().ib_debug_last_path() == 16
    torch.cuda.synchronize()
    # oracle/ref_cpu.py's linear / act / layer_norm (TransformerBaseline.py:15-19,33-36), the hidden activation rounded to bf16
    h = R.act("relu", R.linear(x1.double().cpu(), w1.double().cpu(), b1.double().cpu())).to(bf).double()
    exp = R.layer_norm(R.linear(h, w2.double().cpu(), b2.double().cpu()) + x1.double().cpu(), g.double().cpu(), be.double().cpu())
    close(y, exp, 2e-2, "ffn_infer_fwd")
    first = y.clone()
    for _ in range(20):
        y.zero_()
        hip.ffn_infer_fwd(x1, packed, b1, b2, g, be, y, ws)
    torch.cuda.synchronize()
    assert torch.equal(y, first)
    assert hip.ffn_infer_panels(32769, d, ffn) == 0 and hip.ffn_infer_panels(M, 256, ffn) == 0


@pytest.mark.parametrize("M,N,K", [(256, 512, 1470), (256, 300, 512), (32, 512, 512), (1, 300, 512), (100, 72, 200),
                                   (1000, 512, 512), (64, 16, 64)])
@pytest.mark.parametrize("act", ["none", "sigmoid", "elu"])
def test_small_m_forward_and_dgrad_tiles(hip, M, N, K, act):
    """batches of a few hundred rows take the 64 x 16 tiles whose four waves split the reduction (ragged K tail, ragged
    column tile, rows that are only 4-byte aligned); the results must agree with float64 like the 128 x 128 kernels'"""
    bf = torch.bfloat16
    x = rnd((M, K), 1, 1.0, bf).to(DEV)
    w = rnd((N, K), 2, K ** -0.5, bf).to(DEV)
    b = rnd((N,), 3, 0.1).to(DEV)
    y = torch.full((M, N), 7.0, dtype=bf, device=DEV)
    hip.linear_fwd(x, w, b, y, act=act)
    close(y, R.act(act, x.double() @ w.double().T + b.double()), TIGHT[bf], "small-M forward")
    hip.linear_fwd(x, w, None, y)
    close(y, x.double() @ w.double().T, TIGHT[bf], "small-M forward, no bias")
    # dgrad: dx[M, Kout] = (dz[M, Nred] w[Nred, Kout]) * act'(aux); output columns % 16 == 0 for the small tiles
    Kout = (K + 15) // 16 * 16
    dz = rnd((M, N), 4, 1.0, bf).to(DEV)
    w2 = rnd((N, Kout), 5, N ** -0.5, bf).to(DEV)
    aux = rnd((M, Kout), 6, 1.0, bf).to(DEV)
    if act == "sigmoid":
        aux = torch.sigmoid(aux.float()).to(bf)
    dx = torch.full((M, Kout), 7.0, dtype=bf, device=DEV)
    hip.linear_dgrad(dz, w2, dx, act_below=act, aux=aux if act != "none" else None)
    a = aux.double()
    fac = {"none": torch.ones_like(a), "sigmoid": a * (1 - a), "elu": torch.where(a > 0, torch.ones_like(a), a + 1)}[act]
    close(dx, (dz.double() @ w2.double()) * fac, TIGHT[bf], "small-M dgrad")


@pytest.mark.parametrize("M,N,K", [(256, 512, 1470), (64, 300, 512), (1000, 72, 200), (1, 512, 512), (300, 30, 256)])
def test_linear_wgrad_bias_one_launch(hip, M, N, K):
    """short reductions: dW = dz^T x and db = column sums of dz from one launch (ragged 64 x 64 output tiles, more than one
    256-row LDS chunk, accumulate, strided destination)"""
    bf = torch.bfloat16
    dz = rnd((M, N), 1, 1.0, bf).to(DEV)
    x = rnd((M, K), 2, 1.0, bf).to(DEV)
    wide = torch.full((N, K + 6), 7.0, dtype=torch.float32, device=DEV)
    dw = wide[:, :K]                                        # row pitch K + 6
    db = torch.full((N,), 3.0, dtype=torch.float32, device=DEV)
    assert hip.linear_wgrad_bias(dz, x, dw, db)
    ew, eb = dz.double().T @ x.double(), dz.double().sum(0)
    close(dw, ew, 2e-5, "dW")
    close(db, eb, 2e-5, "db")
    assert float((wide[:, K:] - 7.0).abs().max()) == 0.0
    assert hip.linear_wgrad_bias(dz, x, dw, db, accumulate=True)
    close(dw, 2 * ew, 2e-5, "dW accumulate")
    close(db, 2 * eb, 2e-5, "db accumulate")
    # long reductions are refused (the caller falls back), never mis-computed
    big = torch.zeros(2048, N, dtype=bf, device=DEV)
    assert not hip.linear_wgrad_bias(big, torch.zeros(2048, K, dtype=bf, device=DEV), dw, db)
    # fp32: taken up to 256 rows (csrc/gemm_f32_small.hip, exact-f32 MFMA), refused beyond
    if M <= 256:
        assert hip.linear_wgrad_bias(dz.float(), x.float(), dw, db)
        close(dw, ew, 2e-5, "dW fp32")
        close(db, eb, 2e-5, "db fp32")
        assert float((wide[:, K:] - 7.0).abs().max()) == 0.0
    else:
        assert not hip.linear_wgrad_bias(dz.float(), x.float(), dw, db)


def test_fp32_small_batch_kernels_seeded_shape_sweep(hip):
    """the reference's batch sizes in fp32 (csrc/gemm_f32_small.hip: 16 x 16 tiles, the four waves split the reduction, exact-f32
    MFMA): 30 seeded shapes -- ragged rows / columns / reduction tails, every activation, pre-activation output, residual
    addend, strided destinations -- against float64 at the north_star tolerance (1e-3; observed ~1e-6)"""
    import random
    rng = random.Random(4321)
    acts = ["none", "relu", "tanh", "sigmoid", "silu", "elu"]
    for case in range(30):
        M = rng.choice([1, 4, 5, 16, 17, 64, 100, 256])
        N = rng.choice([16, 30, 48, 300, 512])
        K = rng.choice([64, 66, 200, 512, 1470])
        act = acts[case % 6]
        x = rnd((M, K), 10 + case, 1.0, torch.float32).to(DEV)
        w = rnd((N, K), 50 + case, K ** -0.5, torch.float32).to(DEV)
        b = rnd((N,), 90 + case, 0.3, torch.float32).to(DEV)
        wide = torch.full((M, N + 3), 5.0, device=DEV)
        y, z = wide[:, :N], torch.zeros(M, N, device=DEV)
        hip.linear_fwd(x, w, b, y, act=act, z=z if act == "silu" else None)
        pre = x.double() @ w.double().T + b.double()
        close(y, R.act(act, pre), 1e-5, f"fp32 small fwd {M}x{N}x{K} {act}")
        if act == "silu":
            close(z, pre, 1e-5, "pre-activation")
        assert float((wide[:, N:] - 5.0).abs().max()) == 0.0
        # dgrad: dx = (dz w) * act'(aux) + addend ; aux = layer output (pre-activation for silu)
        dz = rnd((M, N), 130 + case, 1.0, torch.float32).to(DEV)
        aux = rnd((M, K), 170 + case, 0.8, torch.float32).to(DEV)
        add = rnd((M, K), 210 + case, 0.5, torch.float32).to(DEV)
        dx = torch.zeros(M, K, device=DEV)
        use_add = case % 2 == 0
        hip.linear_dgrad(dz, w, dx, act_below=act, aux=aux if act != "none" else None, addend=add if use_add else None)
        a = aux.double()
        sg = torch.sigmoid(a)
        fac = {"none": torch.ones_like(a), "relu": (a > 0).double(), "tanh": 1 - a * a, "sigmoid": a * (1 - a),
               "silu": sg * (1 + a * (1 - sg)), "elu": torch.where(a > 0, torch.ones_like(a), a + 1)}[act]
        exp = (dz.double() @ w.double()) * fac + (add.double() if use_add else 0)
        close(dx, exp, 1e-5, f"fp32 small dgrad {M}x{N}x{K} {act}")


def test_small_m_kernels_seeded_shape_sweep(hip):
    """40 seeded random shapes through the small-M forward / dgrad / weight+bias-gradient kernels (ragged rows, ragged
    column tiles, ragged reduction tails, 4-byte-aligned row pitches) against float64"""
    import random
    bf = torch.bfloat16
    rng = random.Random(1234)
    for case in range(40):
        M = rng.choice([1, 3, 17, 64, 65, 200, 256, 511, 700])
        N = rng.choice([16, 30, 48, 128, 300, 512, 600])
        K = rng.choice([64, 66, 96, 200, 300, 512, 1024, 1470])
        x = rnd((M, K), 10 + case, 1.0, bf).to(DEV)
        w = rnd((N, K), 60 + case, K ** -0.5, bf).to(DEV)
        b = rnd((N,), 110 + case, 0.1).to(DEV)
        y = torch.full((M, N), 7.0, dtype=bf, device=DEV)
        hip.linear_fwd(x, w, b, y, act="relu")
        close(y, torch.relu(x.double() @ w.double().T + b.double()), TIGHT[bf], f"fwd {M}x{N}x{K}")
        # weight + bias gradient over the M rows
        dz = rnd((M, N), 160 + case, 1.0, bf).to(DEV)
        dw = torch.full((N, K), 7.0, dtype=torch.float32, device=DEV)
        db = torch.full((N,), 7.0, dtype=torch.float32, device=DEV)
        if hip.linear_wgrad_bias(dz, x, dw, db):
            close(dw, dz.double().T @ x.double(), 2e-5, f"dW {M}x{N}x{K}")
            close(db, dz.double().sum(0), 2e-5, f"db {M}x{N}x{K}")
        # dgrad back to K columns (rounded up to the 16-column tiles the small kernel wants; reduction over N)
        Kc = (K + 15) // 16 * 16
        w2 = rnd((N, Kc), 210 + case, N ** -0.5, bf).to(DEV)
        dx = torch.full((M, Kc), 7.0, dtype=bf, device=DEV)
        hip.linear_dgrad(dz, w2, dx)
        close(dx, dz.double() @ w2.double(), TIGHT[bf], f"dgrad {M}x{N}x{Kc}")


@pytest.mark.parametrize("M,N,K", [(50, 512, 30), (50, 30, 512), (512, 30, 50), (1, 1, 1), (7, 3, 64), (3, 5, 63), (200, 30, 1000)])
def test_tiny_matmul_strided_views(hip, M, N, K):
    """C (+)= A . B with A, B any 2-D views (transposes, unaligned column slices of a wider matrix), mixed fp32 / bf16 storage:
    the frame-embedding projection of the transformer denoiser and its two gradients"""
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    wide = torch.randn(N, 300 + K, generator=g).to(torch.bfloat16).cuda()        # B = a column slice, transposed
    Bv = wide[:, 300:].t()                                                        # [K, N], strides (1, 300 + K)
    A32 = torch.randn(K, M, generator=g).cuda().t()                               # [M, K] transposed view, fp32
    for A in (A32, A32.to(torch.bfloat16)):
        for cdt in (torch.float32, torch.bfloat16):
            big = torch.randn(M, N + 5, generator=g).to(cdt).cuda()
            C = big[:, 2:2 + N]                                                   # rows of a wider matrix
            before = big.clone()
            exp = A.double().cpu() @ Bv.double().cpu()
            hip.tiny_matmul(A, Bv, C)
            tol = 1e-5 if cdt == torch.float32 else 1e-2
            assert (C.double().cpu() - exp).abs().max() <= tol * max(1.0, exp.abs().max())
            hip.tiny_matmul(A, Bv, C, accumulate=True)
            assert (C.double().cpu() - 2 * exp).abs().max() <= 2 * tol * max(1.0, exp.abs().max())
            assert torch.equal(big[:, :2], before[:, :2]) and torch.equal(big[:, 2 + N:], before[:, 2 + N:])
    again = torch.empty(M, N, device="cuda")
    first = hip.tiny_matmul(A32, Bv, torch.empty(M, N, device="cuda")).clone()
    assert torch.equal(hip.tiny_matmul(A32, Bv, again), first)                    # fixed summation order
    with pytest.raises(hip.HipError):
        hip.tiny_matmul(A32, Bv[:, :1].expand(K, N) if N > 1 else Bv, torch.empty(M + 1, N, device="cuda"))


@pytest.mark.parametrize("M", [4096, 12800, 5003])
@pytest.mark.parametrize("use_res", [True, False])
def test_layernorm_fast_path_large_batch(hip, M, use_res, monkeypatch):
    """bf16, N = 512, M >= 4096: the transformer denoiser's LayerNorms take layernorm_{fwd,bwd}512_kernel (every row of a wave
    requested before the first is used; csrc/rowops.hip).  Against float64, and against the generic kernels on the same
    inputs (IB_NO_LN_FAST is read once per process, so the generic result is taken at M just below the threshold on a
    prefix of the same rows)."""
    N, bf = 512, torch.bfloat16
    x = rnd((M, N), 1, 1.5, bf)
    res = rnd((M, N), 2, 1.0, bf) if use_res else None
    gamma = (1 + rnd((N,), 3, 0.1)).float()
    beta = rnd((N,), 4, 0.1).float()
    dy = rnd((M, N), 5, 1.0, bf)
    xd = x.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if use_res else None
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yd = R.layer_norm(xd + (rd if use_res else 0), gd, bd)
    yd.backward(dy.double())
    d = lambda t: None if t is None else t.to(DEV)
    y = torch.empty(M, N, dtype=bf, device=DEV)
    mean = torch.empty(M, dtype=torch.float32, device=DEV)
    rstd = torch.empty(M, dtype=torch.float32, device=DEV)
    hip.layernorm_fwd(d(x), d(gamma), d(beta), y, mean, rstd, res=d(res))
    close(y, yd, 3e-2, "LN fast fwd")
    v = xd.detach() + (rd.detach() if use_res else 0)
    close(mean, v.mean(-1), 1e-4, "mean")
    dx = torch.empty(M, N, dtype=bf, device=DEV)
    dg = torch.zeros(N, device=DEV)
    db = torch.zeros(N, device=DEV)
    ws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    hip.layernorm_bwd(d(dy), d(x), d(gamma), mean, rstd, dx, dg, db, ws, res=d(res))
    close(dx, xd.grad, 3e-2, "LN fast dx")
    close(dg, gd.grad, 2e-2, "LN fast dgamma")
    close(db, bd.grad, 2e-2, "LN fast dbeta")
    # the generic kernels on the first 4000 rows (below the fast path's threshold): the same values row by row up to one bf16 rounding
    m2 = 4000
    y2 = torch.empty(m2, N, dtype=bf, device=DEV)
    mean2 = torch.empty(m2, dtype=torch.float32, device=DEV)
    rstd2 = torch.empty(m2, dtype=torch.float32, device=DEV)
    hip.layernorm_fwd(d(x[:m2]), d(gamma), d(beta), y2, mean2, rstd2, res=d(res[:m2]) if use_res else None)
    assert (y[:m2].float() - y2.float()).abs().max().item() <= 2 ** -7 * y2.float().abs().max().item()     # one bf16 ulp
    dx2 = torch.empty(m2, N, dtype=bf, device=DEV)
    ws2 = torch.empty(hip.layernorm_bwd_workspace_bytes(m2, N), dtype=torch.uint8, device=DEV)
    hip.layernorm_bwd(d(dy[:m2]), d(x[:m2]), d(gamma), mean2, rstd2, dx2, torch.zeros(N, device=DEV), torch.zeros(N, device=DEV),
                      ws2, res=d(res[:m2]) if use_res else None)
    assert (dx[:m2].float() - dx2.float()).abs().max().item() <= 2 ** -7 * dx2.float().abs().max().item()


def test_optimizer_gradient_sources_many_irregular_ranges(hip):
    """ib_optim_step_sources with 40 sources of irregular lengths (4-element ranges, neighbours inside one wave's 256 elements,
    gaps that read g, slab counts 1 .. 11 across the 8-slab batches, column sums over 5 .. 200 partial rows): bitwise equal to
    the plain optimizer fed a gradient reduced in the same fixed orders (slabs in sequence; rows r = g, g + 16, ... per row
    group, groups combined in order).  Guards the wave-uniform source search and the predicated slab batches."""
    g = torch.Generator().manual_seed(11)
    n = 40_000
    p0 = torch.randn(n, generator=g).to(DEV)
    grad = (torch.randn(n, generator=g) * 1e-2).to(DEV)
    ref_grad = grad.clone()
    items, segs, off = [], [], 12
    lens = [4, 8, 4, 256, 252, 1024, 36, 4, 4, 640, 3000, 16, 128, 2048, 4, 100 * 4, 512, 8, 8, 1200]
    part = torch.randn(200, 4096, generator=g).to(DEV)
    col = 0
    for i, ln in enumerate(lens):
        ns = 1 + (i * 3) % 11
        ws = (torch.randn(ns, ln, generator=g) * 1e-2).to(DEV)
        items.append((ws, ns, grad[off:off + ln]))
        acc = torch.zeros(ln, device=DEV)
        for k in range(ns):
            acc = acc + ws[k]
        ref_grad[off:off + ln] = acc
        off += ln + (0 if i % 3 else 8)                    # every third source is followed by a gap that reads g
        # a column-sum range right behind it (its own row count through a 7-tuple segment)
        nc = [4, 30, 64, 130, 7][i % 5]
        rows = [5, 16, 17, 200, 33][i % 5]
        start = (off + 3) // 4 * 4
        segs.append((col, nc, grad[start:start + nc], None, 0.5, part, rows))
        parts16 = []
        for rg in range(16):
            a = torch.zeros(nc, device=DEV)
            for r in range(rg, rows, 16):
                a = a + part[r, col:col + nc]
            parts16.append(a)
        tot = parts16[0]
        for rg in range(1, 16):
            tot = tot + parts16[rg]
        ref_grad[start:start + nc] = tot * 0.5
        pad_end = start + (nc + 3) // 4 * 4               # alignment padding behind a ragged range: never updated by the
        grad[start + nc:pad_end] = 0.0                     # fused kernel; a zero gradient leaves it untouched in the reference too
        ref_grad[start + nc:pad_end] = 0.0
        col += (nc + 3) // 4 * 4
        off = start + (nc + 3) // 4 * 4
    assert off < n and len(items) + len(segs) == 40
    for opt in ("rmsprop", "adam"):
        pa, pb = p0.clone(), p0.clone()
        s1a, s1b = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        s2a, s2b = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for step in (1, 2):
            hip.optim_step(opt, pa, grad, s1a, s2a, 1e-3, step=step, sources=(items, None, 0, segs))
            hip.optim_step(opt, pb, ref_grad, s1b, s2b, 1e-3, step=step)
        assert torch.equal(pa, pb) and torch.equal(s1a, s1b) and torch.equal(s2a, s2b), opt
