"""Pin the CPU oracle (oracle/ref_cpu.py) to golden vectors generated from the real reference
classes by oracle/make_golden.py.  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.fixture_inputs import (FF_CASES, GL_CASES, LOSS_SUBSETS, TL_CASES, det_state, ff_inputs, ff_labels,
                                   gl_inputs, loss_case_outputs)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(a, b, rtol, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    tol = atol + rtol * max(np.abs(b).max(), 1e-30)
    assert err <= tol, f"max err {err} > tol {tol}"


def ff_state(hist, stride, dtype):
    insz, outsz, _ = R.feedforward_sizes(23, 2, hist, stride)
    dims = [insz, 512, 512, outsz]
    shapes = {}
    for i in range(3):
        shapes[f"net.{2 * i}.weight"] = (dims[i + 1], dims[i])
        shapes[f"net.{2 * i}.bias"] = (dims[i + 1],)
    sd = det_state(shapes)
    # the reference holds fp32 parameters: round through fp32 exactly as load_state_dict did
    return {k: v.to(torch.float32).to(dtype) for k, v in sd.items()}


@pytest.mark.parametrize("name,hist,stride,actn", FF_CASES)
def test_feedforward_matches_reference(golden_dir, name, hist, stride, actn):
    g = load(golden_dir, f"ff_{name}.npz")
    dt = torch.float64
    F = hist // stride
    sd = {k: v.requires_grad_(True) for k, v in ff_state(hist, stride, dt).items()}
    layers = [(sd[f"net.{2 * i}.weight"], sd[f"net.{2 * i}.bias"]) for i in range(3)]
    inputs = {k: v.to(dt) for k, v in ff_inputs(4, F, 23, stride).items()}
    labels = {k: v.to(dt) for k, v in ff_labels(4, F).items()}
    out = R.feedforward_forward(layers, inputs, actn, F)
    for k, v in out.items():
        close(v.detach(), g["out/" + k], rtol=1e-4)  # fp32 reference vs f64 oracle
    loss, parts, metrics = R.regression_loss(out, labels, range(6), range(6), range(6), range(12))
    close(loss.detach(), g["loss"], rtol=2e-5)
    close([metrics[k] for k in ("force", "moment", "cop", "wrench", "wrench_moment", "com_acc")],
          g["metrics"], rtol=2e-5)
    loss.backward()
    for k, p in sd.items():
        close(p.grad.norm(), g["gnorm/" + k], rtol=3e-4)
        close(p.grad.reshape(-1)[:64], g["gslice/" + k], rtol=3e-4, atol=1e-6 * float(g["gnorm/" + k]))
    for opt in ("rmsprop", "adam", "sgd"):
        for k, p in sd.items():
            p32 = p.detach().to(torch.float32)
            g32 = p.grad.to(torch.float32)
            st = R.optim_init_state(opt, p32)
            new = R.optim_step(opt, p32, g32, st, 1e-4, 1)
            close(new.reshape(-1)[:64], g[f"step_{opt}/" + k], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name,fmt,F", GL_CASES)
def test_groundlink_matches_reference(golden_dir, name, fmt, F):
    """oracle Groundlink (explicit replicate-padded gather + matmul convolution, ELU, per-frame MLP) against the real
    reference class in eval mode: outputs, loss through the reference evaluator, every parameter gradient"""
    g = load(golden_dir, f"gl_{name}.npz")
    dt = torch.float64
    sd = {k: v.to(torch.float32).to(dt).requires_grad_(True)
          for k, v in det_state(R.groundlink_param_shapes(), 5.0).items()}
    inputs = {k: v.to(dt) for k, v in gl_inputs(3, F).items()}
    Fo = F if fmt == "all_frames" else 1
    labels = {k: v.to(dt) for k, v in ff_labels(3, Fo).items()}
    out = R.groundlink_forward(sd, inputs, fmt)
    for k, v in out.items():
        close(v.detach(), g["out/" + k], rtol=2e-4)
    loss, _, _ = R.regression_loss(out, labels, range(6), range(6), range(6), range(12))
    close(loss.detach(), g["loss"], rtol=5e-5)
    loss.backward()
    for k, p in sd.items():
        close(p.grad.norm(), g["gnorm/" + k], rtol=5e-4)
        close(p.grad.reshape(-1)[:64], g["gslice/" + k], rtol=5e-4, atol=2e-6 * float(g["gnorm/" + k]))


@pytest.mark.parametrize("name,d,h,ffn,B,T,dt", TL_CASES)
def test_transformer_layer_matches_reference(golden_dir, name, d, h, ffn, B, T, dt):
    g = load(golden_dir, f"tl_{name}.npz")
    shapes = dict(zip(R.TL_KEYS, [(3 * d, d), (3 * d,), (d, d), (d,), (ffn, d), (ffn,), (d, ffn), (d,),
                                  (d,), (d,), (d,), (d,)]))
    sd = {k: v.to(dt).to(torch.float64).requires_grad_(True) for k, v in det_state(shapes).items()}
    x = R.det_fill((B, T, d), 7, 1.0, dt).to(torch.float64).requires_grad_(True)
    wout = R.det_fill((B, T, d), 8, 1.0, dt).to(torch.float64)
    y = R.transformer_layer_forward(sd, x, h)
    rt = 1e-9 if dt == torch.float64 else 3e-5
    close(y.detach()[:, ::7, ::5], g["y_sub"], rtol=rt)
    close(y.detach().sum(), g["y_sum"], rtol=rt, atol=rt * 100)
    close((y * y).detach().sum(), g["y_sq"], rtol=rt)
    (y * wout).sum().backward()
    close(x.grad[:, ::7, ::5], g["dx_sub"], rtol=rt * 10)
    close(x.grad.norm(), g["dx_norm"], rtol=rt * 10)
    if "y_full" in g:
        close(y.detach(), g["y_full"], rtol=rt)
        close(x.grad, g["dx_full"], rtol=rt * 10)
    for k, p in sd.items():
        close(p.grad.norm(), g["gnorm/" + k], rtol=rt * 30)
        close(p.grad.reshape(-1)[:64], g["gslice/" + k], rtol=rt * 30, atol=rt * 30 * float(g["gnorm/" + k]))


def test_transformer_layer_train_mode_dropout_matches_reference(golden_dir):
    """TransformerLayer(dropout=0.25).train() of the REAL class with the masks torch drew (recorded by hooks,
    oracle/make_golden.py::gen_transformer_layer_dropout): the restatement applies the three multipliers where the
    reference does -- on the normalised attention probabilities, on the attention block's output, on the feedforward
    output -- so output and every gradient agree to float64 rounding"""
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in load(golden_dir, "tl_dropout_train.npz").items() if k != "meta_torch"}
    p = float(g["p"])
    masks = {k: g["mask/" + k] for k in ("attn", "drop1", "drop2")}
    for m in masks.values():
        vals = torch.unique(m)
        assert len(vals) == 2 and vals[0] == 0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-12
        assert 0.6 < float((m != 0).double().mean()) < 0.9
    sd = {k: g["param/" + k].clone().requires_grad_(True) for k in R.TL_KEYS}
    x = g["x"].clone().requires_grad_(True)
    y = R.transformer_layer_forward(sd, x, 4, masks=masks)
    close(y.detach(), g["y"], rtol=1e-10)
    (y * g["wout"]).sum().backward()
    close(x.grad, g["dx"], rtol=1e-9)
    for k, q in sd.items():
        close(q.grad, g["grad/" + k], rtol=1e-9, atol=1e-11 * float(g["grad/" + k].norm()))
    # and the masks matter: eval-mode arithmetic is a different function
    assert (R.transformer_layer_forward(sd, x, 4).detach() - g["y"]).abs().max() > 1e-2


@pytest.mark.parametrize("subset", list(LOSS_SUBSETS))
def test_loss_evaluator_matches_reference(golden_dir, subset):
    g = load(golden_dir, "loss_cases.npz")
    dt = torch.float64
    outs = {k: v.to(dt).requires_grad_(True) for k, v in loss_case_outputs().items()}
    labels = {k: v.to(dt) for k, v in ff_labels(5, 7).items()}
    grf, cop, mom, wr = LOSS_SUBSETS[subset]
    loss, parts, metrics = R.regression_loss(outs, labels, grf, cop, mom, wr)
    close(loss.detach(), g[f"{subset}/loss"], rtol=1e-5)
    for k in ("force", "moment", "wrench", "cop"):
        close(parts[k].detach(), g[f"{subset}/{k}"], rtol=1e-5)
    close([metrics[k] for k in ("force", "moment", "cop", "wrench", "wrench_moment", "com_acc")],
          g[f"{subset}/metrics"], rtol=1e-5)
    loss.backward()
    for k, v in outs.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        close(gr, g[f"{subset}/grad/{k}"], rtol=1e-5, atol=1e-9)
    close(R.mask_by_threes(labels[R.K_FORCE], 10.0), g["mask"], rtol=0)
    # exact-threshold edge: norm == 10.0 is masked OUT, 10.0000001 is IN
    m = R.mask_by_threes(labels[R.K_FORCE], 10.0)
    assert m[0, 0, :3].sum() == 0 and m[0, 0, 3:].sum() == 3


@pytest.mark.parametrize("opt", ["sgd", "adam", "rmsprop", "adagrad", "adadelta", "adamax"])
def test_optimizers_match_torch_optim(golden_dir, opt):
    g = load(golden_dir, "optim_traj.npz")
    p = R.det_fill((257,), 3, 0.5)
    st = R.optim_init_state(opt, p)
    for s in range(4):
        grad = R.det_fill((257,), 20 + s, 0.3 * (s + 1))
        p = R.optim_step(opt, p, grad, st, 1e-2, s + 1)
        close(p, g[opt][s], rtol=1e-12, atol=1e-14)


# ---- diffusion known answers (no reference code: literature definitions) ----
def test_schedule_known_answers():
    ab = R.alphas_cumprod(R.linear_beta_schedule(1000))
    assert ab.dtype == torch.float64
    assert abs(ab[0].item() - 0.9999) < 1e-15
    assert abs(ab[499].item() - 0.078587242881778235) < 1e-15
    assert abs(ab[999].item() - 4.0358297653756761e-05) < 1e-18
    ts = R.ddim_timesteps(1000, 100)
    assert ts.dtype == torch.int64 and ts[0].item() == 990 and ts[-1].item() == 0 and len(ts) == 100
    assert torch.equal(ts[:-1] - ts[1:], torch.full((99,), 10, dtype=torch.int64))


def test_ddim_roundtrip_and_coeffs():
    # with the true eps, one DDIM step to ab_prev = 1 returns x0 exactly
    tabs = R.schedule_tables()
    x0 = R.det_fill((2, 5, 7), 1)
    eps = R.det_fill((2, 5, 7), 2)
    t = torch.tensor([100, 900])
    xt = R.q_sample(x0, t, eps, tabs)
    for b in range(2):
        back = R.ddim_step(xt[b], eps[b], tabs["alphas_cumprod"][t[b]], 1.0)
        assert torch.allclose(back, x0[b], atol=1e-9)
    co = R.ddim_coeffs(1000, 100)
    ab = tabs["alphas_cumprod"]
    xs = R.ddim_step(xt[0], eps[0], ab[990], ab[980])
    assert torch.allclose(xs, co[0, 0] * xt[0] + co[0, 1] * eps[0], atol=1e-10)


def test_timestep_embedding_known_answers():
    e = R.timestep_embedding(torch.tensor([0, 1, 999]), 128)
    assert e.shape == (3, 128)
    assert torch.all(e[0, :64] == 0) and torch.all(e[0, 64:] == 1)
    assert abs(e[1, 0].item() - math.sin(1.0)) < 1e-15 and abs(e[1, 64].item() - math.cos(1.0)) < 1e-15
    w63 = math.exp(-math.log(10000.0) * 63 / 64)
    assert abs(e[2, 63].item() - math.sin(999 * w63)) < 1e-12


def test_denoisers_run_and_ddim_loop():
    shp = R.denoiser_mlp_param_shapes(12, [16, 24], temb_dim=8, temb_hidden=16)
    p = R.det_params(shp)
    x = R.det_fill((3, 5, 12), 5)
    t = torch.tensor([0, 10, 999])
    y = R.denoiser_mlp_forward(p, x, t, [16, 24], temb_dim=8)
    assert y.shape == x.shape and torch.isfinite(y).all()
    shp = R.denoiser_transformer_param_shapes(12, 5, d_model=16, ffn=32, num_layers=2, pos_dim=3,
                                              temb_dim=8, temb_hidden=16)
    p2 = R.det_params(shp)
    y2 = R.denoiser_transformer_forward(p2, x, t, 2, 4, temb_dim=8)
    assert y2.shape == x.shape and torch.isfinite(y2).all()
    out = R.ddim_sample(lambda xx, tt: R.denoiser_mlp_forward(p, xx, tt, [16, 24], temb_dim=8), x, 1000, 10)
    assert out.shape == x.shape and torch.isfinite(out).all()


# ---- feedforward model with the optional BatchNorm1d / Dropout layers (FeedForwardRegressionBaseline.py:68-72) ----------
def _ff_opt_oracle(case_bn, train, dt=torch.float64, case_drop=False):
    from oracle.fixture_inputs import FF_OPT_B, FF_OPT_HIDDEN, ff_opt_state
    insz, outsz, _ = R.feedforward_sizes(23, 2, 50, 5)
    dims = [insz] + list(FF_OPT_HIDDEN) + [outsz]
    shapes, j, lin, bns = {}, 0, [], []
    for i in range(3):
        if case_drop:
            j += 1
        if case_bn:
            for nm, shp in (("weight", (dims[i],)), ("bias", (dims[i],)), ("running_mean", (dims[i],)),
                            ("running_var", (dims[i],)), ("num_batches_tracked", ())):
                shapes[f"net.{j}.{nm}"] = shp
            bns.append(j)
            j += 1
        shapes[f"net.{j}.weight"] = (dims[i + 1], dims[i])
        shapes[f"net.{j}.bias"] = (dims[i + 1],)
        lin.append(j)
        j += 2
    sd = {k: (v.to(torch.float32).to(dt).requires_grad_(True) if v.dtype != torch.int64 else v)
          for k, v in ff_opt_state(shapes).items()}
    layers = [(sd[f"net.{j}.weight"], sd[f"net.{j}.bias"]) for j in lin]
    bn = [dict(weight=sd[f"net.{j}.weight"], bias=sd[f"net.{j}.bias"], running_mean=sd[f"net.{j}.running_mean"].detach(),
               running_var=sd[f"net.{j}.running_var"].detach()) for j in bns] if case_bn else None
    inputs = {k: v.to(dt) for k, v in ff_inputs(FF_OPT_B, 10, 23, 5).items()}
    labels = {k: v.to(dt) for k, v in ff_labels(FF_OPT_B, 10).items()}
    out, stats = R.feedforward_forward_opts(layers, inputs, "relu", 10, bn=bn, training=train)
    loss, _, _ = R.regression_loss(out, labels, range(6), range(6), range(6), range(12))
    loss.backward()
    return sd, out, loss, stats, bns


@pytest.mark.parametrize("name,bn,train", [("bn_train", True, True), ("bn_eval", True, False), ("drop_eval", False, False)])
def test_feedforward_options_match_reference(golden_dir, name, bn, train):
    """the oracle's BatchNorm1d / optional-layer restatement against the real reference class (train-mode batch
    statistics + running-statistics update, eval-mode running statistics; Dropout is the identity in eval mode)"""
    g = load(golden_dir, "ff_options.npz")
    sd, out, loss, stats, bns = _ff_opt_oracle(bn, train, case_drop=(name == "drop_eval"))
    for k, v in out.items():
        close(v.detach(), g[f"{name}/out/{k}"], rtol=1e-4)
    close(loss.detach(), g[f"{name}/loss"], rtol=3e-5)
    for k, p in sd.items():
        if p.dtype == torch.int64 or "running" in k:
            continue
        if p.dim() == 1:
            close(p.grad, g[f"{name}/grad/{k}"], rtol=3e-4, atol=1e-6)
        else:
            close(p.grad.norm(), g[f"{name}/gnorm/{k}"], rtol=3e-4)
            close(p.grad.reshape(-1)[:64], g[f"{name}/gslice/{k}"], rtol=3e-4, atol=1e-6 * float(g[f"{name}/gnorm/{k}"]))
    if bn:
        for j, st in zip(bns, stats):
            close(st[0], g[f"{name}/after/net.{j}.running_mean"], rtol=1e-5, atol=1e-7)
            close(st[1], g[f"{name}/after/net.{j}.running_var"], rtol=1e-5, atol=1e-7)
            assert int(g[f"{name}/after/net.{j}.num_batches_tracked"]) == (4 if train else 3)


def test_dropout_is_identity_in_eval_mode_of_the_reference(golden_dir):
    g = load(golden_dir, "ff_options.npz")
    n = 0
    for k in g.files:            # parameter indices differ (the Dropout modules shift `net.{j}`): outputs and loss must not
        if k.startswith("bn_drop_eval/out/") or k == "bn_drop_eval/loss":
            np.testing.assert_array_equal(g[k], g["bn_eval/" + k.split("/", 1)[1]])
            n += 1
    assert n == 5


def test_philox_known_answers_and_draw_definitions():
    """[BUILD-DEFINED noise generator] Philox4x32-10 against the Random123 known-answer vectors (kat_vectors:
    `philox4x32 10` rows), then the draw definitions the kernel is held to (tests/test_noise_gpu.py)"""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert [int(v) for v in R.philox4x32(np.array(ctr, dtype=np.uint64), key)] == list(want)
    w = R.draw_words(5, 0x0000_0007_0000_0003, 9, 2, R.DRAW_DOMAIN_T)
    assert np.array_equal(w[3], R.philox4x32(np.array([3, 9, 2, 1], dtype=np.uint64), (3, 7)))
    t = R.draw_timesteps(4096, 1000, 1, 0, 0)
    assert t.dtype == torch.int64 and int(t.min()) >= 0 and int(t.max()) <= 999
    assert torch.equal(t[:5], torch.from_numpy((R.draw_words(5, 1, 0, 0, 1)[:, 0].astype(np.uint64) * 1000 >> 32).astype(np.int64)))
    z = R.philox_normals(1 << 18, 1, 0, 0).numpy()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and np.abs(z).max() <= 5.77
    assert np.array_equal(R.philox_normals(10, 1, 0, 0).numpy(), z[:10])         # a prefix is a prefix


def test_chain_restatement_equals_the_oracle_denoiser():
    """tests/test_chain_gpu.py holds the fused bf16 chain kernel to `chain_restatement` (with bf16 rounding at the kernel's
    storage points).  Here the SAME function, rounding switched off, must be the oracle's MLP denoiser: x_t = q_sample, the
    prediction = denoiser_mlp_forward, the loss = eps_mse, and the gradients w.r.t. the prediction and the LayerNorm
    parameters those of autograd through the oracle -- so the kernel test's expectation IS the pinned oracle arithmetic."""
    from tests.chain_restatement import chain_restatement
    B, T, D, hidden, temb = 5, 7, 12, [16, 16], 8
    shapes = R.denoiser_mlp_param_shapes(D, hidden, temb_dim=temb, temb_hidden=10)
    p = {k: v.requires_grad_(True) for k, v in R.det_params(shapes, seed0=2.0).items()}
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(B, T, D, generator=g, dtype=torch.float64), torch.randn(B, T, D, generator=g, dtype=torch.float64)
    t = torch.tensor([0, 17, 500, 998, 999])
    tabs = R.schedule_tables()
    xt = R.q_sample(x0, t, eps, tabs)
    pred = R.denoiser_mlp_forward(p, xt, t, hidden, temb_dim=temb)
    pred.retain_grad()
    loss = R.eps_mse(pred, eps)
    loss.backward()
    e = R.time_mlp({k: v.detach() for k, v in p.items()}, t, temb, torch.float64)
    W = [p[f"blocks.{i}.linear.weight"].detach() for i in range(2)] + [p["head.weight"].detach()]
    bias = [p[f"blocks.{i}.linear.bias"].detach() for i in range(2)] + [p["head.bias"].detach()]
    gamma = [p[f"blocks.{i}.norm.weight"].detach() for i in range(2)]
    beta = [p[f"blocks.{i}.norm.bias"].detach() for i in range(2)]
    xt2, us, hs, pred2, loss2, g64, b64 = chain_restatement(x0, eps, t, tabs["sqrt_ab"], tabs["sqrt_1mab"], e, W, bias, gamma,
                                                          beta, lambda v: v)
    close(xt2, xt.reshape(B * T, D), 1e-12)
    close(pred2.detach(), pred.detach().reshape(B * T, D), 1e-12)
    close(loss2.detach(), loss.detach(), 1e-12)
    close(pred2.grad, pred.grad.reshape(B * T, D), 1e-12)
    for i in range(2):
        close(g64[i].grad, p[f"blocks.{i}.norm.weight"].grad, 1e-10)
        close(b64[i].grad, p[f"blocks.{i}.norm.bias"].grad, 1e-10)
        # d loss / d u_i summed over the tokens = the block's bias gradient
        close(us[i].grad.sum(0), p[f"blocks.{i}.linear.bias"].grad, 1e-10)
