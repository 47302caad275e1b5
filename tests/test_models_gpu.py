"""Model-level parity on the GPU: the HIP launch plans (through the drop-in nn.Module / loss-evaluator
surface) against (a) the golden vectors generated from the real reference classes and (b) the float64
oracle on seeded inputs.  fp32 mode: <= 1e-3 relative (north_star); bf16 mode: stated per test."""
import argparse
import math
import os

import numpy as np
import pytest
import torch

from inferbiomechanics_amd._tuning import tuning as TU

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle.fixture_inputs import (FF_CASES, LOSS_SUBSETS, TL_CASES, det_state, ff_inputs, ff_labels,  # noqa: E402
                                   loss_case_outputs)

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd import hip
    hip.lib()


def close(actual, expected, rtol, what="", atol=0.0):
    a = torch.as_tensor(np.asarray(actual.detach().cpu().double() if isinstance(actual, torch.Tensor) else actual),
                        dtype=torch.float64)
    e = torch.as_tensor(np.asarray(expected.detach().cpu().double() if isinstance(expected, torch.Tensor) else expected),
                        dtype=torch.float64)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite"
    err = (a - e).abs().max().item() if a.numel() else 0.0
    ref = max(e.abs().max().item(), 1e-30) if e.numel() else 1.0
    if os.environ.get("IB_TEST_REPORT"):            # measurement aid: how much of the tolerance a comparison uses (pytest -s)
        print(f"[tol] {what}: used {err / max(atol + rtol * ref, 1e-300):.3f} of the bound (rtol {rtol:.1e})")
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} > {atol:.1e} + {rtol:.1e} * {ref:.3e}"


def train_args(grf=range(6), cop=range(6), moment=range(6), wrench=range(12)):
    return argparse.Namespace(predict_grf_components=list(grf), predict_cop_components=list(cop),
                              predict_moment_components=list(moment), predict_wrench_components=list(wrench))


def load_det(module):
    sd = module.state_dict()
    new = det_state({k: tuple(v.shape) for k, v in sd.items()})
    module.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})


@pytest.mark.parametrize("name,hist,stride,actn", FF_CASES)
def test_feedforward_matches_reference_golden(golden_dir, name, hist, stride, actn):
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = np.load(os.path.join(golden_dir, f"ff_{name}.npz"))
    F = hist // stride
    model = FeedForwardBaseline(23, 2, hist, "all_frames", actn, stride, 10, hidden_dims=[512, 512], device=DEV)
    load_det(model)
    inputs = ff_inputs(4, F, 23, stride)            # CPU tensors, as the reference DataLoader yields
    labels = ff_labels(4, F)
    out = model(inputs)
    for k, v in out.items():
        close(v, g["out/" + k], 1e-3, "out/" + k)
    ev = RegressionLossEvaluator(dataset=None, split="train", device=DEV)
    loss = ev({}, out, labels, [], [], train_args())
    close(loss, g["loss"], 1e-4, "loss")
    m = ev.metric_means()
    close(torch.tensor([m["force"], m["moment"], m["cop"], m["wrench"], m["wrench_moment"], m["com_acc"]]),
          g["metrics"], 1e-4, "metrics")
    loss.backward()
    for k, p in model.named_parameters():
        gn = float(g["gnorm/" + k])
        close(p.grad.norm(), g["gnorm/" + k], 1e-3, "gnorm/" + k)
        close(p.grad.reshape(-1)[:64], g["gslice/" + k], 1e-3, "gslice/" + k, atol=1e-5 * gn)
    # one optimizer step with the fused flat kernel == torch.optim on the reference (lr 1e-4, train.py:41)
    from inferbiomechanics_amd import hip
    for opt in ("rmsprop", "adam", "sgd"):
        for k, p in model.named_parameters():
            pp = p.detach().clone().reshape(-1)
            gg = p.grad.detach().clone().reshape(-1)
            s1, s2 = torch.zeros_like(pp), torch.zeros_like(pp)
            hip.optim_step(opt, pp, gg, s1, s2, lr=1e-4, step=1)
            close(pp[:64], g[f"step_{opt}/" + k], 0, f"step_{opt}/{k}", atol=3e-6)


def test_feedforward_bf16_close_to_oracle():
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    model = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, device=DEV, compute_dtype=torch.bfloat16)
    load_det(model)
    inputs = ff_inputs(4, 10, 23, 5)
    out = model(inputs)
    sd = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    layers = [(sd[f"net.{2 * i}.weight"], sd[f"net.{2 * i}.bias"]) for i in range(3)]
    exp = R.feedforward_forward(layers, {k: v.double() for k, v in inputs.items()}, "sigmoid", 10)
    for k in exp:
        close(out[k], exp[k], 3e-2, "bf16 " + k)     # bf16 storage: 8 significant bits
    torch.cat([v.reshape(4, -1) for v in out.values()], 1).float().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("name,d,h,ffn,B,T,dt", TL_CASES)
def test_transformer_layer_matches_reference_golden(golden_dir, name, d, h, ffn, B, T, dt):
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    g = np.load(os.path.join(golden_dir, f"tl_{name}.npz"))
    layer = TransformerLayer(d, h, ffn, 0.0, dtype=torch.float32, device=DEV)   # float64 reference case runs fp32 here
    load_det(layer)
    x = R.det_fill((B, T, d), 7, 1.0, torch.float32).to(DEV).requires_grad_(True)
    wout = R.det_fill((B, T, d), 8, 1.0, torch.float32).to(DEV)
    y = layer(x)
    close(y[:, ::7, ::5], g["y_sub"], 1e-3, "y_sub")
    close(y.sum(), g["y_sum"], 1e-3, "y_sum", atol=1e-2)
    close((y * y).sum(), g["y_sq"], 1e-3, "y_sq")
    (y * wout).sum().backward()
    close(x.grad[:, ::7, ::5], g["dx_sub"], 1e-3, "dx_sub")
    close(x.grad.norm(), g["dx_norm"], 1e-3, "dx_norm")
    if "y_full" in g.files:
        close(y, g["y_full"], 1e-3, "y_full")
        close(x.grad, g["dx_full"], 1e-3, "dx_full")
    for k, p in layer.named_parameters():
        gn = float(g["gnorm/" + k])
        close(p.grad.norm(), g["gnorm/" + k], 1e-3, "gnorm/" + k)
        close(p.grad.reshape(-1)[:64], g["gslice/" + k], 1e-3, "gslice/" + k, atol=1e-4 * gn)


@pytest.mark.parametrize("subset", list(LOSS_SUBSETS))
def test_loss_evaluator_matches_reference_golden(golden_dir, subset):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    outs = {k: v.to(DEV).requires_grad_(True) for k, v in loss_case_outputs().items()}   # separate contiguous tensors
    labels = ff_labels(5, 7)
    ev = RegressionLossEvaluator(dataset=None, split="dev", device=DEV)
    loss = ev({}, dict(outs), labels, [], [], train_args(*LOSS_SUBSETS[subset]))
    close(loss, g[f"{subset}/loss"], 1e-5, "loss")
    close(ev.force_losses[0], g[f"{subset}/force"], 1e-5, "force")
    close(ev.moment_losses[0], g[f"{subset}/moment"], 1e-5, "moment")
    close(ev.wrench_losses[0], g[f"{subset}/wrench"], 1e-5, "wrench")
    close(ev.cop_losses[0], g[f"{subset}/cop"], 1e-5, "cop")
    m = ev.metric_means()
    close(torch.tensor([m["force"], m["moment"], m["cop"], m["wrench"], m["wrench_moment"], m["com_acc"]]),
          g[f"{subset}/metrics"], 1e-5, "metrics")
    loss.backward()
    for k, v in outs.items():
        close(v.grad, g[f"{subset}/grad/{k}"], 1e-5, "grad/" + k, atol=1e-9)
    rep = ev.build_report(train_args(*LOSS_SUBSETS[subset]), ev.force_losses[0].cpu(), ev.cop_losses[0].cpu(),
                          ev.moment_losses[0].cpu(), ev.wrench_losses[0].cpu(), loss.detach().cpu(), 1, 2, 3, 4, 5, None)
    assert "dev/loss" in rep and all(k.startswith("dev/") for k in rep)
    ev.print_report()
    assert ev.losses == [] and ev.wrench_moment_reported_metrics == []


def _oracle_params(model):
    return {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}


@pytest.mark.parametrize("dtype,rt", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_diffusion_mlp_matches_oracle(dtype, rt):
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    B, T, D, hidden = 6, 50, 300, [512, 512]
    model = DiffusionMLP(D, hidden, device=DEV, compute_dtype=dtype)
    load_det(model)
    x = R.det_fill((B, T, D), 5, 1.0, torch.float32)
    eps = R.det_fill((B, T, D), 6, 1.0, torch.float32)
    t = torch.tensor([0, 3, 250, 500, 998, 999])
    pred = model(x, t)
    loss = DiffusionLossEvaluator()(pred, eps)
    loss.backward()
    p = _oracle_params(model)
    xe = x.to(dtype).double()
    ee = eps.to(dtype).double()
    pe = R.denoiser_mlp_forward(p, xe, t, hidden)
    le = R.eps_mse(pe, ee)
    le.backward()
    close(pred, pe, rt, "eps_hat")
    close(loss, le, rt, "loss")
    for k, q in model.named_parameters():
        gn = float(p[k].grad.norm())
        close(q.grad, p[k].grad, rt * (1 if dtype == torch.float32 else 2), "grad/" + k, atol=rt * 0.05 * gn)


@pytest.mark.parametrize("dtype,rt", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_diffusion_transformer_matches_oracle(dtype, rt):
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    B, T, D = 3, 20, 44
    model = DiffusionTransformer(D, T, d_model=128, num_heads=4, dim_feedforward=256, num_layers=2, device=DEV,
                                 compute_dtype=dtype)
    load_det(model)
    x = R.det_fill((B, T, D), 5, 1.0, torch.float32)
    eps = R.det_fill((B, T, D), 6, 1.0, torch.float32)
    t = torch.tensor([0, 500, 999])
    pred = model(x, t)
    loss = DiffusionLossEvaluator()(pred, eps)
    loss.backward()
    p = _oracle_params(model)
    pe = R.denoiser_transformer_forward(p, x.to(dtype).double(), t, 2, 4)
    le = R.eps_mse(pe, eps.to(dtype).double())
    le.backward()
    close(pred, pe, rt, "eps_hat")
    close(loss, le, rt, "loss")
    for k, q in model.named_parameters():
        gn = float(p[k].grad.norm())
        close(q.grad, p[k].grad, rt * (1 if dtype == torch.float32 else 2), "grad/" + k, atol=rt * 0.05 * gn)


@pytest.mark.parametrize("dtype,rt", [(torch.float32, 2e-3), (torch.bfloat16, 4e-2)])
def test_ddim_sampler_loop_matches_oracle(dtype, rt):
    """the whole sampling loop (device step counter, one captured step replayed) against the oracle's ddim_sample over
    the oracle denoiser; bf16 additionally runs the fused Linear + residual + LayerNorm inference path and must agree
    with the unfused launches"""
    import os
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    B, T, D, S = 3, 24, 44, 10
    model = DiffusionTransformer(D, T, d_model=128, num_heads=2, dim_feedforward=256, num_layers=2, device=DEV,
                                 compute_dtype=dtype)
    load_det(model)
    xT = R.det_fill((B, T, D), 11, 1.0, torch.float32)
    got = DDIMSampler(model, S, use_graph=True).sample(xT.to(DEV))
    p = {k: v.detach() for k, v in _oracle_params(model).items()}
    with torch.no_grad():
        exp = R.ddim_sample(lambda x, t: R.denoiser_transformer_forward(p, x, t, 2, 2), xT.double(), 1000, S)
    close(got, exp, rt, "x_0")
    if dtype == torch.bfloat16:
        TU.no_linear_ln = True
        try:
            ref = DDIMSampler(model, S, use_graph=False).sample(xT.to(DEV))
        finally:
            TU.no_linear_ln = False
        close(got, ref, 3e-2, "fused vs unfused inference path")
    # a second call reuses the captured step and must reproduce the first bit for bit
    smp = DDIMSampler(model, S, use_graph=True)
    a = smp.sample(xT.to(DEV))
    b = smp.sample(xT.to(DEV))
    assert torch.equal(a, b)


def test_backward_after_newer_forward_fails_loudly():
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    model = DiffusionMLP(12, [16, 24], temb_dim=8, temb_hidden=16, device=DEV)
    x = torch.randn(2, 5, 12)
    t = torch.tensor([1, 2])
    a = model(x, t)
    _ = model(x, t)
    with pytest.raises(hip.HipError):
        a.sum().backward()
