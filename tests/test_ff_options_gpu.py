"""FeedForwardBaseline with the reference's optional layers (src/models/FeedForwardRegressionBaseline.py:68-72; CLI flags
--batchnorm / --dropout / --dropout-prob, src/cli/train.py:43-47) on the HIP path: BatchNorm1d (csrc/batchnorm.hip) in
train and eval mode against golden vectors of the REAL reference class, Dropout in eval mode (identity) against them, and
Dropout in train mode against the float64 oracle using the masks recovered from the plan's buffers (torch draws its masks
from the global generator: they cannot be pinned, only their statistics can).  -m gpu, through the C-ABI."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle.fixture_inputs import (FF_OPT_B, FF_OPT_CASES, FF_OPT_HIDDEN, FF_OPT_P, ff_inputs, ff_labels,  # noqa: E402
                                   ff_opt_state)

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def train_args():
    return argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))


def close(a, e, rtol, what="", atol=0.0):
    a = torch.as_tensor(np.asarray(a.detach().cpu().double() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    e = torch.as_tensor(np.asarray(e.detach().cpu().double() if isinstance(e, torch.Tensor) else e), dtype=torch.float64)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    assert torch.isfinite(a).all(), what
    err = (a - e).abs().max().item() if a.numel() else 0.0
    ref = max(e.abs().max().item(), 1e-30) if e.numel() else 1.0
    assert err <= atol + rtol * ref, f"{what}: {err:.3e} > {atol:.1e} + {rtol:.1e} * {ref:.3e}"


def make(bn, dr, dtype=torch.float32, act="relu"):
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    m = FeedForwardBaseline(23, 2, 50, "all_frames", act, 5, 10, hidden_dims=list(FF_OPT_HIDDEN), batchnorm=bn, dropout=dr,
                            dropout_prob=FF_OPT_P, device=DEV, compute_dtype=dtype)
    sd = m.state_dict()
    new = ff_opt_state({k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})
    return m


@pytest.mark.parametrize("name,bn,dr,train", FF_OPT_CASES)
def test_feedforward_options_match_reference_golden(golden_dir, name, bn, dr, train):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = np.load(os.path.join(golden_dir, "ff_options.npz"))
    model = make(bn, dr)
    assert list(model.state_dict().keys()) == list(g[f"keys/bn{int(bn)}_drop{int(dr)}"])
    model.train(train)
    out = model(ff_inputs(FF_OPT_B, 10, 23, 5))
    for k, v in out.items():
        close(v, g[f"{name}/out/{k}"], 1e-3, "out/" + k)
    loss = RegressionLossEvaluator(None, "train", device=DEV)({}, dict(out), ff_labels(FF_OPT_B, 10), [], [], train_args())
    close(loss, g[f"{name}/loss"], 1e-4, "loss")
    loss.backward()
    for k, p in model.named_parameters():
        if p.dim() == 1:
            close(p.grad, g[f"{name}/grad/{k}"], 1e-3, "grad/" + k, atol=1e-6)
        else:
            gn = float(g[f"{name}/gnorm/{k}"])
            close(p.grad.norm(), gn, 1e-3, "gnorm/" + k)
            close(p.grad.reshape(-1)[:64], g[f"{name}/gslice/{k}"], 1e-3, "gslice/" + k, atol=1e-5 * gn)
    for k, v in model.state_dict().items():          # running statistics: updated in train mode only
        if "running" in k:
            close(v, g[f"{name}/after/{k}"], 1e-5, k, atol=1e-7)
        elif "num_batches" in k:
            assert int(v) == int(g[f"{name}/after/{k}"]), k


@pytest.mark.parametrize("bn", [False, True])
@pytest.mark.parametrize("dtype,rt", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_dropout_train_mode_matches_oracle_with_recovered_masks(bn, dtype, rt):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    model = make(bn, True, dtype, act="sigmoid")          # sigmoid: every activation is > 0, so every mask bit is visible
    model.train(True)
    B = 48
    inputs, labels = ff_inputs(B, 10, 23, 5), ff_labels(B, 10)
    out = model(inputs)
    loss = RegressionLossEvaluator(None, "train", device=DEV)({}, dict(out), labels, [], [], train_args())
    loss.backward()
    plan = model._plan
    # the dropout outputs ARE mask * input: divide by the oracle's own input of that layer below
    sd = {k: (v.detach().cpu().double().requires_grad_(True) if v.is_floating_point() else v.cpu())
          for k, v in model.state_dict().items()}
    if dtype == torch.bfloat16:        # the GEMMs read bf16 copies of the matrices
        sd = {k: (v.detach().to(torch.bfloat16).double().requires_grad_(True) if (v.is_floating_point() and v.dim() == 2) else v)
              for k, v in sd.items()}
    lin = [k[:-7] for k in sd if k.endswith(".weight") and sd[k].dim() == 2]
    bns = [k[:-13] for k in sd if k.endswith(".running_mean")]
    layers = [(sd[p + ".weight"], sd[p + ".bias"]) for p in lin]
    # running statistics BEFORE the step (the model has already updated its own): rebuild them from the fixture state
    st0 = ff_opt_state({k: tuple(v.shape) for k, v in model.state_dict().items()})
    bnp = [dict(weight=sd[p + ".weight"], bias=sd[p + ".bias"], running_mean=st0[p + ".running_mean"].to(torch.float32).double(),
                running_var=st0[p + ".running_var"].to(torch.float32).double()) for p in bns] if bn else None
    keep = 1.0 / (1.0 - FF_OPT_P)
    masks = []
    for i in range(3):
        d = [v for k, v in plan.buf._b.items() if k[0] == f"ff.drop{i}"][0].float().cpu().double()
        masks.append(d)
    # mask_i = (dropout output != 0) * keep   (inputs are non-zero: model inputs by construction, sigmoid outputs > 0)
    masks = [(m != 0).double() * keep for m in masks]
    for m in masks:                                        # the draw is Bernoulli(1 - p): 3-sigma band on the keep rate
        n = m.numel()
        rate = float((m != 0).double().mean())
        assert abs(rate - (1 - FF_OPT_P)) <= 3 * (FF_OPT_P * (1 - FF_OPT_P) / n) ** 0.5 + 1e-3, rate
    oin = {k: v.to(dtype).double() for k, v in inputs.items()}
    oexp, stats = R.feedforward_forward_opts(layers, oin, "sigmoid", 10, bn=bnp, training=True, drop_masks=masks)
    lexp, _, _ = R.regression_loss(oexp, {k: v.double() for k, v in ff_labels(B, 10).items()}, range(6), range(6), range(6),
                                   range(12))
    lexp.backward()
    for k, v in out.items():
        close(v, oexp[k], rt, "out/" + k)
    close(loss, lexp, rt, "loss")
    for k, q in model.named_parameters():
        e = sd[k].grad
        if dtype == torch.float32:
            close(q.grad, e, rt, "grad/" + k, atol=rt * 0.05 * float(e.norm()))
        else:
            # bf16: a bias / BatchNorm gradient is a sum over 48 rows of bf16-stored terms of both signs that largely
            # cancel (|sum| ~ 1e-2 from terms ~ 5e-2): element-wise error is set by the TERMS, so it is held in the
            # Frobenius norm of the tensor (<= 4 rt = 16 %), matrices additionally element-wise at 2 rt of their max
            a = q.grad.detach().cpu().double()
            assert float((a - e).norm()) <= 4 * rt * float(e.norm()), ("grad/" + k, float((a - e).norm()), float(e.norm()))
            if e.dim() == 2:
                close(q.grad, e, 2 * rt, "grad/" + k, atol=rt * 0.05 * float(e.norm()))
    # a second forward draws DIFFERENT masks (keyed on the call counter), eval mode applies none
    d0 = [v for k, v in plan.buf._b.items() if k[0] == "ff.drop1"][0].clone()
    model(inputs)
    d1 = [v for k, v in plan.buf._b.items() if k[0] == "ff.drop1"][0]
    assert not torch.equal((d0 != 0), (d1 != 0))
    model.eval()
    a = model(inputs)
    b = model(inputs)
    for k in a:
        assert torch.equal(a[k], b[k])


def test_fused_trainer_with_batchnorm_and_dropout_matches_module_path():
    """HipTrainer (captured graph, device step counter for the masks, running statistics updated by the replayed launches)
    against the drop-in module + torch.optim on the same batches: BatchNorm only -> same trajectory; with Dropout the masks
    differ by construction (step keys), so only the statistics are compared"""
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    B = 32
    inputs, labels = ff_inputs(B, 10, 23, 5), ff_labels(B, 10)
    ref = make(True, False)
    ref.train(True)
    opt = torch.optim.RMSprop(ref.parameters(), lr=1e-3)
    ref_losses = []
    for _ in range(5):
        opt.zero_grad()
        out = ref(inputs)
        loss = RegressionLossEvaluator(None, "train", device=DEV)({}, dict(out), {k: v.clone() for k, v in labels.items()}, [],
                                                                   [], train_args())
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    model = make(True, False)
    model.train(True)
    tr = HipTrainer(model, "regression", "rmsprop", 1e-3, args=train_args())
    got = []
    for _ in range(5):
        tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
        got.append(tr.loss_value())
    assert tr._rec is not None
    for a, e in zip(got, ref_losses):
        assert abs(a - e) <= 1e-4 * abs(e), (got, ref_losses)
    for (k, v), (_, w) in zip(model.state_dict().items(), ref.state_dict().items()):
        close(v, w, 1e-4, k, atol=1e-6)
    # dropout + batchnorm through the captured graph: finite, learning, fresh masks every replay
    model = make(True, True)
    model.train(True)
    tr = HipTrainer(model, "regression", "rmsprop", 1e-3, args=train_args())
    ls, keeps = [], []
    for _ in range(8):
        tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
        ls.append(tr.loss_value())
        keeps.append(([v for k, v in tr.plan.buf._b.items() if k[0] == "ff.drop1"][0] != 0).clone())
    assert all(np.isfinite(ls))
    assert not torch.equal(keeps[-1], keeps[-2]) and not torch.equal(keeps[3], keeps[4])
    # a changed train/eval mode re-captures (the signature carries model.training)
    model.eval()
    tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
    assert np.isfinite(tr.loss_value())
