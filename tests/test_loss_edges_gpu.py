"""The edge inputs of the reference's loss unit tests (test/loss/test_RegressionLossEvaluator.py:7-192), pushed through
the fused kernel (ib_regression_loss via RegressionLossEvaluator.__call__): threshold equality of the CoP mask,
last-frame-only metrics, left+right force sum of the COM metric, the 6-wide wrench metric.  Expected values: the
reference's own numbers where its vectors state them, otherwise the CPU oracle (pinned to those vectors by
tests/test_loss_statics_cpu.py).  -m gpu, through the C-ABI."""
import argparse

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None
K_COP, K_FORCE, K_TORQUE, K_WRENCH = R.K_COP, R.K_FORCE, R.K_TORQUE, R.K_WRENCH


def args_all():
    return argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))


def zeros(B, F):
    return {K_COP: torch.zeros(B, F, 6), K_FORCE: torch.zeros(B, F, 6), K_TORQUE: torch.zeros(B, F, 6),
            K_WRENCH: torch.zeros(B, F, 12)}


def run(outs, labs, a=None):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    a = a or args_all()
    ev = RegressionLossEvaluator(None, "dev", device=DEV)
    o = {k: v.clone().to(DEV).requires_grad_(True) for k, v in outs.items()}
    loss = ev({}, dict(o), {k: v.clone() for k, v in labs.items()}, [], [], a)
    loss.backward()
    lo, parts, met = R.regression_loss({k: v.double().requires_grad_(True) for k, v in outs.items()},
                                       {k: v.double() for k, v in labs.items()}, a.predict_grf_components,
                                       a.predict_cop_components, a.predict_moment_components, a.predict_wrench_components)
    return ev, loss, o, lo, parts, met


def close(a, b, tol=1e-6):
    a, b = torch.as_tensor(a).detach().cpu().double().reshape(-1), torch.as_tensor(b).detach().cpu().double().reshape(-1)
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a, b)


def test_cop_mask_threshold_is_strict_on_the_kernel():
    """label force (6, 8, 0): norm exactly 10.0 -> masked OUT (`>`), (6, 8, 0.05) -> in (RegressionLossEvaluator.py:205-214)"""
    outs, labs = zeros(1, 2), zeros(1, 2)
    labs[K_FORCE][0, 0] = torch.tensor([6., 8., 0., 6., 8., 0.05])
    labs[K_FORCE][0, 1] = torch.tensor([0., 0., 10., 0., 0., 10.0001])
    outs[K_COP][:] = 1.0                     # a unit CoP error everywhere: only unmasked 3-vectors may count
    ev, loss, o, lo, parts, met = run(outs, labs)
    # mean over (B, F) = 2 positions: left never passes (both norms == 10), right passes in both frames
    close(ev.cop_losses[0], [0., 0., 0., 1., 1., 1.])
    close(ev.cop_losses[0], parts["cop"])
    g = o[K_COP].grad.cpu()
    assert torch.all(g[..., :3] == 0) and torch.all(g[..., 3:] != 0)      # no gradient through a masked 3-vector
    close(loss, lo)


def test_metrics_use_the_last_frame_only():
    """reference vectors :113-141 (first-frame error -> 0; last-frame error of one of two windows -> 0.5), embedded in
    the left foot of the 6-wide force key (the right foot's zero chunk halves the mean)"""
    lab3 = torch.tensor([[[1., 2., 3.], [4., 5., 6.]], [[1., 2., 3.], [4., 5., 6.]]])
    for out3, want in ((torch.tensor([[[3., -2., 4.], [4., 5., 6.]], [[3., -2., 4.], [4., 5., 6.]]]), 0.0),
                       (torch.tensor([[[3., -2., 4.], [4., 5., 6.]], [[3., -2., 4.], [4., 5., 7.]]]), 0.5)):
        outs, labs = zeros(2, 2), zeros(2, 2)
        outs[K_FORCE][..., :3], labs[K_FORCE][..., :3] = out3, lab3
        outs[K_TORQUE][..., :3], labs[K_TORQUE][..., :3] = out3, lab3
        ev, loss, o, lo, parts, met = run(outs, labs)
        m = ev.metric_means()
        close(m["force"], want / 2)          # mean over 2 windows x 2 chunks (left, right); the reference's 3-wide case has 1 chunk
        close(m["moment"], want / 2)
        close(m["force"], met["force"]); close(m["moment"], met["moment"]); close(m["com_acc"], met["com_acc"])
        close(ev.force_losses[0], parts["force"]); close(loss, lo)


def test_com_acc_sums_left_and_right_force():
    """reference vector :185-192: left/right swapped between output and label -> the summed force agrees -> 0"""
    outs, labs = zeros(1, 2), zeros(1, 2)
    outs[K_FORCE][0] = torch.tensor([[1., 2., 3., 0., 0., 0.], [0., 0., 0., 1., 2., 3.]])
    labs[K_FORCE][0] = torch.tensor([[0., 0., 0., 1., 2., 3.], [1., 2., 3., 0., 0., 0.]])
    ev, loss, o, lo, parts, met = run(outs, labs)
    m = ev.metric_means()
    close(m["com_acc"], 0.0)
    assert m["force"] > 1.0                                   # while the per-foot metric sees the swap
    close(m["force"], met["force"]); close(loss, lo)


def test_wrench_metric_is_six_wide():
    """reference vectors :143-159 (vec_size 6: zero error -> 0; [1..6] against 0 -> ||[1..6]||), in the left wrench"""
    six = torch.tensor([1., 2., 3., 4., 5., 6.])
    outs, labs = zeros(1, 1), zeros(1, 1)
    outs[K_WRENCH][0, 0, :6] = six
    labs[K_WRENCH][0, 0, :6] = six
    ev, *_ = run(outs, labs)
    close(ev.metric_means()["wrench"], 0.0)
    labs[K_WRENCH][0, 0, :6] = 0.0
    ev, loss, o, lo, parts, met = run(outs, labs)
    m = ev.metric_means()
    close(m["wrench"], float(torch.linalg.vector_norm(six)) / 2)         # two 6-wide chunks (left, right), right is exact
    close(m["wrench"], met["wrench"]); close(m["wrench_moment"], met["wrench_moment"])
    close(ev.wrench_losses[0], parts["wrench"]); close(loss, lo)


def test_squared_diff_vectors_of_the_reference_cases():
    """reference vectors :7-21 (zero loss; +1 offset -> all ones) through every key of the kernel"""
    B, F = 2, 4
    base6 = torch.arange(B * F * 6, dtype=torch.float32).reshape(B, F, 6)
    base12 = torch.arange(B * F * 12, dtype=torch.float32).reshape(B, F, 12)
    outs = {K_COP: base6.clone(), K_FORCE: base6.clone() + 20.0, K_TORQUE: base6.clone(), K_WRENCH: base12.clone()}
    for off in (0.0, 1.0):
        labs = {k: v + off for k, v in outs.items()}
        ev, loss, o, lo, parts, met = run(outs, labs)
        for lst in (ev.force_losses, ev.moment_losses, ev.cop_losses, ev.wrench_losses):
            close(lst[0], torch.full_like(lst[0].cpu(), off))
        close(loss, 30.0 * off)
        close(loss, lo)
        gsum = sum(float(v.grad.abs().sum()) for v in o.values())
        assert (gsum == 0.0) == (off == 0.0)


def test_upstream_gradient_scales_the_kernel_gradients():
    """loss.backward() of (3 * loss): the bridge multiplies the stored gradients by the device scalar 3 with the HIP
    scaling launch (no ATen arithmetic on the backward)"""
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = torch.Generator().manual_seed(3)
    outs = {K_COP: torch.randn(3, 5, 6, generator=g), K_FORCE: torch.randn(3, 5, 6, generator=g),
            K_TORQUE: torch.randn(3, 5, 6, generator=g), K_WRENCH: torch.randn(3, 5, 12, generator=g)}
    labs = {k: 12.0 * torch.randn(v.shape, generator=g) for k, v in outs.items()}
    grads = []
    for scale in (1.0, 3.0):
        ev = RegressionLossEvaluator(None, "dev", device=DEV)
        o = {k: v.clone().to(DEV).requires_grad_(True) for k, v in outs.items()}
        loss = ev({}, dict(o), {k: v.clone() for k, v in labs.items()}, [], [], args_all())
        (loss * scale).backward()
        grads.append({k: v.grad.cpu() for k, v in o.items()})
    for k in outs:
        close(grads[1][k], 3.0 * grads[0][k], 1e-6)


def test_second_backward_over_the_same_graph_is_refused():
    """the loss backward scales its saved gradient buffer in place (one launch, no ATen arithmetic): a second backward over
    the same graph (retain_graph=True) would scale it twice, so it raises instead of returning wrong gradients"""
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = torch.Generator().manual_seed(3)
    outs = {k: torch.randn(2, 3, v.shape[-1], generator=g) for k, v in zeros(2, 3).items()}
    labs = {k: torch.randn(2, 3, v.shape[-1], generator=g) for k, v in zeros(2, 3).items()}
    ev = RegressionLossEvaluator(None, "dev", device=DEV)
    o = {k: v.clone().to(DEV).requires_grad_(True) for k, v in outs.items()}
    loss = ev({}, dict(o), labs, [], [], args_all())
    (loss * 3.0).backward(retain_graph=True)
    first = {k: v.grad.clone() for k, v in o.items()}
    with pytest.raises(RuntimeError, match="already run"):
        loss.backward()
    for k, v in o.items():
        assert torch.equal(v.grad, first[k])
    pred = torch.randn(4, 5, 6, generator=g).to(DEV).requires_grad_(True)
    lo = DiffusionLossEvaluator("dev")(pred, torch.randn(4, 5, 6, generator=g))
    lo.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already run"):
        lo.backward()
