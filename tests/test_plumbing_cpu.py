"""Host-side plumbing of every launch plan, run on CPU tensors with the C-ABI in DRY-RUN mode
(inferbiomechanics_amd.hip._DryRunLib: arguments are marshalled through the real ctypes signatures and
shape/stride checks run, but nothing is launched and nothing is computed).  This catches glue errors
(wrong arity, shape mismatches between plan steps, missing buffers) without a GPU; numerical parity is
the job of the -m gpu tests."""
import argparse

import pytest
import torch


@pytest.fixture()
def dry():
    from inferbiomechanics_amd import hip
    hip.set_dry_run(True)
    yield hip
    hip.set_dry_run(False)


def targs():
    return argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))


def test_feedforward_plan_plumbing(dry):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from oracle.fixture_inputs import ff_inputs, ff_labels
    for dt in (torch.float32, torch.bfloat16):
        for actn in ("sigmoid", "silu"):
            m = FeedForwardBaseline(23, 2, 50, 'all_frames', actn, 5, 10, compute_dtype=dt)
            out = m(ff_inputs(3, 10, 23, 5))
            assert out['groundContactWrenchesInRootFrame'].shape == (3, 10, 12)
            ev = RegressionLossEvaluator(None, 'train', device='cpu')
            loss = ev({}, out, ff_labels(3, 10), [], [], targs())
            loss.backward()
            assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    names = dry.lib().calls
    assert "ib_concat_keys" in names and "ib_linear_wgrad" in names and "ib_regression_loss" in names


def test_transformer_layer_plumbing(dry):
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    for dt in (torch.float32, torch.bfloat16):
        layer = TransformerLayer(64, 4, 128, 0.0, dtype=dt)
        x = torch.randn(2, 9, 64, requires_grad=True)
        y = layer(x)
        y.float().sum().backward()
        assert x.grad.shape == x.shape
        assert all(p.grad is not None for p in layer.parameters())


def test_denoiser_plumbing(dry):
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    for dt in (torch.float32, torch.bfloat16):
        for model in (DiffusionMLP(30, [32, 48], temb_dim=16, temb_hidden=24, compute_dtype=dt),
                      DiffusionTransformer(30, 7, d_model=32, num_heads=4, dim_feedforward=64, num_layers=2,
                                           temporal_embedding_dim=6, temb_dim=16, temb_hidden=24, compute_dtype=dt)):
            x = torch.randn(3, 7, 30)
            t = torch.tensor([0, 5, 999])
            pred = model(x, t)
            assert pred.shape == x.shape
            loss = DiffusionLossEvaluator()(pred, torch.randn(3, 7, 30))
            loss.backward()
            assert all(p.grad is not None and p.grad.shape == p.shape for p in model.parameters()), \
                [k for k, p in model.named_parameters() if p.grad is None]
