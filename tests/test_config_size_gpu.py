"""Parity at the sizes BASELINE.json's configs name (round-1 VERDICT, weak #1): the 4-layer d_model = 512 / 8-head
transformer denoiser (configs[2]-[4]: the first MODEL-level tests that reach attn_{fwd,bwd}_mfma, dh = 64) at T = 50 and
T = 200, forward + every gradient against the float64 CPU oracle; the fused trainer's transformer trajectory against an
oracle run; the benchmarked 100-step DDIM loop at T = 200 against the oracle's ddim_sample; and the bf16 loss curve at
the headline shape (B = 256, T = 50, D = 300) against the fp32 one.  Small batches keep the CPU oracle to seconds.
Tolerances: fp32 <= 1e-3 relative (north_star); bf16 stated per test (8 significant bits of storage)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda"
D, DM, HEADS, FFN, LAYERS = 300, 512, 8, 2048, 4


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.set_num_threads(16)


def rel_err(a, e):
    a, e = a.detach().cpu().double(), e.detach().cpu().double()
    assert a.shape == e.shape and torch.isfinite(a).all()
    return (a - e).abs().max().item() / max(e.abs().max().item(), 1e-30)


def oracle_params(model, grad=True):
    return {k: v.detach().cpu().double().clone().requires_grad_(grad) for k, v in model.state_dict().items()}


def make_transformer(T, dtype, layers=LAYERS, seed=0):
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    torch.manual_seed(seed)
    return DiffusionTransformer(D, T, d_model=DM, num_heads=HEADS, dim_feedforward=FFN, num_layers=layers, device=DEV,
                                compute_dtype=dtype)


@pytest.mark.parametrize("T", [50, 200])
@pytest.mark.parametrize("dtype,rt", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_transformer_denoiser_config_size_matches_oracle(dtype, rt, T):
    """configs[2] (T = 50) and configs[4] (T = 200) model, B = 2: eps_hat, loss and all 58 parameter gradients"""
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    model = make_transformer(T, dtype)
    g = torch.Generator().manual_seed(1)
    x, eps = torch.randn(2, T, D, generator=g), torch.randn(2, T, D, generator=g)
    t = torch.tensor([17, 903])
    pred = model(x, t)
    loss = DiffusionLossEvaluator()(pred, eps)
    loss.backward()
    p = oracle_params(model)
    pe = R.denoiser_transformer_forward(p, x.to(dtype).double(), t, LAYERS, HEADS)
    le = R.eps_mse(pe, eps.to(dtype).double())
    le.backward()
    assert rel_err(pred, pe) <= rt, ("eps_hat", rel_err(pred, pe))
    assert abs(float(loss) - float(le)) <= rt * abs(float(le)), (float(loss), float(le))
    worst = {}
    for k, q in model.named_parameters():
        # bf16: a gradient is a sum over 2 T tokens of products of bf16-stored tensors; compared against the tensor's
        # own max with twice the forward tolerance (fp32: the north_star bound itself).
        # The FFN input layer's gradients are gated by ReLU'(f1): where a pre-activation lies within bf16 rounding of
        # zero the stored mask differs from the float64 one and that (token, unit) contribution flips as a whole --
        # with 2 T tokens per unit a handful of flips moves single rows by ~20 % of the tensor's max (the fp32 run of
        # this very test holds 1e-3 on the same launches).  Those two tensors are held in the Frobenius norm here, and
        # in the max norm below once the oracle is given the kernel's own gate.
        if dtype == torch.bfloat16 and k.endswith(("feedforward.0.weight", "feedforward.0.bias")):
            a, e = q.grad.detach().cpu().double(), p[k].grad
            worst[k] = 0.5 * float((a - e).norm() / e.norm())       # so the same threshold reads "<= 4 rt = 24 %"
        else:
            worst[k] = rel_err(q.grad, p[k].grad)
    bound = rt if dtype == torch.float32 else 2 * rt
    if os.environ.get("IB_TEST_REPORT"):            # measurement aid (pytest -s): the tensors closest to the bound
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        print(f"[tol] T={T} {dtype}: eps_hat {rel_err(pred, pe) / rt:.3f} of rt; gradients (fraction of the bound):",
              [(k, round(v / bound, 3)) for k, v in top])
    bad = {k: v for k, v in worst.items() if v > bound}
    assert not bad, bad
    if dtype == torch.bfloat16:
        # SHOWN, not asserted: hand the float64 oracle the ReLU gate the kernel used (the stored post-ReLU activations
        # f1 of every layer, read back from the plan's buffers).  With the gate fixed the two feedforward.0 gradients
        # meet the same 2 rt max-norm bound as every other tensor -- the residual of the unmasked comparison above is
        # gate flips at near-zero pre-activations, not a gradient-path error.
        plan = model._plan
        masks = []
        for l in range(LAYERS):
            f1 = plan.buf.get(f"tl{l}.f1", (2 * T, FFN), torch.bfloat16)
            masks.append({"relu": (f1.float() > 0).double().cpu().reshape(2, T, FFN)})
        p2 = oracle_params(model)
        pe2 = R.denoiser_transformer_forward(p2, x.to(dtype).double(), t, LAYERS, HEADS, layer_masks=masks)
        R.eps_mse(pe2, eps.to(dtype).double()).backward()
        worst2 = {k: rel_err(q.grad, p2[k].grad) for k, q in model.named_parameters()}
        bad2 = {k: v for k, v in worst2.items() if v > 2 * rt}
        assert not bad2, bad2
        ff0 = {k: round(v, 4) for k, v in worst2.items() if "feedforward.0" in k}
        print("feedforward.0 gradients with the kernel's ReLU gate (max norm):", ff0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 2e-2)])
def test_transformer_trainer_matches_oracle_trajectory(dtype, tol):
    """HipTrainer (flat buffers, grouped weight gradients, reductions folded into the optimizer, hipGraph replay) on the
    d = 512 / 8-head denoiser (2 layers, T = 50, B = 4) against the oracle's RMSprop trajectory on the same batches"""
    from inferbiomechanics_amd.engine import HipTrainer
    T, B, steps, lr, layers = 50, 4, 6, 1e-4, 2
    model = make_transformer(T, dtype, layers=layers, seed=3)
    g = torch.Generator().manual_seed(5)
    bs = [(torch.randn(B, T, D, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randn(B, T, D, generator=g))
          for _ in range(3)]
    p = oracle_params(model)
    st = {k: R.optim_init_state("rmsprop", v.detach()) for k, v in p.items()}
    tabs = R.schedule_tables()
    ref = []
    for i in range(steps):
        x0, t, eps = bs[i % 3]
        x0, eps = x0.to(dtype).double(), eps.to(dtype).double()
        for v in p.values():
            v.grad = None
        loss = R.eps_mse(R.denoiser_transformer_forward(p, R.q_sample(x0, t, eps, tabs), t, layers, HEADS), eps)
        loss.backward()
        ref.append(float(loss))
        with torch.no_grad():
            for k, v in p.items():
                v.copy_(R.optim_step("rmsprop", v, v.grad, st[k], lr, i + 1))
    tr = HipTrainer(model, "diffusion", "rmsprop", lr, use_graph=True)
    got = []
    for i in range(steps):
        x0, t, eps = bs[i % 3]
        tr.step((x0.to(DEV, dtype), t.to(DEV), eps.to(DEV, dtype)))
        got.append(tr.loss_value())
    assert tr._rec is not None
    for a, e in zip(got, ref):
        assert abs(a - e) <= tol * abs(e), (got, ref)
    if dtype == torch.float32:
        # RMSprop's first steps move every weight by ~lr / sqrt(1 - alpha) whatever the gradient's size: compare the
        # UPDATE (p - p0), in units of the largest update
        for k, v in model.state_dict().items():
            e = p[k].detach()
            err = (v.detach().cpu().double() - e).abs().max().item()
            assert err <= 2e-3 * max(e.abs().max().item(), 1e-6) + 3e-5, (k, err)


def test_ddim_100_steps_T200_matches_oracle_fp32():
    """the loop bench.py times (configs[4]: T = 200, 100 DDIM steps, one captured step replayed), B = 1, fp32, against
    the oracle's ddim_sample over the oracle denoiser"""
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    T, S = 200, 100
    model = make_transformer(T, torch.float32, seed=7)
    g = torch.Generator().manual_seed(9)
    xT = torch.randn(1, T, D, generator=g)
    got = DDIMSampler(model, S, use_graph=True).sample(xT.to(DEV))
    p = oracle_params(model, grad=False)
    with torch.no_grad():
        exp = R.ddim_sample(lambda x, t: R.denoiser_transformer_forward(p, x, t, LAYERS, HEADS), xT.double(), 1000, S)
    err = rel_err(got, exp)
    assert err <= 2e-3, err            # 100 chained fp32 denoiser evaluations against float64


def test_ddim_100_steps_T200_bf16_tracks_fp32():
    """the bf16 loop (fused Linear + residual + LayerNorm inference path, per-loop time-embedding table) against the fp32
    loop of the same weights: x_0 within 5 % of its range after 100 chained steps"""
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    T, S = 200, 100
    g = torch.Generator().manual_seed(9)
    xT = torch.randn(2, T, D, generator=g)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        model = make_transformer(T, dt, seed=7)
        outs[dt] = DDIMSampler(model, S, use_graph=True).sample(xT.to(DEV)).float()
    assert rel_err(outs[torch.bfloat16], outs[torch.float32]) <= 5e-2


def _curve(kind, dtype, B, steps, nb=8):
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    T = 50
    torch.manual_seed(11)
    model = DiffusionMLP(D, [512, 512], device=DEV, compute_dtype=dtype) if kind == "mlp" else make_transformer(T, dtype, seed=11)
    g = torch.Generator().manual_seed(13)
    bs = [(torch.randn(B, T, D, generator=g).to(DEV, dtype), torch.randint(0, 1000, (B,), generator=g).to(DEV),
           torch.randn(B, T, D, generator=g).to(DEV, dtype)) for _ in range(nb)]
    tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4)
    c = []
    for i in range(steps):
        tr.step(bs[i % nb])
        c.append(tr.result[0].clone())
    return torch.stack(c).cpu().tolist()


@pytest.mark.parametrize("kind,B,steps,win", [("mlp", 256, 60, 1), ("transformer", 64, 30, 5)])
def test_bf16_loss_curve_at_headline_shape_tracks_fp32(kind, B, steps, win):
    """'matched diffusion loss' at the benchmarked shapes: the bf16 training curve (MLP: the fused chain kernel path) stays
    within 2 % of the fp32 curve (per-op plan, oracle-checked above) on the same batches, and both learn.  MLP: step by
    step.  The 4-layer transformer's first RMSprop steps swing the loss by +-15 % from step to step (1.33, 1.51, 1.20,
    ...), and a bf16 run lands on slightly different points of that swing: its curve is compared as a 5-step moving mean
    at the same 2 % (single steps stay within 4 %)."""
    f32 = _curve(kind, torch.float32, B, steps)
    b16 = _curve(kind, torch.bfloat16, B, steps)
    mov = lambda c: [sum(c[i:i + win]) / win for i in range(len(c) - win + 1)]
    worst = max(abs(a - b) / abs(a) for a, b in zip(mov(f32), mov(b16)))
    single = max(abs(a - b) / abs(a) for a, b in zip(f32, b16))
    assert worst <= 0.02, (worst, single, f32[:5], b16[:5], f32[-5:], b16[-5:])
    assert single <= 0.04, (single, worst)
    assert f32[-1] < f32[0] and b16[-1] < b16[0]


@pytest.mark.parametrize("attn_inside", [True, False])
def test_transformer_large_batch_paths_match_oracle(attn_inside, monkeypatch):
    """attn_inside: the attention core inside the layers' fused launches (round 5: one-window panels, ib_ffn_chain_*_attn)
    or as separate launches with the in-projection dgrad on the NT kernel.
    M = B T = 6400 token rows: the sizes at which the 256 x 128 LDS-DMA kernels take over (forward + dgrad through
    csrc/gemm_nt.hip with transposed weight copies, weight gradients + bias partial sums through csrc/gemm_tn.hip, one
    ib_step_reduce_parts per layer) -- drop-in tier, 2 layers of the d = 512 / 8-head denoiser, bf16, against the float64
    oracle on the same bf16-rounded inputs"""
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd._tuning import tuning as TU
    monkeypatch.setattr(TU, "no_attn_fuse", not attn_inside)
    T, B, layers, rt = 50, 128, 2, 6e-2
    model = make_transformer(T, torch.bfloat16, layers=layers, seed=21)
    g = torch.Generator().manual_seed(22)
    x, eps = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with hip.record_launches() as rec:
        pred = model(x, t)
        loss = DiffusionLossEvaluator()(pred, eps)
        loss.backward()
    names = {n for n, _ in rec.calls}
    assert {"ib_linear_wgrad_slabs_multi_bias", "ib_step_reduce_parts"} <= names, names
    if attn_inside:
        assert {"ib_ffn_chain_fwd_attn", "ib_ffn_chain_bwd_attn"} <= names and "ib_attention_bwd" not in names, names
    else:
        assert {"ib_linear_dgrad_wt", "ib_transpose_multi", "ib_attention_bwd", "ib_ffn_chain_fwd"} <= names, names
    p = oracle_params(model)
    pe = R.denoiser_transformer_forward(p, x.to(torch.bfloat16).double(), t, layers, HEADS)
    le = R.eps_mse(pe, eps.to(torch.bfloat16).double())
    le.backward()
    assert rel_err(pred, pe) <= rt, rel_err(pred, pe)
    assert abs(float(loss) - float(le)) <= 2e-2 * abs(float(le)), (float(loss), float(le))
    bad = {}
    for k, q in model.named_parameters():
        a, e = q.grad.detach().cpu().double(), p[k].grad
        fro = float((a - e).norm() / e.norm())
        if fro > 4e-2:                      # sums over 6400 tokens average the bf16 storage noise: 4 % in the Frobenius norm
            bad[k] = fro
    assert not bad, bad
