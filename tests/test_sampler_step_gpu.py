"""ONE bf16 denoise step of the BENCHED sampler plan (BASELINE configs[4]: 4 layers, d_model = 512, 8 heads, ffn 2048,
T = 200, D = 300) against the oracle's `denoiser_transformer_forward` (oracle/ref_cpu.py; float64) on the bf16-rounded
weights and the bf16-rounded start state -- the direct comparison of the sampler's d = 512 launches (csrc/linln_panel.hip:
linear_panel / linear_ln_panel / ffn_infer with its 128- / 256-column sub-chunks; from 8193 rows the training-shape panel
launches of csrc/ffn_chain.hip in their frozen-weight form) with the oracle, at the bf16 tolerance of the training-mode
test (3e-2 of the largest prediction).  The kernel families that dispatched are read back per launch
(`ib_debug_last_path`) and asserted, so a threshold edit cannot quietly move the comparison to other kernels.  -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
T, D, S = 200, 300, 100


@pytest.fixture(scope="module")
def bench_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    torch.manual_seed(0)                               # bench.py::build_model
    m = DiffusionTransformer(D, T, d_model=512, num_heads=8, dim_feedforward=2048, num_layers=4, device=DEV, compute_dtype=BF)
    # what the bf16 launches read: matrices through the bf16 shadow, vectors (biases, LayerNorm) in fp32
    p = {k: (v.detach().float().to(BF) if v.dim() >= 2 else v.detach().float()).cpu().double() for k, v in m.state_dict().items()}
    return m, p


@pytest.mark.parametrize("B", [1, 2, 16, 256])
def test_one_denoise_step_of_the_benched_plan_matches_the_oracle(bench_model, B):
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    model, p = bench_model
    g = torch.Generator().manual_seed(100 + B)
    xT = torch.randn(B, T, D, generator=g).to(BF)
    smp = DDIMSampler(model, S, use_graph=False)
    with hip.record_launches() as rec:
        x1 = smp.sample(xT.to(DEV), steps=1)
        torch.cuda.synchronize()
    fam = {hip.PATH_NAMES[pth] for (name, _), pth in zip(rec.calls, rec.paths) if pth}
    names = [n for n, _ in rec.calls]
    if B <= 32:
        assert {"linear_ln_panel", "ffn_infer"} <= fam, fam
        assert ("linear_panel" in fam) == (B * T <= 1600), (B, fam)      # the in-projection over row panels up to 1600 rows
    else:
        assert "ffn_chain" in fam and "ib_ffn_chain_fwd_infer" in names, (fam, sorted(set(names)))
        # 800 panels = three full rounds of 256 workgroups + 32: the windows beyond the last full round (245 ... 255) take
        # the per-op row-panel kernels on a side branch, beside the fused launches of windows 0 ... 244
        assert {"linear_ln_panel", "ffn_infer"} <= fam, fam
        assert names.count("ib_ffn_chain_fwd_infer") == 4 and names.count("ib_ffn_infer_fwd") == 4, sorted(set(names))
    eps_hat = smp._bufs["eps"][:, :, :D]              # the prediction the step's DDIM update consumed (pitched buffer)
    # the oracle on a few windows (windows never mix: attention is per window, everything else per row)
    wins = sorted({0, B // 2, B - 1} | ({244, 245} if B == 256 else set()))     # both sides of the main / side split
    tabs = model.tables(torch.device(DEV))
    t0 = int(tabs.ddim_t[0])
    assert t0 == int(R.ddim_timesteps(1000, S)[0])
    with torch.no_grad():
        want = R.denoiser_transformer_forward(p, xT[wins].double(), torch.full((len(wins),), t0, dtype=torch.int64), 4, 8)
    got = eps_hat[wins].cpu().double()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= 3e-2 * scale, (B, err, scale)
    assert float((got - want).norm() / want.norm()) <= 1.5e-2
    # ... and the state after the step = the oracle's DDIM update of the oracle's prediction
    coef = R.ddim_coeffs(1000, S)[0]
    ab = R.alphas_cumprod(R.linear_beta_schedule(1000))
    ts = R.ddim_timesteps(1000, S)
    want_x1 = R.ddim_step(xT[wins].double(), want, ab[ts[0]], ab[ts[1]])
    assert float((x1[wins].cpu().double() - want_x1).abs().max()) <= 3e-2 * float(want_x1.abs().max()), coef
