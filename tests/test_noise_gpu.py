"""The device-side diffusion batch (csrc/noise.hip::ib_diffusion_draw, [BUILD-DEFINED]: the reference has no diffusion
path) against the CPU oracle (oracle/ref_cpu.py: philox4x32 pinned by the Random123 known answers in
tests/test_oracle_golden.py): the 32-bit words and the timestep indices are BIT-EXACT; the normals are Box-Muller over
those words in hardware log2 / sin / cos, held to 2e-5 absolute (+ one bf16 rounding); the x0 gather is bit-exact.  Then
distribution sanity (moments, Kolmogorov-Smirnov), bitwise reproducibility / stream separation, and the contract that
matters: a training step fed by the draw launch IS the step fed the same (x0, t, eps) through the existing path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle.fixture_inputs import det_state  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_philox_words_are_bit_exact():
    from inferbiomechanics_amd import hip
    for seed, step, stream, domain, blocks in ((0, 0, 0, 0, 1), (0x1234_5678_9ABC_DEF0, 7, 3, 1, 1000), (2 ** 64 - 1, 2 ** 31 - 1, 2 ** 32 - 1, 0, 257)):
        out = torch.zeros(4 * blocks, dtype=torch.int32, device=DEV)
        hip.philox_words(out, seed, step, stream, domain)
        got = out.cpu().numpy().view(np.uint32).reshape(blocks, 4)
        assert np.array_equal(got, R.draw_words(blocks, seed, step, stream, domain))
    # the Random123 known answer itself, through the kernel: counter (0,0,0,0), key (0,0)
    out = torch.zeros(4, dtype=torch.int32, device=DEV)
    hip.philox_words(out, 0, 0, 0, 0)
    assert [hex(v) for v in out.cpu().numpy().view(np.uint32)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,D", [(16, 50, 300), (5, 7, 9), (3, 1, 8), (1, 13, 3)])
def test_draw_matches_the_oracle(dtype, B, T, D):
    from inferbiomechanics_amd import hip
    seed, step, stream, S = 0xC0FFEE_0000_0042, 11, 2, 1000
    g = torch.Generator().manual_seed(1)
    N = 3 * B + 1
    per = T * D
    pitch = (per + 7) // 8 * 8
    table = torch.zeros(N, pitch, dtype=dtype)
    table[:, :per] = torch.randn(N, per, generator=g).to(dtype)
    idx = torch.randint(0, N, (B,), generator=g)
    x0 = torch.full((B, T, D), float("nan"), dtype=dtype, device=DEV)
    eps = torch.full((B, T, D), float("nan"), dtype=dtype, device=DEV)
    t = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    sd = torch.tensor([4], dtype=torch.int32, device=DEV)                 # device counter: step = 7 + 4
    hip.diffusion_draw(seed, step=step - 4, step_dev=sd, stream_id=stream, eps=eps, t=t, num_train_steps=S,
                       table=table.to(DEV), idx=idx.to(DEV), x0=x0)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), R.draw_timesteps(B, S, seed, step, stream))                 # bit-exact indices
    assert torch.equal(x0.cpu().reshape(B, per), table[idx][:, :per])                       # bit-exact gather
    want = R.philox_normals(B * per, seed, step, stream).reshape(B, T, D)
    got = eps.cpu().double()
    tol = 2e-5 + (2.0 ** -8) * want.abs() if dtype == torch.bfloat16 else 2e-5
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    # t only / eps only
    t2 = torch.empty_like(t)
    hip.diffusion_draw(seed, step=step, stream_id=stream, t=t2, num_train_steps=S)
    e2 = torch.empty_like(eps)
    hip.diffusion_draw(seed, step=step, stream_id=stream, eps=e2)
    assert torch.equal(t2, t) and torch.equal(e2, eps)


def test_distribution_reproducibility_and_stream_separation():
    from scipy import stats
    from inferbiomechanics_amd import hip
    n = 1 << 21
    e = torch.empty(1, n, dtype=torch.float32, device=DEV)
    hip.diffusion_draw(99, step=3, stream_id=0, eps=e)
    z = e.cpu().numpy().reshape(-1).astype(np.float64)
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.std() - 1) < 3e-3
    assert abs((z ** 3).mean()) < 0.01 and abs((z ** 4).mean() - 3) < 0.03
    assert stats.kstest(z[::16], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5e-3                       # neighbours (cos / sin of one pair) uncorrelated
    t = torch.empty(1 << 18, dtype=torch.int64, device=DEV)
    hip.diffusion_draw(99, step=3, stream_id=0, t=t, num_train_steps=1000)
    c = np.bincount(t.cpu().numpy(), minlength=1000)
    assert c.min() > 0 and stats.chisquare(c).pvalue > 1e-3 and t.max().item() == 999 and t.min().item() == 0
    again = torch.empty_like(e)
    hip.diffusion_draw(99, step=3, stream_id=0, eps=again)
    assert torch.equal(again, e)                                             # a pure function of (seed, step, stream)
    for kw in (dict(seed=100, step=3, stream_id=0), dict(seed=99, step=4, stream_id=0), dict(seed=99, step=3, stream_id=1)):
        other = torch.empty_like(e)
        hip.diffusion_draw(kw.pop("seed"), eps=other, **kw)
        assert abs(np.corrcoef(z[:1 << 18], other.cpu().numpy().reshape(-1)[:1 << 18])[0, 1]) < 0.01


def load_det(module):
    sd = module.state_dict()
    new = det_state({k: tuple(v.shape) for k, v in sd.items()})
    module.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})


@pytest.mark.parametrize("kind,dtype", [("mlp", torch.bfloat16), ("mlp", torch.float32), ("tr", torch.bfloat16)])
def test_drawn_step_is_the_fed_step(kind, dtype):
    """step_drawn(cache, idx) == step((x0, t, eps)) with the very tensors the draw launch produced: same losses, same
    parameters, bit for bit -- and graph replays draw FRESH noise (the device step counter advances)"""
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticMotionWindows
    from inferbiomechanics_amd.data.WindowCache import DeviceMotionCache
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    T, D, B = 10, 48, 8
    torch.manual_seed(5)

    def fresh():
        if kind == "mlp":
            m = DiffusionMLP(D, [64, 64], temb_dim=32, temb_hidden=64, device=DEV, compute_dtype=dtype)
        else:
            m = DiffusionTransformer(D, T, d_model=64, num_heads=4, dim_feedforward=128, num_layers=2,
                                     temporal_embedding_dim=6, temb_dim=32, temb_hidden=64, device=DEV, compute_dtype=dtype)
        load_det(m)
        return m, HipTrainer(m, "diffusion", "rmsprop", 1e-3)

    cache = DeviceMotionCache(SyntheticMotionWindows(64, T, D, seed=3), DEV, dtype)
    ma, ta = fresh()
    fed, la = [], []
    for idx in list(cache.batches(B)) * 2:                       # 16 steps: eager warm-up, capture, pinned replays
        ta.step_drawn(cache, idx)
        x0, t, eps = ta.drawn_batch()
        fed.append((x0.clone(), t.clone(), eps.clone()))
        la.append(ta.loss_value())
    assert ta._rec is not None
    assert not torch.equal(fed[-1][2], fed[-2][2]) and not torch.equal(fed[-1][1], fed[-2][1])    # fresh draws on replay
    for i, (x0, t, eps) in enumerate(fed):                      # what the launch drew is what the oracle says for step i
        assert torch.equal(t.cpu(), R.draw_timesteps(B, 1000, ta.noise_seed, i, 0))
    mb, tb = fresh()
    lb = []
    for x0, t, eps in fed:
        tb.step((x0, t, eps))
        lb.append(tb.loss_value())
    assert la == lb, (la, lb)
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k
    # caller-supplied windows (DataLoader path): t / eps drawn, x0 staged
    mc, tc = fresh()
    lc = []
    for idx in list(cache.batches(B)) * 2:
        x0 = cache.table[idx][:, :T * D].reshape(B, T, D).float().cpu()
        tc.step_x0(x0)
        lc.append(tc.loss_value())
    assert lc == la


def test_sampler_draws_its_start_state_on_the_device():
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    m = DiffusionMLP(24, [32, 32], temb_dim=16, temb_hidden=32, device=DEV, compute_dtype=torch.float32)
    load_det(m)
    s = DDIMSampler(m, num_sample_steps=10)
    a = s.sample_noise(3, 6, 24, seed=11)
    want = R.philox_normals(3 * 6 * 24, 11, 0, 0x40000000).reshape(3, 6, 24).float().to(DEV)
    b = s.sample(want)
    assert torch.allclose(a, b, atol=1e-3, rtol=1e-3)           # same start state (to 2e-5) -> same sample
    assert torch.equal(a, s.sample_noise(3, 6, 24, seed=11)) and not torch.equal(a, s.sample_noise(3, 6, 24, seed=11, draw=1))
