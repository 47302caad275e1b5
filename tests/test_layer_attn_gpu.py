"""The attention INSIDE the panel launches (round 5; csrc/ffn_chain.h, ib_ffn_chain_fwd_attn / ib_ffn_chain_bwd_attn):
with panels of exactly one window the forward launch of a layer also computes the NEXT layer's softmax(Q K^T / 8) V behind
its QKV tail, and the backward launch continues through the layer's own attention backward and in-projection dgrad
(nn.MultiheadAttention's core and autograd: src/models/TransformerBaseline.py:12-13,29).

Kernel level: every stage is held to the oracle (oracle/ref_cpu.py: attention_core, mha_forward, linear, layer_norm and
torch autograd through them) evaluated on the values the launch itself stored for the previous stage, so the tolerances are
a few bf16 ulps -- and end to end against the plain float64 layer at bf16-model tolerance.  Shapes: the headline
(256 windows x 50 frames, 8 x 64, ffn 2048), T = 64 (no padding rows), T = 16 and T = 37 (three quarters of the score
tiles masked), one to four hidden chunks.  Stack level: HipTrainer on the 4-layer denoiser with the fused form against the
separate attention launches (same batches, same weights).  -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from inferbiomechanics_amd._tuning import tuning as TU  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
D, H = 512, 8


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def rb(t):
    return t.to(torch.float32).to(BF).to(torch.float64)


def close(got, want, ulps, what):
    got, want = got.detach().cpu().double(), want.double()
    tol = ulps * 2.0 ** -8 * want.abs().clamp_min(want.abs().max() * 2.0 ** -6)
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - want).abs().max()), float(want.abs().max()))


def rel(a, b):
    return float((a.detach().cpu().double() - b.double()).norm() / b.double().norm())


def make(B, T, ffn, seed, qk_gain=2.0):
    g = torch.Generator().manual_seed(seed)
    M = B * T
    q = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF)
    v = lambda n, sc=0.1, base=0.0: base + sc * torch.randn(n, generator=g)
    return dict(x=q(M, D), attn=q(M, D), wo=q(D, D, sc=D ** -0.5), bo=v(D), g1=v(D, 0.2, 1.0), be1=v(D),
                w1=q(ffn, D, sc=D ** -0.5), b1=v(ffn), w2=q(D, ffn, sc=ffn ** -0.5), b2=v(D), g2=v(D, 0.2, 1.0), be2=v(D),
                # the in-projection of the layer above (forward tail) -- also used as THIS layer's for the backward tail;
                # scaled up so that the softmax is not flat (scores of a few units)
                wq=q(3 * D, D, sc=qk_gain * D ** -0.5), bq=v(3 * D), dy=q(M, D))


@pytest.mark.parametrize("B,T,ffn,qk_gain", [(256, 50, 2048, 2.0), (8, 64, 512, 2.0), (5, 16, 1024, 2.0), (3, 37, 512, 2.0),
                                              (8, 50, 512, 5.0)])
def test_layer_launches_with_the_attention_inside(B, T, ffn, qk_gain):
    """qk_gain 5: scores of +-25 -- a PEAKED softmax (most rows put > 0.9 on one key), where dS = P (dP - D) cancels and a
    D taken from the rounded output instead of the recomputed probabilities was 35 % off in round 2"""
    from inferbiomechanics_amd import hip
    M = B * T
    pr = make(B, T, ffn, seed=1000 * T + B, qk_gain=qk_gain)
    dev = {k: t.to(DEV) for k, t in pr.items()}
    assert hip.ffn_chain_workgroups(M, D, ffn, T) == B
    packed = torch.zeros(hip.ffn_chain_packed_elems(D, ffn), dtype=BF, device=DEV)
    hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed, dev["wo"], dev["wq"])])
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=BF, device=DEV)
    f1, s2, y, s1, x1o, qkv, ao = nan(M, ffn), nan(M, D), nan(M, D), nan(M, D), nan(M, D), nan(M, 3 * D), nan(M, D)
    lse = torch.full((B, H, T), float("nan"), device=DEV)
    mean, rstd, mean1, rstd1 = (torch.zeros(M, device=DEV) for _ in range(4))
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, D, ffn, T), dtype=torch.uint8, device=DEV)
    hip.ffn_chain_fwd(dev["x"], packed, dev["b1"], dev["b2"], dev["g2"], dev["be2"], f1, s2, y, mean, rstd, mask,
                      attn_out=(dev["attn"], dev["bo"], dev["g1"], dev["be1"], s1, x1o, mean1, rstd1),
                      qkv_next=(packed, dev["bq"], qkv), attn_next=(ao, lse, T), panel_T=T)
    torch.cuda.synchronize()
    f64 = {k: t.double() for k, t in pr.items()}
    # ---- forward, stage by stage from the launch's own stored values
    close(s1, rb(f64["x"] + R.linear(f64["attn"], f64["wo"], f64["bo"])), 2, "s1")
    close(x1o, R.layer_norm(s1.cpu().double(), f64["g1"], f64["be1"]), 2, "x1")
    x1k = x1o.cpu().double()
    got_f1 = f1.cpu().double()
    z = R.linear(x1k, f64["w1"], f64["b1"])
    assert not bool(((got_f1 - rb(torch.relu(z))).abs() > 2.0 ** -7 * got_f1.abs() + 1e-4)[z.abs() >= 1e-4].any())
    close(s2, rb(x1k + R.linear(got_f1, f64["w2"], f64["b2"])), 2, "s2")
    close(y, R.layer_norm(s2.cpu().double(), f64["g2"], f64["be2"]), 2, "y")
    yk = y.cpu().double()
    close(qkv, R.linear(yk, f64["wq"], f64["bq"]), 2, "qkv_next")
    # the attention of the launch's own bf16 in-projection (P is rounded to bf16 once, the output once more)
    qk = qkv.cpu().double().view(B, T, 3 * D)
    want_o = R.attention_core(qk, H).reshape(M, D)
    # sum_key P V with P rounded to bf16 (relative 2^-9 each, sum P = 1) and the result rounded once more:
    # |error| <= 2^-9 (max |V| + |o|); twice that allowed
    vmax = float(qk[..., 2 * D:].abs().max())
    err_o = (ao.cpu().double() - want_o).abs()
    assert not bool((err_o > 2.0 ** -8 * (want_o.abs() + vmax)).any()), (float(err_o.max()), vmax)
    assert rel(ao, want_o) < 3e-3
    sc = (qk[..., :D].reshape(B, T, H, 64).transpose(1, 2) @ qk[..., D:2 * D].reshape(B, T, H, 64).transpose(1, 2).transpose(-1, -2)) / 8.0
    assert torch.allclose(lse.cpu().double(), torch.logsumexp(sc, dim=-1), rtol=1e-4, atol=1e-4)
    # ... and of the oracle's attention of the float64 in-projection of the same rows (bf16-model tolerance)
    eye, zero = torch.eye(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
    # (a peaked softmax amplifies the bf16 rounding of q and k -- scores of +-25 move by +-0.1 -- which this comparison,
    # unlike the one above, does not share with the launch)
    assert rel(ao, R.mha_forward(yk.view(B, T, D), f64["wq"], f64["bq"], eye, zero, H).reshape(M, D)) < (1e-2 if qk_gain < 3 else 3e-2)

    # ---- backward: THIS layer's attention = (qkv, lse) just produced (as data), its in-projection = wq (own image)
    nwg = hip.ffn_chain_workgroups(M, D, ffn, T)
    ds2, dz1, ds1, dqkv, dx = nan(M, D), nan(M, ffn), nan(M, D), nan(M, 3 * D), nan(M, D)
    part = torch.full((4 * nwg, D), float("nan"), device=DEV)
    hip.ffn_chain_bwd(dev["dy"], s2, mean, rstd, dev["g2"], packed, mask, ds2, dz1, None, part,
                      attn_out=(s1, mean1, rstd1, dev["g1"], ds1, None), attn_bwd=(qkv, lse, dqkv, dx, T))
    torch.cuda.synchronize()
    # LayerNorm2 backward -> feed-forward dgrads -> LayerNorm1 backward through autograd on the oracle's functions
    def ln_bwd(sin, gam, bet, dout):
        sin = sin.clone().requires_grad_(True)
        R.layer_norm(sin, gam, bet).backward(dout)
        return sin.grad
    want_ds2 = rb(ln_bwd(s2.cpu().double(), f64["g2"], f64["be2"], f64["dy"]))
    close(ds2, want_ds2, 2, "ds2")
    ds2k = ds2.cpu().double()
    close(dz1, rb((ds2k @ f64["w2"]) * (got_f1 > 0)), 2, "dz1")
    dx1 = rb(dz1.cpu().double() @ f64["w1"] + ds2k)
    close(ds1, rb(ln_bwd(s1.cpu().double(), f64["g1"], f64["be1"], dx1)), 5, "ds1")
    ds1k = ds1.cpu().double()
    dattn = rb(ds1k @ f64["wo"])                      # the launch rounds dattn into its LDS image; it is not stored
    q64 = qk.clone().requires_grad_(True)
    R.attention_core(q64, H).backward(dattn.view(B, T, D))
    want_dqkv = q64.grad.reshape(M, 3 * D)
    # probabilities, dS and the three products' operands are bf16 inside the launch: a few ulps of the largest gradient
    err = (dqkv.cpu().double() - want_dqkv).abs()
    sharp = qk_gain > 3
    if sharp:
        pmax = torch.softmax(sc, dim=-1).max(-1).values
        assert float((pmax > 0.9).double().mean()) > 0.3, float((pmax > 0.9).double().mean())     # the case is what it says
    assert float(err.max()) < (6e-2 if sharp else 3e-2) * float(want_dqkv.abs().max()), (float(err.max()), float(want_dqkv.abs().max()))
    for c, nm in enumerate(("dq", "dk", "dv")):
        assert rel(dqkv[:, c * D:(c + 1) * D], want_dqkv[:, c * D:(c + 1) * D]) < (2e-2 if sharp else 8e-3), \
            (nm, rel(dqkv[:, c * D:(c + 1) * D], want_dqkv[:, c * D:(c + 1) * D]))
    close(dx, dqkv.cpu().double() @ f64["wq"] + ds1k, 3, "dx")
    assert bool(torch.isfinite(part).all())
    dgam2 = part[:nwg].sum(0).cpu().double()
    s2k = s2.cpu().double()
    xh2 = (s2k - s2k.mean(-1, keepdim=True)) / torch.sqrt(s2k.var(-1, unbiased=False, keepdim=True) + 1e-5)
    want = (f64["dy"] * xh2).sum(0)
    assert torch.allclose(dgam2, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))


def test_one_window_panels_without_the_attention_tail():
    """a layer with nothing above it (the top of a stack, a stand-alone layer): the same one-window geometry, no tail"""
    from inferbiomechanics_amd import hip
    B, T, ffn = 6, 50, 1024
    M = B * T
    pr = make(B, T, ffn, seed=3)
    dev = {k: t.to(DEV) for k, t in pr.items()}
    packed = torch.zeros(hip.ffn_chain_packed_elems(D, ffn), dtype=BF, device=DEV)
    hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed, dev["wo"], dev["wq"])])
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=BF, device=DEV)
    outs = []
    for pt in (T, 0):
        f1, s2, y, s1, x1o = nan(M, ffn), nan(M, D), nan(M, D), nan(M, D), nan(M, D)
        mean, rstd, mean1, rstd1 = (torch.zeros(M, device=DEV) for _ in range(4))
        mask = torch.zeros(hip.ffn_chain_mask_bytes(M, D, ffn, pt), dtype=torch.uint8, device=DEV)
        hip.ffn_chain_fwd(dev["x"], packed, dev["b1"], dev["b2"], dev["g2"], dev["be2"], f1, s2, y, mean, rstd, mask,
                          attn_out=(dev["attn"], dev["bo"], dev["g1"], dev["be1"], s1, x1o, mean1, rstd1), panel_T=pt)
        outs.append((f1, s2, y, s1, x1o, mean, rstd))
    torch.cuda.synchronize()
    for a, b in zip(*outs):                   # rows never mix: the panel size does not change a single bit
        assert torch.equal(a, b)
    assert hip.ffn_chain_workgroups(M, D, ffn, 8) == 0 and hip.ffn_chain_workgroups(M, D, ffn, 7) == 0      # T < 16; M % T
    with pytest.raises(hip.HipError):
        hip.ffn_chain_fwd(dev["x"], packed, dev["b1"], dev["b2"], dev["g2"], dev["be2"], *outs[0][:3], outs[0][5], outs[0][6],
                          torch.zeros(1 << 20, dtype=torch.uint8, device=DEV), panel_T=8)


@pytest.mark.parametrize("B,T", [(256, 50), (128, 32)])
def test_trainer_fused_attention_against_separate_launches(B, T):
    """the 4-layer denoiser's training step, attention inside the layer launches vs the separate attention kernels: same
    weights, same batches; the two bf16 trajectories differ by rounding only (the separate kernels read the same bf16 qkv),
    and the fused step issues none of the separate attention / in-projection dgrad launches for layers whose neighbour
    carries them"""
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    Dm = 300
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(B, T, Dm, generator=g).to(DEV, BF), torch.randint(0, 1000, (B,), generator=g).to(DEV),
                torch.randn(B, T, Dm, generator=g).to(DEV, BF)) for _ in range(3)]

    def run(no_fuse):
        TU.no_attn_fuse = no_fuse
        try:
            torch.manual_seed(0)
            m = DiffusionTransformer(Dm, T, d_model=512, num_heads=8, dim_feedforward=2048, num_layers=4, device=DEV,
                                     compute_dtype=BF)
            tr = HipTrainer(m, "diffusion", "sgd", 1e-2, use_graph=False)
            with hip.record_launches() as rec:
                tr.step(batches[0])
            names = [n for n, _ in rec.calls]
            losses = [tr.loss_value()]
            for b in batches[1:]:
                tr.step(b)
                losses.append(tr.loss_value())
            return names, losses, {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
        finally:
            TU.no_attn_fuse = False
    n1, l1, p1 = run(False)
    n0, l0, p0 = run(True)
    assert n0.count("ib_attention_fwd") == 4 and n0.count("ib_attention_bwd") == 4
    assert n1.count("ib_attention_fwd") == 1 and n1.count("ib_attention_bwd") == 0       # layer 0's forward core only
    assert n1.count("ib_ffn_chain_fwd_attn") == 4 and n1.count("ib_ffn_chain_bwd_attn") == 4
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 2e-3 * abs(b), (l1, l0)
    for k in p0:
        dn = float((p1[k] - p0[k]).norm())
        assert dn <= 2e-2 * max(float(p0[k].norm()), 1e-6) or dn < 1e-4, (k, dn, float(p0[k].norm()))
