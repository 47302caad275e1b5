"""The fused feed-forward sublayer kernels (csrc/ffn_chain.hip: ib_ffn_chain_pack / _fwd / _bwd) against a float64
restatement of the reference arithmetic (src/models/TransformerBaseline.py:15-19,33-36: x2 = LN2(x1 + W2 relu(W1 x1 + b1) + b2)
through oracle/ref_cpu.py's linear / layer_norm) that rounds to bf16 exactly where the kernel stores bf16 (the hidden
activation, the LayerNorm input, dz2, dz1, dx1), so the comparison is at a few bf16 ulps, not at a loose "bf16 model"
tolerance.  Panels: ragged last panel, one panel, the headline shape (12800 rows = 256 panels of 50); hidden widths of one,
two and four chunks.  -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def rb(t):
    """round a float64 tensor to bf16 and back (what a bf16 store + load does)"""
    return t.to(torch.float32).to(BF).to(torch.float64)


def problem(M, ffn, seed):
    g = torch.Generator().manual_seed(seed)
    d = 512
    q = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF)
    return dict(x1=q(M, d), w1=q(ffn, d, sc=d ** -0.5), w2=q(d, ffn, sc=ffn ** -0.5), b1=torch.randn(ffn, generator=g) * 0.1,
                b2=torch.randn(d, generator=g) * 0.1, gamma=1 + 0.2 * torch.randn(d, generator=g),
                beta=0.1 * torch.randn(d, generator=g), dy=q(M, d))


def restate_fwd(pr):
    x1, w1, w2 = pr["x1"].double(), pr["w1"].double(), pr["w2"].double()
    b1, b2 = pr["b1"].double(), pr["b2"].double()
    z = R.linear(x1, w1, b1)
    f1 = rb(torch.relu(z))
    s2 = rb(x1 + R.linear(f1, w2, b2))
    return dict(z=z, f1=f1, s2=s2)


def layer_norm_of(s2, pr):
    """LayerNorm of the bf16 LayerNorm input the KERNEL stored (each stage is held to the previous stage's actual output:
    a one-ulp difference in s2 would otherwise be amplified by the cancellation against the row mean)"""
    gamma, beta = pr["gamma"].double(), pr["beta"].double()
    mean = s2.mean(-1, keepdim=True)
    var = (s2 * s2).mean(-1, keepdim=True) - mean * mean            # the kernel's one-pass form
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)
    y = (s2 - mean) * rstd * gamma + beta
    assert torch.allclose(y, R.layer_norm(s2, gamma, beta), rtol=1e-7, atol=1e-7)
    return y, mean.squeeze(-1), rstd.squeeze(-1)


def restate_bwd(pr, s2, mean, rstd, relu_on):
    """from the kernel's own saved state (s2, mean, rstd, ReLU pattern), rounding where the kernel stores bf16"""
    w1, w2, gamma, dy = pr["w1"].double(), pr["w2"].double(), pr["gamma"].double(), pr["dy"].double()
    xh = (s2 - mean[:, None]) * rstd[:, None]
    dxh = dy * gamma
    dz2 = rb(rstd[:, None] * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True)))
    dz1 = rb((dz2 @ w2) * relu_on)
    return dict(ds2=dz2, dz1=dz1, dx1=dz1 @ w1 + dz2, dgamma=(dy * xh).sum(0), dbeta=dy.sum(0))


def close(got, want, ulps, what):
    got, want = got.detach().cpu().double(), want.double()
    tol = ulps * 2.0 ** -8 * want.abs().clamp_min(want.abs().max() * 2.0 ** -6)
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - want).abs().max()), float(want.abs().max()))


@pytest.mark.parametrize("M,ffn", [(100, 2048), (64, 512), (777, 1024), (3200, 2048), (12800, 2048)])
def test_ffn_chain_matches_the_restatement(M, ffn):
    from inferbiomechanics_amd import hip
    pr = problem(M, ffn, seed=M + ffn)
    ex = restate_fwd(pr)
    d = 512
    dev = {k: v.to(DEV) for k, v in pr.items()}
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=BF, device=DEV)
    hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed)])
    f1 = torch.full((M, ffn), float("nan"), dtype=BF, device=DEV)
    s2, y = torch.full((M, d), float("nan"), dtype=BF, device=DEV), torch.full((M, d), float("nan"), dtype=BF, device=DEV)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device=DEV)
    hip.ffn_chain_fwd(dev["x1"], packed, dev["b1"], dev["b2"], dev["gamma"], dev["beta"], f1, s2, y, mean, rstd, mask)
    torch.cuda.synchronize()
    # the hidden activation: bf16 of fp32-accumulated sums -- one ulp of slack for sums that land on a rounding boundary;
    # elements whose pre-activation is within rounding noise of zero may flip the ReLU
    near0 = ex["z"].abs() < 1e-4
    got_f1 = f1.cpu().double()
    assert not bool(((got_f1 - ex["f1"]).abs() > 2.0 ** -7 * ex["f1"].abs() + 1e-4)[~near0].any())
    # (held to the restatement FROM THE KERNEL'S f1: a one-ulp difference in one large f1 element -- 35 of 800k differ, sums
    # on a rounding boundary -- moves every s2 element of its row by f1 ulp x w2, more than two ulps of a small s2)
    close(s2, rb(pr["x1"].double() + got_f1 @ pr["w2"].double().t() + pr["b2"].double()), 2, "s2")
    assert float((s2.cpu().double() - ex["s2"]).norm() / ex["s2"].norm()) < 1e-3
    s2k = s2.cpu().double()
    ey, emean, erstd = layer_norm_of(s2k, pr)
    close(y, ey, 2, "y")
    assert torch.allclose(mean.cpu().double(), emean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rstd.cpu().double(), erstd, rtol=1e-4)
    # backward, restated from the state the forward kernel saved
    nwg = hip.ffn_chain_workgroups(M, d, ffn)
    ds2, dx1 = torch.full((M, d), float("nan"), dtype=BF, device=DEV), torch.full((M, d), float("nan"), dtype=BF, device=DEV)
    dz1 = torch.full((M, ffn), float("nan"), dtype=BF, device=DEV)
    part = torch.full((2 * nwg, d), float("nan"), device=DEV)
    hip.ffn_chain_bwd(dev["dy"], s2, mean, rstd, dev["gamma"], packed, mask, ds2, dz1, dx1, part)
    torch.cuda.synchronize()
    eb = restate_bwd(pr, s2k, mean.cpu().double(), rstd.cpu().double(), (got_f1 > 0).double())
    close(ds2, eb["ds2"], 2, "ds2")
    close(dz1, rb((ds2.cpu().double() @ pr["w2"].double()) * (got_f1 > 0)), 2, "dz1")        # from the kernel's own dz2
    close(dx1, dz1.cpu().double() @ pr["w1"].double() + ds2.cpu().double(), 2, "dx1")        # from the kernel's own dz1, dz2
    dgam, dbet = part[:nwg].sum(0).cpu().double(), part[nwg:].sum(0).cpu().double()
    assert torch.allclose(dgam, eb["dgamma"], rtol=1e-4, atol=1e-4 * float(eb["dgamma"].abs().max()))
    assert torch.allclose(dbet, eb["dbeta"], rtol=1e-4, atol=1e-4 * float(eb["dbeta"].abs().max()))
    assert bool(torch.isfinite(part).all())
    # end to end against the plain float64 forward / autograd (no rounding emulation): bf16-level agreement
    x = pr["x1"].double().requires_grad_(True)
    yy = R.layer_norm(x + R.linear(torch.relu(R.linear(x, pr["w1"].double(), pr["b1"].double())), pr["w2"].double(),
                                   pr["b2"].double()), pr["gamma"].double(), pr["beta"].double())
    yy.backward(pr["dy"].double())
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel(y, yy.detach()) < 6e-3 and rel(dx1, x.grad) < 1.5e-2, (rel(y, yy.detach()), rel(dx1, x.grad))


@pytest.mark.parametrize("M,ffn,qkv", [(100, 1024, False), (12800, 2048, False), (100, 1024, True), (777, 512, True),
                                       (12800, 2048, True)])
def test_attention_epilogue_form_matches_the_restatement(M, ffn, qkv):
    """attn != NULL: x1 = LN1(x + attn Wo^T + bo) is computed inside the launch (TransformerBaseline.py:29-31) and the backward
    continues through LayerNorm1 and the out-projection's dgrad; every stage held to the kernel's own previous stage.
    qkv: the NEXT layer's in-projection rides behind LayerNorm2 (forward) and its dgrad + residual addend in front of
    LayerNorm2's backward (the launch computes its own dy)"""
    from inferbiomechanics_amd import hip
    d = 512
    pr = problem(M, ffn, seed=7 * M + ffn)
    g = torch.Generator().manual_seed(M)
    q = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(BF)
    x, attn, wo = q(M, d), q(M, d), q(d, d, sc=d ** -0.5)
    bo, gamma1, beta1 = torch.randn(d, generator=g) * 0.1, 1 + 0.2 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    dev = {k: v.to(DEV) for k, v in pr.items()}
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=BF, device=DEV)
    hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed, wo.to(DEV))])
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=BF, device=DEV)
    # the layer above: only its in-projection matters here (its packed image holds just those six blocks)
    wq, bq = q(3 * d, d, sc=d ** -0.5), torch.randn(3 * d, generator=g) * 0.1
    dqkv_n, ds1_n = q(M, 3 * d, sc=0.5), q(M, d)
    packed_n = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=BF, device=DEV)
    if qkv:
        hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed_n, None, wq.to(DEV))])
    qkv_out = nan(M, 3 * d)
    f1, s2, y, s1, x1o = nan(M, ffn), nan(M, d), nan(M, d), nan(M, d), nan(M, d)
    mean, rstd, mean1, rstd1 = (torch.zeros(M, device=DEV) for _ in range(4))
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device=DEV)
    hip.ffn_chain_fwd(x.to(DEV), packed, dev["b1"], dev["b2"], dev["gamma"], dev["beta"], f1, s2, y, mean, rstd, mask,
                      attn_out=(attn.to(DEV), bo.to(DEV), gamma1.to(DEV), beta1.to(DEV), s1, x1o, mean1, rstd1),
                      qkv_next=(packed_n, bq.to(DEV), qkv_out) if qkv else None)
    torch.cuda.synchronize()
    close(s1, rb(x.double() + attn.double() @ wo.double().t() + bo.double()), 2, "s1")
    pr1 = dict(gamma=gamma1, beta=beta1)
    ex1, em1, er1 = layer_norm_of(s1.cpu().double(), pr1)
    close(x1o, ex1, 2, "x1")
    assert torch.allclose(mean1.cpu().double(), em1, rtol=1e-5, atol=1e-5) and torch.allclose(rstd1.cpu().double(), er1, rtol=1e-4)
    # the feed-forward part continues from the kernel's own x1
    pr2 = dict(pr, x1=x1o.cpu())
    ex = restate_fwd(pr2)
    near0 = ex["z"].abs() < 1e-4
    got_f1 = f1.cpu().double()
    assert not bool(((got_f1 - ex["f1"]).abs() > 2.0 ** -7 * ex["f1"].abs() + 1e-4)[~near0].any())
    close(s2, rb(pr2["x1"].double() + got_f1 @ pr["w2"].double().t() + pr["b2"].double()), 2, "s2")
    ey, _, _ = layer_norm_of(s2.cpu().double(), pr)
    close(y, ey, 2, "y")
    if qkv:       # the next layer's in-projection of the kernel's own (bf16) y
        close(qkv_out, y.cpu().double() @ wq.double().t() + bq.double(), 2, "qkv_next")
    # backward
    nwg = hip.ffn_chain_workgroups(M, d, ffn)
    ds2, dz1, ds1, dattn = nan(M, d), nan(M, ffn), nan(M, d), nan(M, d)
    part = torch.full((4 * nwg, d), float("nan"), device=DEV)
    hip.ffn_chain_bwd(None if qkv else dev["dy"], s2, mean, rstd, dev["gamma"], packed, mask, ds2, dz1, None, part,
                      attn_out=(s1, mean1, rstd1, gamma1.to(DEV), ds1, dattn),
                      qkv_head=(packed_n, dqkv_n.to(DEV), ds1_n.to(DEV)) if qkv else None)
    torch.cuda.synchronize()
    if qkv:       # dy = dqkv_next . Wqkv_next + ds1_next, rounded to bf16 where the kernel parks it in LDS
        pr = dict(pr, dy=rb(dqkv_n.double() @ wq.double() + ds1_n.double()))
    eb = restate_bwd(pr, s2.cpu().double(), mean.cpu().double(), rstd.cpu().double(), (got_f1 > 0).double())
    # (with the QKV head the kernel's dy is an fp32 sum rounded to bf16 inside the launch and never stored: a few of its 6.5 M
    # elements round the other way than the float64 restatement's, and LayerNorm's backward moves ds2 by that much)
    close(ds2, eb["ds2"], 5 if qkv else 2, "ds2")
    dx1 = rb(dz1.cpu().double() @ pr["w1"].double() + ds2.cpu().double())          # the kernel rounds dx1 into its LDS image
    s1k, m1k, r1k = s1.cpu().double(), mean1.cpu().double(), rstd1.cpu().double()
    xh1 = (s1k - m1k[:, None]) * r1k[:, None]
    dxh1 = dx1 * gamma1.double()
    want_ds1 = rb(r1k[:, None] * (dxh1 - dxh1.mean(-1, keepdim=True) - xh1 * (dxh1 * xh1).mean(-1, keepdim=True)))
    # (dx1 is re-rounded here from the kernel's bf16 dz1 / ds2 in float64; the kernel rounds its fp32 sum -- two of 6.5 M
    # elements land on the other side of a rounding boundary of dx1 and move ds1 by one more ulp)
    close(ds1, want_ds1, 5, "ds1")
    close(dattn, ds1.cpu().double() @ wo.double(), 2, "dattn")
    sums = [part[k * nwg:(k + 1) * nwg].sum(0).cpu().double() for k in range(4)]
    for got, want in zip(sums, (eb["dgamma"], eb["dbeta"], (dx1 * xh1).sum(0), dx1.sum(0))):
        assert torch.allclose(got, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))
    assert bool(torch.isfinite(part).all())


def test_pack_layout_of_all_four_images():
    """block (nt, kb) of W_eff at (kb * 32 + nt) KiB, lane (n = lane % 16, q = lane / 16) holding k = 8 q .. 8 q + 7"""
    from inferbiomechanics_amd import hip
    ffn, d = 1024, 512
    g = torch.Generator().manual_seed(1)
    w1 = torch.randn(ffn, d, generator=g).to(BF)
    w2 = torch.randn(d, ffn, generator=g).to(BF)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=BF, device=DEV)
    hip.ffn_chain_pack([(w1.to(DEV), w2.to(DEV), packed)])
    pk = packed.cpu()[:4 * 2 * 512 * 512].view(4, 2, 16, 32, 64, 8)  # image, chunk, kb, nt, lane, j (then Wo, Wo^T)
    eff = {0: lambda c: w1[512 * c:512 * c + 512, :], 1: lambda c: w2[:, 512 * c:512 * c + 512],
           2: lambda c: w2[:, 512 * c:512 * c + 512].t(), 3: lambda c: w1[512 * c:512 * c + 512, :].t()}
    for img in range(4):
        for c in range(2):
            W = eff[img](c)                                          # [n, k]
            want = W.reshape(32, 16, 16, 4, 8).permute(2, 0, 3, 1, 4).reshape(16, 32, 64, 8)   # kb, nt, (q, n) -> lane
            assert torch.equal(pk[img, c], want), (img, c)
