"""TransformerLayer(dropout != 0) in train mode on the GPU (reference: src/models/TransformerBaseline.py:12-13 --
nn.MultiheadAttention(dropout=p) drops softmax probabilities --, :30 dropout1, :35 dropout2).

torch's generator cannot be reproduced by a counter-based hash, so parity in train mode is "the same arithmetic given the
masks": the float64 oracle (pinned on the REAL class with the masks torch drew, tests/test_oracle_golden.py::
test_transformer_layer_train_mode_dropout_matches_reference) is evaluated with the masks the HIP kernels used
(ib_attention_drop_mask / ib_dropout on ones, same (seed, step)) and must give the same outputs and gradients.  Both attention
implementations are covered: the fp32 VALU kernels (any head size) and the bf16 MFMA kernels (head size 64).  -m gpu."""
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
    from inferbiomechanics_amd import hip
    hip.lib()


def rel(a, e):
    a, e = a.detach().cpu().double(), e.detach().cpu().double()
    assert a.shape == e.shape and torch.isfinite(a).all()
    return float((a - e).abs().max() / e.abs().max().clamp_min(1e-30))


def fro(a, e):
    a, e = a.detach().cpu().double(), e.detach().cpu().double()
    return float((a - e).norm() / e.norm().clamp_min(1e-30))


def attn_oracle(qkv, H, mult):
    """softmax(q k^T / sqrt(dh)) x mult, then . v -- float64, packed qkv [B,T,3d] as the in-proj produces it"""
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, T, H, dh).transpose(1, 2) for i in range(3))
    p = torch.softmax((q @ k.transpose(-1, -2)) / dh ** 0.5, dim=-1)
    if mult is not None:
        p = p * mult
    return (p @ v).transpose(1, 2).reshape(B, T, d)


CASES = [  # name, dtype, B, T, H, dh, tolerance
    ("valu_f32_dh36", torch.float32, 3, 37, 3, 36, 2e-5),
    ("valu_f32_T200", torch.float32, 2, 200, 2, 16, 2e-5),
    ("valu_bf16_dh32", torch.bfloat16, 2, 50, 4, 32, 2e-2),
    ("mfma_bf16_T50", torch.bfloat16, 3, 50, 8, 64, 2e-2),
    ("mfma_bf16_T200", torch.bfloat16, 2, 200, 2, 64, 2e-2),
    ("mfma_bf16_T256", torch.bfloat16, 1, 256, 2, 64, 2e-2),
    # enough (window, head) pairs to fill the chip with 8-wave workgroups: the two-pass long-window forward kernel
    ("mfma2p_bf16_T200", torch.bfloat16, 16, 200, 8, 64, 2e-2),
    ("mfma2p_bf16_T100", torch.bfloat16, 32, 100, 8, 64, 2e-2),
    ("mfma2p_bf16_T250", torch.bfloat16, 32, 250, 4, 64, 2e-2),
]


@pytest.mark.parametrize("name,dt,B,T,H,dh,tol", CASES, ids=[c[0] for c in CASES])
def test_attention_probability_dropout_matches_oracle(name, dt, B, T, H, dh, tol):
    from inferbiomechanics_amd import hip
    d = H * dh
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B, T, 3 * d, generator=g) * 0.7).to(dt).to(DEV)
    dout = torch.randn(B, T, d, generator=g).to(dt).to(DEV)
    drop = (0.3, 0x51, 7, None)
    out, lse = torch.empty(B, T, d, dtype=dt, device=DEV), torch.empty(B, H, T, dtype=torch.float32, device=DEV)
    with hip.record_launches() as rec:
        hip.attention_fwd(qkv, out, lse, H, drop=drop)
    assert [n for n, _ in rec.calls] == ["ib_attention_fwd_drop"]
    mult = hip.attention_drop_mask(B, T, H, drop, DEV)
    vals = torch.unique(mult).cpu()
    assert len(vals) == 2 and vals[0] == 0 and abs(float(vals[1]) - 1 / 0.7) < 1e-6
    assert abs(float((mult != 0).float().mean()) - 0.7) < 0.02
    q64 = qkv.cpu().double().requires_grad_(True)
    exp = attn_oracle(q64, H, mult.cpu().double())
    assert rel(out, exp) <= tol, rel(out, exp)
    # the log-sum-exp is of the UNdropped scores (the softmax is normalised before its dropout)
    lse_e = torch.logsumexp((lambda q, k: (q @ k.transpose(-1, -2)) / dh ** 0.5)(
        q64[..., :d].reshape(B, T, H, dh).transpose(1, 2), q64[..., d:2 * d].reshape(B, T, H, dh).transpose(1, 2)), -1)
    assert rel(lse, lse_e.detach()) <= (1e-5 if dt == torch.float32 else 2e-2)
    exp.backward(dout.cpu().double())
    dqkv = torch.empty_like(qkv)
    hip.attention_bwd(qkv, out, dout, lse, dqkv, H, drop=drop)
    assert rel(dqkv, q64.grad) <= tol * (1 if dt == torch.float32 else 2), rel(dqkv, q64.grad)
    # not the plain kernels' answer, and p = 0 through the _drop entry is exactly the plain kernel
    plain, plain0 = torch.empty_like(out), torch.empty_like(out)
    hip.attention_fwd(qkv, plain, lse, H)
    hip.attention_fwd(qkv, plain0, lse, H, drop=(0.0, 1, 2, None))
    assert torch.equal(plain, plain0) and rel(out, plain) > 0.05
    assert rel(plain, attn_oracle(qkv.cpu().double(), H, None)) <= tol            # and the plain kernel of this shape


def test_attention_dropout_draws():
    """a step changes the draw; a device-resident step counter gives the draw of the same host step (graph replay reads the
    counter); heads and windows draw independently; p outside [0, 1) is refused"""
    from inferbiomechanics_amd import hip
    B, T, H = 2, 50, 4
    a = hip.attention_drop_mask(B, T, H, (0.2, 9, 3, None), DEV)
    b = hip.attention_drop_mask(B, T, H, (0.2, 9, 4, None), DEV)
    c = hip.attention_drop_mask(B, T, H, (0.2, 9, 0, torch.tensor([3], dtype=torch.int32, device=DEV)), DEV)
    s = hip.attention_drop_mask(B, T, H, (0.2, 10, 3, None), DEV)
    assert torch.equal(a, c) and not torch.equal(a, b) and not torch.equal(a, s)
    flat = (a != 0).reshape(B * H, -1).float()
    for i in range(B * H):
        for j in range(i):
            agree = float((flat[i] == flat[j]).float().mean())
            assert abs(agree - (0.8 * 0.8 + 0.2 * 0.2)) < 0.04, (i, j, agree)       # independent Bernoulli(0.8) draws
    with pytest.raises(hip.HipError):
        hip.attention_drop_mask(B, T, H, (1.0, 9, 3, None), DEV)


def layer_masks(layer, B, T):
    """the three multipliers of the layer's LAST train-mode forward, from the same (seed, step) the plan used"""
    from inferbiomechanics_amd import hip
    plan, p = layer._plan, layer.dropout_p
    step = layer._fwd_calls
    ones = torch.ones(B * T, layer.d, dtype=torch.float32, device=DEV)
    m = {"attn": hip.attention_drop_mask(B, T, layer.h, (p, plan.seed, step, None), DEV)}
    for k, off in (("drop1", 1), ("drop2", 2)):
        m[k] = hip.dropout(ones, torch.empty_like(ones), p, plan.seed + off, step).view(B, T, layer.d)
    return {k: v.cpu().double() for k, v in m.items()}


LAYERS = [  # name, d, heads, ffn, B, T, compute dtype, tolerance (fp32: north_star's 1e-3; bf16: storage rounding)
    ("f32_d108_h3", 108, 3, 60, 3, 37, torch.float32, 1e-3),
    ("f32_d128_h2", 128, 2, 256, 2, 50, torch.float32, 1e-3),
    ("bf16_d128_h2", 128, 2, 256, 4, 50, torch.bfloat16, 4e-2),
    ("bf16_d512_h8", 512, 8, 2048, 2, 50, torch.bfloat16, 4e-2),
]


@pytest.mark.parametrize("name,d,h,ffn,B,T,dt,tol", LAYERS, ids=[c[0] for c in LAYERS])
def test_layer_train_mode_matches_oracle_with_recovered_masks(name, d, h, ffn, B, T, dt, tol):
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    p = 0.2
    layer = TransformerLayer(d, h, ffn, p, dtype=dt, device=DEV)
    sd0 = layer.state_dict()
    layer.load_state_dict({k: v.to(sd0[k].dtype) for k, v in det_state({k: tuple(v.shape) for k, v in sd0.items()}).items()})
    layer.train()
    x = R.det_fill((B, T, d), 7, 1.0, torch.float32).to(dt).to(DEV).requires_grad_(True)
    wout = R.det_fill((B, T, d), 8, 1.0, torch.float32).to(DEV)
    with hip.record_launches() as rec:
        y = layer(x)
    names = [n for n, _ in rec.calls]
    assert names.count("ib_attention_fwd_drop") == 1 and names.count("ib_dropout") == 2, names
    masks = layer_masks(layer, B, T)
    for m in masks.values():
        assert abs(float((m != 0).double().mean()) - (1 - p)) < 0.03
    with hip.record_launches() as rec:
        (y.float() * wout).sum().backward()
    names = [n for n, _ in rec.calls]
    assert names.count("ib_attention_bwd_drop") == 1 and names.count("ib_dropout") == 2, names
    # float64 restatement on the values the kernels saw (bf16 mode: bf16-rounded weights and input)
    sd = {k: v.detach().to(dt).cpu().double().requires_grad_(True) if v.dim() == 2 else
          v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    xe = x.detach().cpu().double().requires_grad_(True)
    ye = R.transformer_layer_forward(sd, xe, h, masks=masks)
    (ye * wout.cpu().double()).sum().backward()
    assert rel(y, ye) <= tol, ("y", rel(y, ye))
    # bf16: single elements of dx sit on LayerNorm-backward cancellations of bf16-stored values (deterministic sinusoid
    # weights give peaked softmaxes and large activations): Frobenius norm, as for the bf16 parameter gradients below
    dx_err = rel(x.grad, xe.grad) if dt == torch.float32 else fro(x.grad, xe.grad)
    assert dx_err <= tol * 2, ("dx", dx_err, rel(x.grad, xe.grad))
    for k, q in layer.named_parameters():
        if dt == torch.float32:
            err = rel(q.grad, sd[k].grad)
            assert err <= tol * 2 or (q.grad.cpu().double() - sd[k].grad).abs().max() <= 1e-4 * sd[k].grad.norm(), (k, err)
        else:
            assert fro(q.grad, sd[k].grad) <= tol * 1.5, (k, fro(q.grad, sd[k].grad))
    # eval-mode arithmetic is a different function, and the eval-mode layer IS the dropout-0 layer
    assert rel(y, R.transformer_layer_forward(sd, xe, h)) > 5 * tol
    layer.eval()
    plain = TransformerLayer(d, h, ffn, 0.0, dtype=dt, device=DEV)
    plain.load_state_dict(layer.state_dict())
    with torch.no_grad():
        assert torch.equal(layer(x.detach()), plain(x.detach()))
    # a second train-mode forward draws other masks
    layer.train()
    with torch.no_grad():
        y2 = layer(x.detach())
    assert not torch.equal(y2, y.detach())
    m2 = layer_masks(layer, B, T)
    assert not torch.equal(m2["attn"], masks["attn"]) and not torch.equal(m2["drop1"], masks["drop1"])


def test_dropout_probability_is_validated():
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    with pytest.raises(ValueError):
        TransformerLayer(64, 1, 64, 1.0, device=DEV)
    with pytest.raises(ValueError):
        TransformerLayer(64, 1, 64, -0.1, device=DEV)


@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("dh,H", [(64, 4), (32, 4)], ids=["mfma", "valu"])
def test_bf16_attention_backward_with_peaked_softmax(p, dh, H):
    """scores of a few hundred (every softmax row is nearly one-hot): dS = P (dP - D) cancels, so D must be the sum the
    products themselves give.  The kernels sum D = sum_key P dP from the recomputed row in fp32 (rowsum(dO x O) of the bf16-
    ROUNDED saved output left dQ / dK 35 % off -- 300 % with dropout -- on such inputs); dQ, dK, dV each within 2 % in the
    Frobenius norm of the float64 restatement on the same bf16 values"""
    from inferbiomechanics_amd import hip
    B, T, d = 2, 50, H * dh
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, T, 3 * d, generator=g)
    qkv[..., :2 * d] *= 6.0                                      # q . k / sqrt(dh) ~ N(0, 36^2)
    qkv = qkv.to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    drop = (p, 3, 1, None) if p else None
    out, lse = torch.empty(B, T, d, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, T, dtype=torch.float32, device=DEV)
    hip.attention_fwd(qkv, out, lse, H, drop=drop)
    dqkv = torch.empty_like(qkv)
    hip.attention_bwd(qkv, out, dout, lse, dqkv, H, drop=drop)
    q64 = qkv.cpu().double().requires_grad_(True)
    mult = hip.attention_drop_mask(B, T, H, drop, DEV).cpu().double() if p else None
    exp = attn_oracle(q64, H, mult)
    exp.backward(dout.cpu().double())
    probs = torch.softmax((q64[..., :d].reshape(B, T, H, dh).transpose(1, 2) @
                           q64[..., d:2 * d].reshape(B, T, H, dh).transpose(1, 2).transpose(-1, -2)) / dh ** 0.5, -1)
    assert float(probs.max(-1).values.median()) > 0.95           # the regime this test is about
    assert rel(out, exp) <= 2e-2
    for i, nm in enumerate("qkv"):
        err = fro(dqkv[..., i * d:(i + 1) * d], q64.grad[..., i * d:(i + 1) * d])
        assert err <= 2e-2, (nm, err)
