"""CPU oracle: a torch-eager restatement of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``inferbiomechanics_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed
CPU baseline, never as the product path.

Every function is written as explicit tensor math (no ``torch.nn`` modules) and
cites the reference file:line (relative to the reference repo root) whose
arithmetic it restates.  Parity is PINNED: ``oracle/make_golden.py`` imports the
real reference classes in the build container and writes golden vectors to
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against them.  The diffusion section (schedule, q_sample, eps-loss, DDIM,
timestep embedding, denoiser wiring) has NO reference counterpart (SURVEY.md
§0.1) -> "parity unpinned" by the reference for that section; it follows the
DDPM (Ho et al. 2020) / DDIM (Song et al. 2021) definitions and is pinned only
by literature known-answers in ``tests/test_oracle_golden.py``.

All functions are dtype-generic: run in float64 for checking, float32 for the
CPU-baseline timing.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

# --------------------------------------------------------------------------
# tensor-dict contract (src/data/AddBiomechanicsDataset.py:9-42)
# --------------------------------------------------------------------------
INPUT_KEY_ORDER = [  # concat order: src/models/FeedForwardRegressionBaseline.py:97-107
    "pos", "vel", "acc",
    "rootLinearVelInRootFrame", "rootAngularVelInRootFrame",
    "rootLinearAccInRootFrame", "rootAngularAccInRootFrame",
    "jointCentersInRootFrame",
    "rootPosHistoryInRootFrame", "rootEulerHistoryInRootFrame",
]
K_COP = "groundContactCenterOfPressureInRootFrame"
K_FORCE = "groundContactForceInRootFrame"
K_TORQUE = "groundContactTorqueInRootFrame"
K_WRENCH = "groundContactWrenchesInRootFrame"


def input_widths(num_dofs: int, stride: int) -> List[int]:
    """Per-key channel widths asserted at FeedForwardRegressionBaseline.py:83-94."""
    return [num_dofs, num_dofs, num_dofs, 3, 3, 3, 3, 36, 3 * stride, 3 * stride]


def det_fill(shape: Sequence[int], seed: float, scale: float = 1.0,
             dtype=torch.float64) -> torch.Tensor:
    """Closed-form deterministic tensor (no RNG, no weight blobs in fixtures)."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.float64)
    v = torch.sin(idx * 0.7390851332151607 + float(seed) * 1.6180339887498949)
    v = v + 0.5 * torch.cos(idx * 0.2718281828459045 + float(seed))
    return (scale * v).reshape(tuple(shape)).to(dtype)


# --------------------------------------------------------------------------
# activations (src/models/FeedForwardRegressionBaseline.py:7-11; silu is
# BUILD-DEFINED, added for the diffusion blocks)
# --------------------------------------------------------------------------
def act(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "relu":
        return torch.clamp_min(x, 0)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return 1.0 / (1.0 + torch.exp(-x))
    if name == "silu":
        return x / (1.0 + torch.exp(-x))
    if name == "elu":                                   # torch.nn.ELU(alpha=1): x > 0 ? x : exp(x) - 1
        return torch.where(x > 0, x, torch.exp(torch.clamp_max(x, 0)) - 1.0)
    if name in ("none", "identity"):
        return x
    raise KeyError(name)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Linear: y = x W^T + b, W is [out, in]."""
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


# --------------------------------------------------------------------------
# FeedForwardBaseline (src/models/FeedForwardRegressionBaseline.py:20-121)
# --------------------------------------------------------------------------
def feedforward_sizes(num_dofs: int, num_contact_bodies: int, history_len: int, stride: int,
                      output_data_format: str = "all_frames") -> Tuple[int, int, int]:
    """input_size / output_size / num_output_frames: FeedForwardRegressionBaseline.py:52,61-62."""
    frames = history_len // stride
    input_size = (3 * num_dofs + 4 * 3 + 2 * stride * 3 + 12 * 3) * frames
    nof = frames if output_data_format == "all_frames" else 1
    output_size = num_contact_bodies * (3 * 3 + 6) * nof
    return input_size, output_size, nof


def feedforward_forward(layers: List[Tuple[torch.Tensor, torch.Tensor]],
                        inputs: Dict[str, torch.Tensor], activation: str,
                        num_output_frames: int) -> Dict[str, torch.Tensor]:
    """forward(): FeedForwardRegressionBaseline.py:80-121 (no dropout / batchnorm: flags
    default off, train.py:43,47).  ``layers`` = [(weight[out,in], bias[out]), ...]."""
    x = torch.cat([inputs[k] for k in INPUT_KEY_ORDER], dim=-1)        # :97-107
    B = x.shape[0]
    x = x.reshape(B, -1)                                               # frame-major flatten
    n = len(layers)
    for i, (w, b) in enumerate(layers):                                # :68-75
        x = linear(x, w, b)
        if i < n - 1:
            x = act(activation, x)
    F = num_output_frames                                              # :116-121
    return {
        K_COP: x[:, 0 * F:6 * F].reshape(B, F, 6),
        K_FORCE: x[:, 6 * F:12 * F].reshape(B, F, 6),
        K_TORQUE: x[:, 12 * F:18 * F].reshape(B, F, 6),
        K_WRENCH: x[:, 18 * F:30 * F].reshape(B, F, 12),
    }


def batchnorm1d(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor,
                running_var: torch.Tensor, training: bool, momentum: float = 0.1, eps: float = 1e-5):
    """nn.BatchNorm1d(h0) over [B, C] with torch defaults, as the reference inserts it
    (FeedForwardRegressionBaseline.py:71-72).  Arithmetic lives in torch (third-party): restated from
    the published definition -- train: batch mean / BIASED variance normalise, the running statistics
    move by `momentum` towards the batch mean / UNBIASED variance; eval: running statistics normalise.
    Returns (y, new_running_mean, new_running_var)."""
    if training:
        B = x.shape[0]
        mean = x.mean(dim=0)
        var = ((x - mean) ** 2).mean(dim=0)
        new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
        new_rv = (1 - momentum) * running_var + momentum * var.detach() * B / (B - 1)
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, new_rm, new_rv


def feedforward_forward_opts(layers: List[Tuple[torch.Tensor, torch.Tensor]], inputs: Dict[str, torch.Tensor],
                             activation: str, num_output_frames: int,
                             bn: Optional[List[Optional[Dict[str, torch.Tensor]]]] = None, training: bool = False,
                             drop_masks: Optional[List[Optional[torch.Tensor]]] = None):
    """forward() with the optional layers of FeedForwardRegressionBaseline.py:68-75: per layer
    ``[Dropout(p)] [BatchNorm1d(h0)] Linear``, activation on all but the last.  ``bn[i]`` = dict(weight, bias,
    running_mean, running_var) or None; ``drop_masks[i]`` = the SCALED keep mask (0 or 1/(1-p)) of layer i's
    Dropout, or None (eval mode / no dropout) -- torch draws it from its global generator, so a test
    recovers the mask the implementation under test drew and hands it over.
    Returns (outputs, [(new_running_mean, new_running_var) or None per layer])."""
    x = torch.cat([inputs[k] for k in INPUT_KEY_ORDER], dim=-1)
    B = x.shape[0]
    x = x.reshape(B, -1)
    n = len(layers)
    stats = []
    for i, (w, b) in enumerate(layers):
        if drop_masks is not None and drop_masks[i] is not None:
            x = x * drop_masks[i]
        if bn is not None and bn[i] is not None:
            q = bn[i]
            x, rm, rv = batchnorm1d(x, q["weight"], q["bias"], q["running_mean"], q["running_var"], training)
            stats.append((rm, rv))
        else:
            stats.append(None)
        x = linear(x, w, b)
        if i < n - 1:
            x = act(activation, x)
    F = num_output_frames
    out = {
        K_COP: x[:, 0 * F:6 * F].reshape(B, F, 6),
        K_FORCE: x[:, 6 * F:12 * F].reshape(B, F, 6),
        K_TORQUE: x[:, 12 * F:18 * F].reshape(B, F, 6),
        K_WRENCH: x[:, 18 * F:30 * F].reshape(B, F, 12),
    }
    return out, stats


# --------------------------------------------------------------------------
# Groundlink (src/models/Groundlink.py:19-156)
# --------------------------------------------------------------------------
def conv1d_replicate(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """torch.nn.Conv1d(C_in, C_out, k, padding=k//2, padding_mode="replicate") on channels-LAST data
    (Groundlink.py:41): x [N,F,C_in], w [C_out,C_in,k] -> y [N,F,C_out],
    y[n,f,o] = b[o] + sum_{c,j} w[o,c,j] * x[n, clamp(f + j - k//2, 0, F-1), c]  (explicit gather + matmul)."""
    N, F, C = x.shape
    k = w.shape[2]
    idx = (torch.arange(F)[:, None] + torch.arange(k)[None, :] - k // 2).clamp(0, F - 1)      # [F,k]
    col = x[:, idx, :]                                       # [N,F,k,C]
    col = col.permute(0, 1, 3, 2).reshape(N, F, C * k)       # column index c*k + j == w.view(C_out, C_in*k)
    y = col @ w.reshape(w.shape[0], C * k).transpose(0, 1)
    return y if b is None else y + b


def groundlink_forward(p: Dict[str, torch.Tensor], inputs: Dict[str, torch.Tensor],
                       output_data_format: str = "all_frames") -> Dict[str, torch.Tensor]:
    """Groundlink.forward (Groundlink.py:105-156) in eval mode (Dropout = identity; cnn_dropout = 0.0 and
    fc_dropout = 0.2 only act in train mode, :20): concat of the ten input keys (:122-133), four
    Conv1d(k=7, replicate) + ELU (:41-48, channels 177 -> 128 -> 128 -> 256 -> 256 at the reference sizes), then per
    frame Linear+ELU x2 and Linear(256 -> 30, no bias) (:51-62); `last_frame` feeds only the last frame to the
    fully-connected part (:145-148); outputs are slices of the last dimension (:151-156).
    p: the reference state_dict (cnn.{1,4,7,10}.{weight,bias}, fc.{2,5}.{weight,bias}, fc.8.weight)."""
    x = torch.cat([inputs[k] for k in INPUT_KEY_ORDER], dim=-1)           # [N,F,C0]
    for i in (1, 4, 7, 10):
        x = act("elu", conv1d_replicate(x, p[f"cnn.{i}.weight"], p[f"cnn.{i}.bias"]))
    if output_data_format != "all_frames":
        x = x[:, -1:, :]
    for i in (2, 5):
        x = act("elu", linear(x, p[f"fc.{i}.weight"], p[f"fc.{i}.bias"]))
    x = linear(x, p["fc.8.weight"], None)
    return {K_COP: x[:, :, 0:6], K_FORCE: x[:, :, 6:12], K_TORQUE: x[:, :, 12:18], K_WRENCH: x[:, :, 18:30]}


def groundlink_param_shapes(num_dofs: int = 23, num_joints: int = 12, root_history_len: int = 10, k: int = 7
                            ) -> Dict[str, Tuple[int, ...]]:
    c0 = num_dofs * 3 + 12 + num_joints * 3 + root_history_len * 6        # Groundlink.py:26
    feats = [c0, 128, 128, 256, 256]
    s: Dict[str, Tuple[int, ...]] = {}
    for i, (ci, co) in zip((1, 4, 7, 10), zip(feats[:-1], feats[1:])):
        s[f"cnn.{i}.weight"] = (co, ci, k)
        s[f"cnn.{i}.bias"] = (co,)
    for i in (2, 5):
        s[f"fc.{i}.weight"] = (256, 256)
        s[f"fc.{i}.bias"] = (256,)
    s["fc.8.weight"] = (30, 256)
    return s


# --------------------------------------------------------------------------
# TransformerLayer (src/models/TransformerBaseline.py:8-38)
# --------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (TransformerBaseline.py:21-22)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def attention_core(qkv: torch.Tensor, num_heads: int, prob_mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """the core of nn.MultiheadAttention between its two projections (TransformerBaseline.py:12-13,29): qkv [B,T,3d]
    (packed q | k | v) -> per-head softmax(QK^T/sqrt(d_h))V -> [B,T,d].  Split out of mha_forward so that a test can hold
    a kernel's attention to the in-projection values the kernel itself stored."""
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // num_heads
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    q = q.reshape(B, T, num_heads, dh).transpose(1, 2)                 # [B,H,T,dh]
    k = k.reshape(B, T, num_heads, dh).transpose(1, 2)
    v = v.reshape(B, T, num_heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s - s.max(dim=-1, keepdim=True).values
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    if prob_mult is not None:
        p = p * prob_mult
    return (p @ v).transpose(1, 2).reshape(B, T, d)


def mha_forward(x: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor,
                out_w: torch.Tensor, out_b: torch.Tensor, num_heads: int,
                prob_mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True) self-attention, no mask
    (TransformerBaseline.py:12-13,29): packed in-proj [3d,d], per-head
    softmax(QK^T/sqrt(d_h))V, out-proj.  prob_mult [B,H,T,T] (train mode, dropout=p at :13): the
    dropout multipliers (0 or 1/(1-p)) applied to the NORMALISED probabilities before P.V."""
    qkv = linear(x, in_w, in_b)                                        # [B,T,3d]
    return linear(attention_core(qkv, num_heads, prob_mult), out_w, out_b)


TL_KEYS = [  # state_dict names of the reference TransformerLayer
    "multihead_attention.in_proj_weight", "multihead_attention.in_proj_bias",
    "multihead_attention.out_proj.weight", "multihead_attention.out_proj.bias",
    "feedforward.0.weight", "feedforward.0.bias", "feedforward.2.weight", "feedforward.2.bias",
    "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
]


def transformer_layer_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, num_heads: int,
                              prefix: str = "", masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """TransformerLayer.forward: TransformerBaseline.py:24-38 (post-norm, ReLU FFN).  masks (train mode with
    dropout != 0): the multipliers (0 or 1/(1-p)) of the three dropouts -- "attn" [B,H,T,T] on the attention
    probabilities (:12-13), "drop1" [B,T,d] on the attention block's output (:30), "drop2" [B,T,d] on the
    feedforward output (:35); None = eval mode / dropout 0."""
    g = lambda k: p[prefix + k]
    masks = masks or {}
    a = mha_forward(x, g("multihead_attention.in_proj_weight"), g("multihead_attention.in_proj_bias"),
                    g("multihead_attention.out_proj.weight"), g("multihead_attention.out_proj.bias"),
                    num_heads, prob_mult=masks.get("attn"))
    if "drop1" in masks:
        a = a * masks["drop1"]
    x = layer_norm(x + a, g("norm1.weight"), g("norm1.bias"))
    z1 = linear(x, g("feedforward.0.weight"), g("feedforward.0.bias"))
    # masks["relu"] (0 / 1, [B,T,ffn]): the ReLU gate taken from somewhere else (a test hands over the gate the bf16
    # KERNEL used: where a pre-activation lies within bf16 rounding of zero the two gates differ, and with them whole
    # (token, unit) contributions of the FFN input layer's gradient); None = the reference's own ReLU (:17-20)
    f = linear(z1 * masks["relu"] if "relu" in masks else act("relu", z1),
               g("feedforward.2.weight"), g("feedforward.2.bias"))
    if "drop2" in masks:
        f = f * masks["drop2"]
    return layer_norm(x + f, g("norm2.weight"), g("norm2.bias"))


# --------------------------------------------------------------------------
# RegressionLossEvaluator arithmetic (src/loss/RegressionLossEvaluator.py)
# --------------------------------------------------------------------------
def squared_diff_mean_vector(o: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    """get_squared_diff_mean_vector: RegressionLossEvaluator.py:73-83."""
    if o.shape != l.shape:
        raise ValueError("Output and label tensors must have the same shape")
    if o.dim() != 3:
        raise ValueError("Output and label tensors must be 3-dimensional")
    if o.numel() == 0:
        raise ValueError("Output and label tensors must not be empty")
    return ((o - l) ** 2).mean(dim=(0, 1))


def mask_by_threes(t: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
    """get_mask_by_threes: RegressionLossEvaluator.py:85-108 (strict '>')."""
    if t.dim() != 3:
        raise ValueError("Mask tensor must be 3-dimensional")
    if t.numel() == 0:
        raise ValueError("Mask tensor must not be empty")
    if t.shape[-1] % 3 != 0:
        raise ValueError("Mask tensor must have a final dimension divisible by 3")
    with torch.no_grad():
        r = t.reshape(t.shape[0], t.shape[1], -1, 3)
        n = torch.sqrt((r * r).sum(dim=-1))
        m = (n > threshold).to(t.dtype)
        return m.unsqueeze(3).expand(-1, -1, -1, 3).reshape(t.shape)


def mean_norm_error(o: torch.Tensor, l: torch.Tensor, vec_size: int = 3) -> torch.Tensor:
    """get_mean_norm_error: RegressionLossEvaluator.py:119-141 (LAST FRAME ONLY, :136)."""
    if o.shape != l.shape:
        raise ValueError("Output and label tensors must have the same shape")
    if o.dim() != 3:
        raise ValueError("Output and label tensors must be 3-dimensional")
    if o.numel() == 0:
        raise ValueError("Output and label tensors must not be empty")
    if o.shape[-1] % vec_size != 0:
        raise ValueError("Tensors must have a final dimension divisible by vec_size=" + str(vec_size))
    d = (o - l).reshape(o.shape[0], o.shape[1], -1, vec_size)[:, -1:, :, :]
    return torch.sqrt((d * d).sum(dim=3)).mean()


def com_acc_error(o: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    """get_com_acc_error: RegressionLossEvaluator.py:143-158."""
    if o.shape != l.shape:
        raise ValueError("Output and label tensors must have the same shape")
    if o.dim() != 3:
        raise ValueError("Output and label tensors must be 3-dimensional")
    if o.numel() == 0:
        raise ValueError("Output and label tensors must not be empty")
    if o.shape[-1] != 6:
        raise ValueError("Output and label tensors must have a 6 dimensional final dimension")
    return mean_norm_error(o[:, :, :3] + o[:, :, 3:], l[:, :, :3] + l[:, :, 3:], vec_size=3)


def regression_loss(outputs: Dict[str, torch.Tensor], labels: Dict[str, torch.Tensor],
                    grf: Sequence[int], cop: Sequence[int], moment: Sequence[int],
                    wrench: Sequence[int]):
    """RegressionLossEvaluator.__call__ steps 1-2.2: RegressionLossEvaluator.py:184-263.
    Returns (loss, {force,moment,wrench,cop loss vectors}, {7 metrics})."""
    force = squared_diff_mean_vector(outputs[K_FORCE], labels[K_FORCE])          # :184-188
    mom = squared_diff_mean_vector(outputs[K_TORQUE], labels[K_TORQUE])          # :190-194
    wr = squared_diff_mean_vector(outputs[K_WRENCH], labels[K_WRENCH])           # :198-202
    mask = mask_by_threes(labels[K_FORCE], threshold=10.0)                       # :205-209
    cp = squared_diff_mean_vector(outputs[K_COP] * mask, labels[K_COP] * mask)   # :210-214
    idx = lambda v, i: v[list(i)].sum() if len(i) else v.new_zeros(())
    loss = idx(force, grf) + idx(cp, cop) + idx(mom, moment) + idx(wr, wrench)   # :217-220
    with torch.no_grad():                                                        # :230-263
        m1 = mean_norm_error(outputs[K_WRENCH][:, :, :3], labels[K_WRENCH][:, :, :3], 3)
        m2 = mean_norm_error(outputs[K_WRENCH][:, :, 6:9], labels[K_WRENCH][:, :, 6:9], 3)
        metrics = {
            "force": mean_norm_error(outputs[K_FORCE], labels[K_FORCE]),
            "moment": mean_norm_error(outputs[K_TORQUE], labels[K_TORQUE]),
            "cop": mean_norm_error(outputs[K_COP] * mask, labels[K_COP] * mask),
            "wrench_moment": (m1 + m2) / 2.0,
            "wrench": mean_norm_error(outputs[K_WRENCH], labels[K_WRENCH], vec_size=6),
            "com_acc": com_acc_error(outputs[K_FORCE], labels[K_FORCE]),
        }
    return loss, {"force": force, "moment": mom, "wrench": wr, "cop": cp}, metrics


# --------------------------------------------------------------------------
# optimizers: torch.optim defaults with only lr set (src/cli/train.py:183-194).
# The arithmetic lives in torch (third-party, torch~=2.1.0, requirements.txt:2);
# restated from the published update rules and pinned by fixtures generated from
# torch.optim itself.  state: dict of tensors, mutated in place; step is 1-based.
# --------------------------------------------------------------------------
def optim_init_state(opt: str, p: torch.Tensor) -> Dict[str, torch.Tensor]:
    z = lambda: torch.zeros_like(p)
    return {"sgd": {}, "adam": {"m": z(), "v": z()}, "rmsprop": {"sq": z()},
            "adagrad": {"sum": z()}, "adadelta": {"sq": z(), "acc": z()},
            "adamax": {"m": z(), "u": z()}}[opt]


def optim_step(opt: str, p: torch.Tensor, g: torch.Tensor, st: Dict[str, torch.Tensor],
               lr: float, step: int) -> torch.Tensor:
    if opt == "sgd":
        return p - lr * g
    if opt == "adam":                       # betas (0.9, 0.999), eps 1e-8
        b1, b2, eps = 0.9, 0.999, 1e-8
        st["m"] = b1 * st["m"] + (1 - b1) * g
        st["v"] = b2 * st["v"] + (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = torch.sqrt(st["v"]) / math.sqrt(bc2) + eps
        return p - (lr / bc1) * st["m"] / denom
    if opt == "rmsprop":                    # alpha 0.99, eps 1e-8, momentum 0, not centered
        a, eps = 0.99, 1e-8
        st["sq"] = a * st["sq"] + (1 - a) * g * g
        return p - lr * g / (torch.sqrt(st["sq"]) + eps)
    if opt == "adagrad":                    # lr_decay 0, eps 1e-10, initial accumulator 0
        st["sum"] = st["sum"] + g * g
        return p - lr * g / (torch.sqrt(st["sum"]) + 1e-10)
    if opt == "adadelta":                   # rho 0.9, eps 1e-6
        rho, eps = 0.9, 1e-6
        st["sq"] = rho * st["sq"] + (1 - rho) * g * g
        delta = torch.sqrt(st["acc"] + eps) / torch.sqrt(st["sq"] + eps) * g
        st["acc"] = rho * st["acc"] + (1 - rho) * delta * delta
        return p - lr * delta
    if opt == "adamax":                     # betas (0.9, 0.999), eps 1e-8
        b1, b2, eps = 0.9, 0.999, 1e-8
        st["m"] = b1 * st["m"] + (1 - b1) * g
        st["u"] = torch.maximum(b2 * st["u"], g.abs() + eps)
        return p - (lr / (1 - b1 ** step)) * st["m"] / st["u"]
    raise KeyError(opt)


# --------------------------------------------------------------------------
# [BUILD-DEFINED] diffusion wrapper -- no reference counterpart.
# DDPM linear schedule, eps-prediction loss, DDIM eta=0.
# --------------------------------------------------------------------------
def linear_beta_schedule(num_steps: int = 1000, beta_start: float = 1e-4,
                         beta_end: float = 0.02) -> torch.Tensor:
    return torch.linspace(beta_start, beta_end, num_steps, dtype=torch.float64)


def alphas_cumprod(betas: torch.Tensor) -> torch.Tensor:
    return torch.cumprod(1.0 - betas.to(torch.float64), dim=0)


def schedule_tables(num_steps: int = 1000) -> Dict[str, torch.Tensor]:
    """float64 tables; the product casts them ONCE to fp32 -> bit-exactness target."""
    ab = alphas_cumprod(linear_beta_schedule(num_steps))
    return {"alphas_cumprod": ab, "sqrt_ab": torch.sqrt(ab), "sqrt_1mab": torch.sqrt(1.0 - ab)}


def ddim_timesteps(num_train_steps: int = 1000, num_sample_steps: int = 100) -> torch.Tensor:
    """int64 descending subsequence: 990, 980, ..., 0 for (1000, 100)."""
    stride = num_train_steps // num_sample_steps
    return torch.arange(num_sample_steps - 1, -1, -1, dtype=torch.int64) * stride


def q_sample(x0: torch.Tensor, t: torch.Tensor, eps: torch.Tensor,
             tables: Dict[str, torch.Tensor]) -> torch.Tensor:
    a = tables["sqrt_ab"].to(x0.dtype)[t].reshape(-1, 1, 1)
    s = tables["sqrt_1mab"].to(x0.dtype)[t].reshape(-1, 1, 1)
    return a * x0 + s * eps


def timestep_embedding(t: torch.Tensor, dim: int, dtype=torch.float64,
                       max_period: float = 10000.0) -> torch.Tensor:
    """[sin(t*w_i), cos(t*w_i)], w_i = exp(-ln(max_period) * i / half), i = 0..half-1."""
    half = dim // 2
    w = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float64) / half)
    a = t.to(torch.float64).reshape(-1, 1) * w.reshape(1, -1)
    return torch.cat([torch.sin(a), torch.cos(a)], dim=-1).to(dtype)


def eps_mse(pred: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    return ((pred - eps) ** 2).mean()


def ddim_step(x_t: torch.Tensor, eps_pred: torch.Tensor, ab_t, ab_prev) -> torch.Tensor:
    """eta = 0: x0 = (x_t - sqrt(1-ab_t) eps)/sqrt(ab_t);  x_prev = sqrt(ab_prev) x0 + sqrt(1-ab_prev) eps."""
    ab_t = torch.as_tensor(ab_t, dtype=x_t.dtype)
    ab_prev = torch.as_tensor(ab_prev, dtype=x_t.dtype)
    x0 = (x_t - torch.sqrt(1 - ab_t) * eps_pred) / torch.sqrt(ab_t)
    return torch.sqrt(ab_prev) * x0 + torch.sqrt(1 - ab_prev) * eps_pred


def ddim_coeffs(num_train_steps: int = 1000, num_sample_steps: int = 100) -> torch.Tensor:
    """Per-step (c_x, c_eps) so that x_prev = c_x * x_t + c_eps * eps (float64, [S,2])."""
    ab = alphas_cumprod(linear_beta_schedule(num_train_steps))
    ts = ddim_timesteps(num_train_steps, num_sample_steps)
    out = []
    for i, t in enumerate(ts.tolist()):
        ab_t = ab[t]
        ab_p = ab[ts[i + 1]] if i + 1 < len(ts) else torch.tensor(1.0, dtype=torch.float64)
        cx = torch.sqrt(ab_p) / torch.sqrt(ab_t)
        ce = torch.sqrt(1 - ab_p) - torch.sqrt(ab_p) * torch.sqrt(1 - ab_t) / torch.sqrt(ab_t)
        out.append(torch.stack([cx, ce]))
    return torch.stack(out)


def time_mlp(p: Dict[str, torch.Tensor], t: torch.Tensor, temb_dim: int, dtype) -> torch.Tensor:
    s = timestep_embedding(t, temb_dim, dtype)
    u = act("silu", linear(s, p["time_mlp.0.weight"], p["time_mlp.0.bias"]))
    return linear(u, p["time_mlp.2.weight"], p["time_mlp.2.bias"])


def denoiser_mlp_forward(p: Dict[str, torch.Tensor], x_t: torch.Tensor, t: torch.Tensor,
                         hidden: Sequence[int], temb_dim: int = 128) -> torch.Tensor:
    """Token-wise MLP denoiser (BASELINE config 2):
    e = time_mlp(t) [B, sum(hidden)]; per block i: h = LN(SiLU(W_i h + b_i + e[:, off_i:off_i+h_i]));
    out = W_head h + b_head."""
    e = time_mlp(p, t, temb_dim, x_t.dtype)
    h, off = x_t, 0
    for i, hd in enumerate(hidden):
        z = linear(h, p[f"blocks.{i}.linear.weight"], p[f"blocks.{i}.linear.bias"]) \
            + e[:, None, off:off + hd]
        h = layer_norm(act("silu", z), p[f"blocks.{i}.norm.weight"], p[f"blocks.{i}.norm.bias"])
        off += hd
    return linear(h, p["head.weight"], p["head.bias"])


def denoiser_transformer_forward(p: Dict[str, torch.Tensor], x_t: torch.Tensor, t: torch.Tensor,
                                 num_layers: int, num_heads: int, temb_dim: int = 128,
                                 layer_masks: Optional[List[Dict[str, torch.Tensor]]] = None) -> torch.Tensor:
    """Transformer denoiser (BASELINE configs 3-5): in-proj over concat(x, pos_emb[frame]) (the
    reference concatenates a learned per-frame embedding, TransformerBaseline.py:119-126) plus the
    time embedding, `num_layers` reference TransformerLayers, out-proj."""
    B, T, D = x_t.shape
    e = time_mlp(p, t, temb_dim, x_t.dtype)                                  # [B, d_model]
    pos = p["temporal_embedding.embedding.weight"][:T]                       # [T, pos_dim]
    xin = torch.cat([x_t, pos.unsqueeze(0).expand(B, -1, -1)], dim=-1)
    h = linear(xin, p["in_proj.weight"], p["in_proj.bias"]) + e[:, None, :]
    for l in range(num_layers):
        h = transformer_layer_forward(p, h, num_heads, prefix=f"transformer_layers.{l}.",
                                      masks=None if layer_masks is None else layer_masks[l])
    return linear(h, p["out_proj.weight"], p["out_proj.bias"])


def denoiser_mlp_param_shapes(feat: int, hidden: Sequence[int], temb_dim: int = 128,
                              temb_hidden: int = 512) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {
        "time_mlp.0.weight": (temb_hidden, temb_dim), "time_mlp.0.bias": (temb_hidden,),
        "time_mlp.2.weight": (sum(hidden), temb_hidden), "time_mlp.2.bias": (sum(hidden),),
    }
    prev = feat
    for i, hd in enumerate(hidden):
        s[f"blocks.{i}.linear.weight"] = (hd, prev)
        s[f"blocks.{i}.linear.bias"] = (hd,)
        s[f"blocks.{i}.norm.weight"] = (hd,)
        s[f"blocks.{i}.norm.bias"] = (hd,)
        prev = hd
    s["head.weight"] = (feat, prev)
    s["head.bias"] = (feat,)
    return s


def denoiser_transformer_param_shapes(feat: int, window: int, d_model: int = 512, ffn: int = 2048,
                                      num_layers: int = 4, pos_dim: int = 30, temb_dim: int = 128,
                                      temb_hidden: int = 512) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {
        "time_mlp.0.weight": (temb_hidden, temb_dim), "time_mlp.0.bias": (temb_hidden,),
        "time_mlp.2.weight": (d_model, temb_hidden), "time_mlp.2.bias": (d_model,),
        "temporal_embedding.embedding.weight": (window, pos_dim),
        "in_proj.weight": (d_model, feat + pos_dim), "in_proj.bias": (d_model,),
    }
    for l in range(num_layers):
        pre = f"transformer_layers.{l}."
        s[pre + "multihead_attention.in_proj_weight"] = (3 * d_model, d_model)
        s[pre + "multihead_attention.in_proj_bias"] = (3 * d_model,)
        s[pre + "multihead_attention.out_proj.weight"] = (d_model, d_model)
        s[pre + "multihead_attention.out_proj.bias"] = (d_model,)
        s[pre + "feedforward.0.weight"] = (ffn, d_model)
        s[pre + "feedforward.0.bias"] = (ffn,)
        s[pre + "feedforward.2.weight"] = (d_model, ffn)
        s[pre + "feedforward.2.bias"] = (d_model,)
        for n in ("norm1", "norm2"):
            s[pre + n + ".weight"] = (d_model,)
            s[pre + n + ".bias"] = (d_model,)
    s["out_proj.weight"] = (feat, d_model)
    s["out_proj.bias"] = (feat,)
    return s


def det_params(shapes: Dict[str, Tuple[int, ...]], dtype=torch.float64, seed0: float = 1.0
               ) -> Dict[str, torch.Tensor]:
    """Deterministic parameters: weights ~ det_fill scaled by 1/sqrt(fan_in), norm weights near 1."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        if k.endswith("norm.weight") or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            out[k] = (1.0 + det_fill(shp, seed0 + i, 0.1)).to(dtype)
        elif len(shp) == 1:
            out[k] = det_fill(shp, seed0 + i, 0.05).to(dtype)
        else:
            out[k] = det_fill(shp, seed0 + i, 1.0 / math.sqrt(shp[-1])).to(dtype)
    return out


def ddim_sample(eps_fn, x_T: torch.Tensor, num_train_steps: int = 1000,
                num_sample_steps: int = 100) -> torch.Tensor:
    """The sampling loop: for t in ddim_timesteps: eps = eps_fn(x, t); x = ddim update."""
    ab = alphas_cumprod(linear_beta_schedule(num_train_steps))
    ts = ddim_timesteps(num_train_steps, num_sample_steps)
    x = x_T
    B = x.shape[0]
    for i, t in enumerate(ts.tolist()):
        tt = torch.full((B,), t, dtype=torch.int64)
        eps = eps_fn(x, tt)
        ab_p = ab[ts[i + 1]] if i + 1 < len(ts) else torch.tensor(1.0, dtype=torch.float64)
        x = ddim_step(x, eps, ab[t].to(x.dtype), ab_p.to(x.dtype))
    return x


# =====================================================================================================
# [BUILD-DEFINED] counter-based noise for the diffusion training step (no reference counterpart: the reference has no
# diffusion path).  Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as
# 1, 2, 3", SC'11; constants and known answers from the Random123 distribution's kat_vectors) -- pinned by those known
# answers in tests/test_oracle_golden.py.  The kernel (csrc/diffusion.hip::diffusion_draw_kernel) must produce exactly
# these 32-bit words and timestep indices; its normals are Box-Muller over the same words in float32 hardware
# transcendentals and are compared with `philox_normals` within a stated tolerance.
# =====================================================================================================
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
DRAW_DOMAIN_EPS, DRAW_DOMAIN_T = 0, 1       # counter word 3: which quantity a block of words feeds


def philox4x32(counter, key, rounds: int = 10):
    """counter: uint32 array [..., 4]; key: (k0, k1).  Returns uint32 [..., 4]."""
    import numpy as np
    c = np.asarray(counter, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = PHILOX_M0 * c0                       # 64-bit products
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def draw_words(n_blocks: int, seed: int, step: int, stream: int, domain: int):
    """words of blocks 0..n_blocks-1: counter = (block, step, stream, domain), key = (seed low, seed high)"""
    import numpy as np
    ctr = np.zeros((n_blocks, 4), dtype=np.uint64)
    ctr[:, 0] = np.arange(n_blocks, dtype=np.uint64)
    ctr[:, 1], ctr[:, 2], ctr[:, 3] = step & 0xFFFFFFFF, stream & 0xFFFFFFFF, domain
    return philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def draw_timesteps(B: int, num_train_steps: int, seed: int, step: int, stream: int):
    """t[b] = floor(word0(block b) * num_train_steps / 2^32): uniform on {0 .. num_train_steps-1}; int64 [B].  Bit-exact
    contract with the kernel."""
    import numpy as np
    w = draw_words(B, seed, step, stream, DRAW_DOMAIN_T)[:, 0].astype(np.uint64)
    return torch.from_numpy(((w * np.uint64(num_train_steps)) >> np.uint64(32)).astype(np.int64))


def philox_normals(n: int, seed: int, step: int, stream: int):
    """float64 N(0,1) values 0..n-1: element e lives in block e // 4; words (w0, w1) give elements 4b, 4b+1 and (w2, w3)
    give 4b+2, 4b+3 by Box-Muller on the words' top 24 bits (exact in float32): u1 = ((w >> 8) + 1) / 2^24 in (0, 1],
    u2 = (w' >> 8) / 2^24 in [0, 1), r = sqrt(-2 ln u1), z = r cos(2 pi u2), r sin(2 pi u2)  (|z| <= 5.77)."""
    import numpy as np
    nb = (n + 3) // 4
    w = (draw_words(nb, seed, step, stream, DRAW_DOMAIN_EPS) >> np.uint32(8)).astype(np.float64)
    out = np.empty((nb, 4), dtype=np.float64)
    for a in (0, 2):
        r = np.sqrt(-2.0 * np.log((w[:, a] + 1.0) * 2.0 ** -24))
        th = 2.0 * np.pi * (w[:, a + 1] * 2.0 ** -24)
        out[:, a], out[:, a + 1] = r * np.cos(th), r * np.sin(th)
    return torch.from_numpy(out.reshape(-1)[:n])
