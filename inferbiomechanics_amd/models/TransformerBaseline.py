"""TransformerLayer on gfx950 kernels (src/models/TransformerBaseline.py:8-38).

Same constructor and state_dict names as the reference layer (``multihead_attention.in_proj_weight``,
``...out_proj.weight``, ``feedforward.{0,2}.*``, ``norm{1,2}.*``).  The reference's whole-model
``TransformerBaseline.forward`` is dead code (it needs data keys that no longer exist);
its layer arithmetic is what the transformer denoiser is built from.
dropout (the reference model passes 0.0, TransformerBaseline.py:79) acts in train mode exactly where the reference layer has
it: on the attention probabilities inside nn.MultiheadAttention(dropout=p) (:12-13), dropout1 (:30) and dropout2 (:35); the
masks are counter-based hashes regenerated in the backward (the draws differ from torch's generator, the arithmetic given a
mask does not: tests/test_transformer_dropout_gpu.py).
"""
import math

import torch
import torch.nn as nn

from .. import hip

from ..module import HipModule
from ..plans import TransformerLayerPlan


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names and initialisation."""

    def __init__(self, d: int, dtype, device=None):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d, dtype=dtype, device=device))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d, dtype=dtype, device=device))
        self.out_proj = nn.Linear(d, d, dtype=dtype, device=device)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


def make_layer_params(d: int, ffn: int, dtype=torch.float32, device=None) -> nn.ModuleDict:
    return nn.ModuleDict({
        "multihead_attention": _MHAParams(d, dtype, device),
        "feedforward": nn.ModuleDict({"0": nn.Linear(d, ffn, dtype=dtype, device=device),
                                      "2": nn.Linear(ffn, d, dtype=dtype, device=device)}),
        "norm1": nn.LayerNorm(d, dtype=dtype, device=device),
        "norm2": nn.LayerNorm(d, dtype=dtype, device=device),
    })


class TransformerLayer(HipModule):
    _instances = 0              # construction order; enters each layer's dropout seed (independent masks per layer)

    def __init__(self, timestep_vector_dim: int, num_heads: int, dim_feedforward: int, dropout: float = 0.0,
                 dtype=torch.float32, device=None, seed=None):
        super().__init__(torch.bfloat16 if dtype == torch.bfloat16 else torch.float32)
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), but got {dropout}")
        if timestep_vector_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.d, self.h, self.ffn = timestep_vector_dim, num_heads, dim_feedforward
        sub = make_layer_params(self.d, self.ffn, torch.float32, device)
        self.multihead_attention = sub["multihead_attention"]
        self.feedforward = sub["feedforward"]
        self.norm1, self.norm2 = sub["norm1"], sub["norm2"]
        self.dropout_p = float(dropout)
        self.train_mode_matters = self.dropout_p > 0.0
        self._fwd_calls = 0
        self._plan = None
        # seed=None: plans.mask_seed of (construction index, torch's seed, rank) -- stacked layers and data-parallel ranks
        # draw independent masks, as the reference's nn.Dropout modules do; an int fixes the mask stream (tests)
        self.seed = seed
        self._index = TransformerLayer._instances
        TransformerLayer._instances += 1

    def _get_plan(self, device):
        if self._plan is None or self._plan.buf.device != device or self._plan.dtype != self.compute_dtype:
            self._plan = TransformerLayerPlan("", self.d, self.h, self.ffn, self.compute_dtype, device,
                                              tag=f"tl{self._index}", dropout_p=self.dropout_p, seed=self.seed)
        return self._plan

    def _plan_forward(self, x):
        out = torch.empty_like(x)
        if self.training:
            self._fwd_calls += 1           # the masks are keyed on (seed, step, element): a fresh draw per call
        return self._get_plan(x.device).forward(x, self.param_source(), out=out, training=self.training,
                                                step=self._fwd_calls)

    def _plan_backward(self, dout, P, accumulate):
        return {0: self._plan.backward(dout, P, accumulate).clone()}

    def forward(self, x: torch.Tensor):
        self.ensure_packed()
        x = x.to(device=self._flat.device, dtype=self.compute_dtype).contiguous()
        return self.run_plan(x)


class _EmbeddingRows(torch.autograd.Function):
    """weight[idx] through ib_gather_rows; the weight gradient through ib_gather_rows_bwd (position order: deterministic)"""

    @staticmethod
    def forward(ctx, weight, idx):
        out = torch.empty((idx.numel(), weight.shape[1]), dtype=torch.float32, device=weight.device)
        hip.gather_rows(weight.detach().contiguous(), idx, out)
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dw = torch.empty((ctx.rows, dout.shape[1]), dtype=torch.float32, device=dout.device)
        hip.gather_rows_bwd(dout.detach().to(torch.float32).contiguous(), idx, dw)
        return dw, None


class TemporalEmbedding(nn.Module):
    """Learned per-frame vector (src/models/TransformerBaseline.py:41-48); looked up with arange(T).  The parameter keeps
    nn.Embedding's name (`embedding.weight`, the checkpoint grammar); a float32 table in HBM is read through the library's
    row gather (fwd) and its position-order transpose (bwd).  A host table, or the reference's float64 default, takes
    nn.Embedding's own lookup (host-only shim: the denoiser plans never call this forward -- they gather the rows inside
    their own launch sequence)."""

    def __init__(self, window_size: int, embedding_dim: int, dtype=torch.float32, device=None):
        super().__init__()
        self.embedding = nn.Embedding(window_size, embedding_dim, dtype=dtype, device=device)

    def forward(self, x):
        w = self.embedding.weight
        if (w.is_cuda or hip._dry_run) and w.dtype == torch.float32 and isinstance(x, torch.Tensor) \
                and x.dtype in (torch.int64, torch.int32):
            idx = x.to(device=w.device, dtype=torch.int64).contiguous()
            return _EmbeddingRows.apply(w, idx.reshape(-1)).reshape(tuple(idx.shape) + (w.shape[1],))
        return self.embedding(x)
