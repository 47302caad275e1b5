"""[BUILD-DEFINED] diffusion denoisers (registry names ``diffusion-mlp`` / ``diffusion-transformer``).

The reference snapshot ships no diffusion code; these are the networks BASELINE.json's
configs 2-5 name, built from the reference's layer arithmetic (Linear, LayerNorm, TransformerLayer,
learned per-frame embedding concatenated to the features -- TransformerBaseline.py:119-126) plus a
sinusoidal timestep embedding and SiLU.  Call convention: ``model(x_t [B,T,D], t [B] int64) -> eps_hat``.
"""
from typing import List, Sequence

import torch
import torch.nn as nn

from ..diffusion.schedule import DiffusionTables
from ..module import HipModule
from ..plans import DenoiserMLPPlan, DenoiserTransformerPlan
from .TransformerBaseline import TemporalEmbedding, make_layer_params


class _DenoiserBase(HipModule):
    def __init__(self, compute_dtype, temb_dim, num_train_steps):
        super().__init__(compute_dtype)
        self.temb_dim, self.num_train_steps = temb_dim, num_train_steps
        self._tables = None
        self._plan = None

    def tables(self, device) -> DiffusionTables:
        if self._tables is None or self._tables.device != device:
            self._tables = DiffusionTables(device, self.num_train_steps, temb_dim=self.temb_dim)
        return self._tables

    def _make_plan(self, device):
        raise NotImplementedError

    def _get_plan(self, device):
        if self._plan is None or self._plan.buf.device != device or self._plan.dtype != self.compute_dtype:
            self._plan = self._make_plan(device)
        return self._plan

    def _plan_forward(self, x, t):
        out = torch.empty_like(x)
        return self._get_plan(x.device).forward(x, t, self.tables(x.device).temb, self.param_source(), out=out)

    def _plan_backward(self, dout, P, accumulate):
        self._plan.backward(dout, P, accumulate)
        return None

    def forward(self, x_t: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        self.ensure_packed()
        dev = self._flat.device
        x = x_t.to(device=dev, dtype=self.compute_dtype).contiguous()
        t = t.to(device=dev, dtype=torch.int64).contiguous()
        if x.dim() != 3 or t.shape != (x.shape[0],):
            raise AssertionError(f"expected x_t [B,T,D] and t [B]; got {tuple(x.shape)}, {tuple(t.shape)}")
        return self.run_plan(x, t)


class DiffusionMLP(_DenoiserBase):
    """Token-wise MLP denoiser: e = time_mlp(t); per block h = LN(SiLU(W h + b + e_slice)); head."""

    def __init__(self, feat_dim: int = 300, hidden_dims: Sequence[int] = (512, 512), temb_dim: int = 128,
                 temb_hidden: int = 512, num_train_steps: int = 1000, device=None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__(compute_dtype, temb_dim, num_train_steps)
        self.feat_dim, self.hidden_dims = feat_dim, list(hidden_dims)
        self.time_mlp = nn.ModuleDict({"0": nn.Linear(temb_dim, temb_hidden, device=device),
                                       "2": nn.Linear(temb_hidden, sum(self.hidden_dims), device=device)})
        blocks = []
        prev = feat_dim
        for hd in self.hidden_dims:
            blocks.append(nn.ModuleDict({"linear": nn.Linear(prev, hd, device=device),
                                         "norm": nn.LayerNorm(hd, device=device)}))
            prev = hd
        self.blocks = nn.ModuleList(blocks)
        self.head = nn.Linear(prev, feat_dim, device=device)

    def _make_plan(self, device):
        return DenoiserMLPPlan(self.hidden_dims, self.compute_dtype, device)


class DiffusionTransformer(_DenoiserBase):
    """in-proj(x ++ frame embedding) + time embedding -> N x reference TransformerLayer -> out-proj."""

    def __init__(self, feat_dim: int = 300, window: int = 50, d_model: int = 512, num_heads: int = 8,
                 dim_feedforward: int = 2048, num_layers: int = 4, temporal_embedding_dim: int = 30,
                 temb_dim: int = 128, temb_hidden: int = 512, num_train_steps: int = 1000, device=None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__(compute_dtype, temb_dim, num_train_steps)
        self.feat_dim, self.window, self.d_model, self.num_heads = feat_dim, window, d_model, num_heads
        self.ffn, self.num_layers, self.pos_dim = dim_feedforward, num_layers, temporal_embedding_dim
        self.time_mlp = nn.ModuleDict({"0": nn.Linear(temb_dim, temb_hidden, device=device),
                                       "2": nn.Linear(temb_hidden, d_model, device=device)})
        self.temporal_embedding = TemporalEmbedding(window, temporal_embedding_dim, device=device)
        self.in_proj = nn.Linear(feat_dim + temporal_embedding_dim, d_model, device=device)
        self.transformer_layers = nn.ModuleList([make_layer_params(d_model, dim_feedforward, torch.float32, device)
                                                 for _ in range(num_layers)])
        self.out_proj = nn.Linear(d_model, feat_dim, device=device)

    def _make_plan(self, device):
        return DenoiserTransformerPlan(self.feat_dim, self.pos_dim, self.d_model, self.num_heads, self.ffn,
                                       self.num_layers, self.compute_dtype, device)
