"""nn.Module base for models whose forward/backward are HIP launch plans.

Drop-in contract kept from the reference (SURVEY.md §8b "Model call"): an ``nn.Module`` with
``.parameters()``, ``.state_dict()``, ``.train()/.eval()``, DDP-wrappable, usable with
``torch.optim`` -- so a plan is exposed to autograd as ONE node (``_PlanFunction``) whose inputs are
the module's parameters.  The parameters live in one flat fp32 buffer in HBM (views), which is what
the fused optimizer / bucketed RCCL all-reduce of the trainer operate on; in bf16 throughput mode a
flat bf16 shadow of the same layout feeds the GEMMs.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import hip
from .plans import ParamSource

_ALIGN = 64  # elements (256 B): every parameter starts on a 256-byte boundary of the flat buffer


def flat_layout(shapes: "OrderedDict[str, Tuple[int, ...]]") -> Tuple["OrderedDict[str, Tuple[int, int]]", int]:
    """name -> (offset, numel) with 256-byte aligned offsets; total length (elements, multiple of 64)."""
    off = 0
    lay: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    for k, shp in shapes.items():
        n = 1
        for s in shp:
            n *= int(s)
        lay[k] = (off, n)
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    return lay, off


class HipModule(nn.Module):
    """Base: flat parameter packing, bf16 shadows, the autograd bridge."""

    def __init__(self, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 (parity) or torch.bfloat16 (throughput)")
        self.compute_dtype = compute_dtype
        self._flat: Optional[torch.Tensor] = None
        self._shadow: Optional[torch.Tensor] = None
        self._layout = None
        self._shadow_fresh = False   # set by the trainer when its optimizer kernel refreshed the shadow
        self._generation = 0

    # ---- flat packing --------------------------------------------------------------------------
    def _packed_ok(self) -> bool:
        if self._flat is None:
            return False
        base = self._flat.data_ptr()
        for k, p in self.named_parameters():
            off, n = self._layout[k]
            if p.device != self._flat.device or p.data_ptr() != base + 4 * off or p.numel() != n:
                return False
        return True

    def pack_(self):
        """(Re)pack all parameters into one flat fp32 HBM buffer; parameters become views of it."""
        params = list(self.named_parameters())
        if not params:
            raise hip.HipError("model has no parameters")
        dev = params[0][1].device
        if dev.type != "cuda" and not hip._dry_run:
            raise hip.HipError("the HIP path needs the parameters in HBM: move the model to a 'cuda' device "
                               "(there is no CPU fallback)")
        lay, total = flat_layout(OrderedDict((k, tuple(p.shape)) for k, p in params))
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k, p in params:
                off, n = lay[k]
                v = flat[off:off + n].view(p.shape)
                v.copy_(p.data.to(torch.float32))
                p.data = v
        self._flat, self._layout, self._shadow = flat, lay, None
        return self

    def ensure_packed(self):
        if not self._packed_ok():
            self.pack_()

    def flat_view(self, flat: torch.Tensor, name: str, shape) -> torch.Tensor:
        off, n = self._layout[name]
        return flat[off:off + n].view(shape)

    def sync_shadow(self):
        if self.compute_dtype != torch.bfloat16:
            return
        if self._shadow is None or self._shadow.numel() != self._flat.numel():
            self._shadow = torch.empty(self._flat.numel(), dtype=torch.bfloat16, device=self._flat.device)
            self._shadow_fresh = False
        if not self._shadow_fresh:
            hip.cast(self._flat, self._shadow)

    def param_source(self, grads: Optional[Dict[str, torch.Tensor]] = None) -> ParamSource:
        params = dict(self.named_parameters())
        if self.compute_dtype == torch.bfloat16:
            w = lambda k: self.flat_view(self._shadow, k, params[k].shape)
        else:
            w = lambda k: params[k].data
        v = lambda k: params[k].data
        g = (lambda k: grads[k]) if grads is not None else None
        return ParamSource(w, v, g)

    # ---- to be provided by subclasses ----------------------------------------------------------
    def _plan_forward(self, *inputs) -> torch.Tensor:
        raise NotImplementedError

    def _plan_backward(self, dout: torch.Tensor, P: ParamSource, accumulate: bool):
        raise NotImplementedError

    def run_plan(self, *inputs) -> torch.Tensor:
        """Forward through the HIP plan as one autograd node over (inputs, parameters)."""
        self.ensure_packed()
        self.sync_shadow()
        params = [p for _, p in self.named_parameters()]
        return _PlanFunction.apply(self, len(inputs), *inputs, *params)


class _PlanFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: HipModule, n_in: int, *tensors):
        inputs = tensors[:n_in]
        module._generation += 1
        ctx.module, ctx.gen, ctx.n_in = module, module._generation, n_in
        ctx.in_req = [isinstance(t, torch.Tensor) and t.requires_grad for t in inputs]
        with torch.no_grad():
            out = module._plan_forward(*inputs)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.module
        if ctx.gen != m._generation:
            raise hip.HipError("backward() after a later forward(): the saved activations of this plan were "
                               "overwritten (plans keep ONE set of resident activation buffers). Run "
                               "forward->loss->backward for one batch at a time, as src/cli/train.py:240-284 does.")
        names = [k for k, _ in m.named_parameters()]
        params = dict(m.named_parameters())
        grads = {k: torch.empty_like(params[k].data) for k in names}
        with torch.no_grad():
            d = dout.to(m.compute_dtype).contiguous()
            dx = m._plan_backward(d, m.param_source(grads), False)
        in_grads: List[Optional[torch.Tensor]] = [None] * ctx.n_in
        if dx is not None:
            for i, req in enumerate(ctx.in_req):
                if req and i in dx:
                    in_grads[i] = dx[i]
        return (None, None, *in_grads, *[grads[k] for k in names])
