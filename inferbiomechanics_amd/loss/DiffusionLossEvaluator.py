"""[BUILD-DEFINED] eps-prediction loss of the diffusion wrapper: loss = mean((eps_hat - eps)^2).

One launch pair (per-block partials + fixed-order final sum) that also writes d loss/d eps_hat; the
loss value stays on the device (no host sync per step)."""
from typing import List

import torch

from .. import hip


def _as_f32_scalar(dloss: torch.Tensor) -> torch.Tensor:
    """the upstream gradient of the loss as a one-element fp32 tensor (the loss IS fp32 on the device, so this is a view)"""
    return dloss.detach().reshape(1).to(torch.float32).contiguous()


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        n = pred.numel()
        result = torch.zeros(1, dtype=torch.float32, device=pred.device)
        ws = torch.empty(hip.mse_loss_workspace_bytes(n), dtype=torch.uint8, device=pred.device)
        dpred = torch.empty_like(pred) if pred.requires_grad else None
        hip.mse_loss(pred, target, result, ws, dpred=dpred)
        ctx.dpred = dpred
        return result[0]

    @staticmethod
    def backward(ctx, dloss):
        if ctx.dpred is None:
            return None, None
        # dpred already holds 2 (pred - target) / n from the forward launch; the upstream gradient of the scalar loss is
        # a device scalar: one in-place HIP launch, no ATen arithmetic on the backward.  In place means consumed: a second
        # backward over the same graph (retain_graph=True) is refused rather than scaled twice.
        if getattr(ctx, "consumed", False):
            raise RuntimeError("eps-MSE loss: backward was already run on this graph; its gradient buffer is scaled in "
                               "place and cannot be reused (call the evaluator again for a second backward)")
        ctx.consumed = True
        return hip.scale_by_device_scalar(ctx.dpred, _as_f32_scalar(dloss)), None


class DiffusionLossEvaluator:
    def __init__(self, split: str = 'train'):
        self.split = split
        self.losses: List[torch.Tensor] = []

    def __call__(self, eps_pred: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        eps = eps.to(device=eps_pred.device, dtype=eps_pred.dtype).contiguous()
        loss = _MSEFn.apply(eps_pred.contiguous(), eps)
        self.losses.append(loss.detach())
        return loss

    def print_report(self, reset: bool = True):
        if self.losses:
            m = float(torch.stack(self.losses).mean().cpu())
            print(f'\t[{self.split}] eps-MSE: {m}')
        else:
            m = None
        if reset:
            self.losses = []
        return m
