"""[BUILD-DEFINED] DDPM / DDIM schedule tables (no reference counterpart).

Everything is computed in float64 on the host exactly as the oracle does (literature definitions:
linear beta schedule 1e-4..0.02 over 1000 steps, alpha_bar = cumprod(1-beta); DDIM eta = 0; sinusoidal
embedding [sin(t w_i), cos(t w_i)], w_i = exp(-ln(1e4) i / half)), cast ONCE to fp32 and uploaded; the
kernels only index the tables, so schedule values and timestep indices are bit-exact.
"""
import math
from typing import Dict

import torch


def linear_beta_schedule(num_steps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02) -> torch.Tensor:
    return torch.linspace(beta_start, beta_end, num_steps, dtype=torch.float64)


def alphas_cumprod(num_steps: int = 1000) -> torch.Tensor:
    return torch.cumprod(1.0 - linear_beta_schedule(num_steps), dim=0)


def ddim_timesteps(num_train_steps: int = 1000, num_sample_steps: int = 100) -> torch.Tensor:
    if num_train_steps % num_sample_steps:
        raise ValueError("num_sample_steps must divide num_train_steps")
    stride = num_train_steps // num_sample_steps
    return torch.arange(num_sample_steps - 1, -1, -1, dtype=torch.int64) * stride


def ddim_coefficients(num_train_steps: int = 1000, num_sample_steps: int = 100) -> torch.Tensor:
    """[S, 2] float64 (c_x, c_eps): x_prev = c_x x_t + c_eps eps (eta = 0; last step goes to alpha_bar = 1)."""
    ab = alphas_cumprod(num_train_steps)
    ts = ddim_timesteps(num_train_steps, num_sample_steps).tolist()
    rows = []
    for i, t in enumerate(ts):
        ab_t = ab[t]
        ab_p = ab[ts[i + 1]] if i + 1 < len(ts) else torch.tensor(1.0, dtype=torch.float64)
        cx = torch.sqrt(ab_p) / torch.sqrt(ab_t)
        ce = torch.sqrt(1 - ab_p) - torch.sqrt(ab_p) * torch.sqrt(1 - ab_t) / torch.sqrt(ab_t)
        rows.append(torch.stack([cx, ce]))
    return torch.stack(rows)


def timestep_embedding_table(num_steps: int, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[num_steps, dim] float64: row t = [sin(t w), cos(t w)]."""
    half = dim // 2
    w = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float64) / half)
    a = torch.arange(num_steps, dtype=torch.float64).reshape(-1, 1) * w.reshape(1, -1)
    return torch.cat([torch.sin(a), torch.cos(a)], dim=-1)


class DiffusionTables:
    """fp32 device copies of the float64 tables (cast once)."""

    def __init__(self, device, num_train_steps: int = 1000, num_sample_steps: int = 100, temb_dim: int = 128):
        ab = alphas_cumprod(num_train_steps)
        self.num_train_steps, self.num_sample_steps, self.temb_dim = num_train_steps, num_sample_steps, temb_dim
        self.alphas_cumprod64 = ab
        self.sqrt_ab = torch.sqrt(ab).to(torch.float32).to(device)
        self.sqrt_1mab = torch.sqrt(1.0 - ab).to(torch.float32).to(device)
        self.temb = timestep_embedding_table(num_train_steps, temb_dim).to(torch.float32).to(device).contiguous()
        self.set_sampler(num_sample_steps)
        self.device = device

    def set_sampler(self, num_sample_steps: int):
        self.num_sample_steps = num_sample_steps
        dev = self.sqrt_ab.device
        self.ddim_t = ddim_timesteps(self.num_train_steps, num_sample_steps).to(dev)
        self.ddim_coef = ddim_coefficients(self.num_train_steps, num_sample_steps).to(torch.float32).to(dev).contiguous()

    def host_tables(self) -> Dict[str, torch.Tensor]:
        return {"sqrt_ab": self.sqrt_ab.cpu(), "sqrt_1mab": self.sqrt_1mab.cpu(), "temb": self.temb.cpu(),
                "ddim_t": self.ddim_t.cpu(), "ddim_coef": self.ddim_coef.cpu()}
