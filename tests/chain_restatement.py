"""float64 restatement of what the fused chain kernel (csrc/chain.hip) computes -- q_sample, every block
(Linear + time-embedding row + SiLU + LayerNorm), the head, the eps-MSE loss -- with an injectable rounding `bf` at the
kernel's bf16 storage points (value rounded, gradient passed straight through).  Shared by tests/test_chain_gpu.py (bf =
round to bf16: the kernel's expected tensors) and tests/test_oracle_golden.py (bf = identity: must equal
oracle/ref_cpu.denoiser_mlp_forward and its autograd on the same parameters)."""
import torch


def chain_restatement(x0, eps, t, sab, s1m, e, W, bias, gamma, beta, bf):
    """x0 / eps [B, T, D]; t [B]; sab / s1m the schedule tables; e [B, L * H] time-embedding rows; W[i] / bias[i] the blocks'
    Linear parameters then the head's; gamma / beta the LayerNorm parameters.  Returns (x_t [M, D], the per-block rounded
    pre-activations u_i (retain_grad), the per-block outputs h_i, pred (retain_grad), loss (already backward()ed: the
    .grad of u_i / pred / gamma / beta leaves are populated), gamma leaves, beta leaves)."""
    B, T, D = x0.shape
    L = len(W) - 1
    H = W[0].shape[0]
    M = B * T
    a = sab.to(torch.float64)[t][:, None, None]
    s = s1m.to(torch.float64)[t][:, None, None]
    xt_ref = bf(a * x0.to(torch.float64) + s * eps.to(torch.float64)).reshape(M, D)
    hcur = xt_ref.clone().requires_grad_(True)
    us, hs = [], []
    g64 = [g.to(torch.float64).requires_grad_(True) for g in gamma]
    b64 = [b.to(torch.float64).requires_grad_(True) for b in beta]
    for i in range(L):
        z = hcur @ W[i].to(torch.float64).T + bias[i].to(torch.float64) \
            + e.to(torch.float64)[:, i * H:(i + 1) * H].repeat_interleave(T, dim=0)
        # value rounded to bf16, gradient passes straight through
        u = z + (bf(z) - z).detach()
        u.retain_grad()
        v = u * torch.sigmoid(u)
        mu = v.mean(-1, keepdim=True)
        var = ((v - mu) ** 2).mean(-1, keepdim=True)
        hh = (v - mu) / torch.sqrt(var + 1e-5) * g64[i] + b64[i]
        hh = hh + (bf(hh) - hh).detach()
        us.append(u)
        hs.append(hh)
        hcur = hh
    pred = hcur @ W[L].to(torch.float64).T + bias[L].to(torch.float64)
    pred = pred + (bf(pred) - pred).detach()
    pred.retain_grad()
    loss = ((pred - eps.to(torch.float64).reshape(M, D)) ** 2).mean()
    loss.backward()
    return xt_ref, us, hs, pred, loss, g64, b64
