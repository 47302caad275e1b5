"""The fused feed-forward sublayer (csrc/ffn_chain.hip) against the per-op launches it replaces, at the transformer
denoiser's shape (M = 12800 token rows, d = 512, ffn = 2048, bf16): forward = Linear + ReLU, Linear, residual + LayerNorm;
backward = LayerNorm backward, two dgrad GEMMs.  Back-to-back launches between two events.  Usage: python tools/ffn_bench.py [M]"""
import sys

import torch

sys.path.insert(0, ".")
from inferbiomechanics_amd import hip  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
    d, ffn, dev, bf = 512, 2048, "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    q = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, bf)
    x1, w1, w2, dy = q(M, d), q(ffn, d, sc=d ** -0.5), q(d, ffn, sc=ffn ** -0.5), q(M, d)
    b1, b2 = torch.zeros(ffn, device=dev), torch.zeros(d, device=dev)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=dev)
    f1, dz1 = torch.empty(M, ffn, dtype=bf, device=dev), torch.empty(M, ffn, dtype=bf, device=dev)
    s2, y, ds2, dx1 = (torch.empty(M, d, dtype=bf, device=dev) for _ in range(4))
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device=dev)
    part = torch.zeros(2 * hip.ffn_chain_workgroups(M, d, ffn), d, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t_pack = timeit(lambda: hip.ffn_chain_pack([(w1, w2, packed)] * 4))
        t_fwd = timeit(lambda: hip.ffn_chain_fwd(x1, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask))
        t_bwd = timeit(lambda: hip.ffn_chain_bwd(dy, s2, mean, rstd, gamma, packed, mask, ds2, dz1, dx1, part))
        # the per-op launches
        f2 = torch.empty(M, d, dtype=bf, device=dev)
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
        lnws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, d), dtype=torch.uint8, device=dev)

        def fwd_ops():
            hip.linear_fwd(x1, w1, b1, f1, act="relu")
            hip.linear_fwd(f1, w2, b2, f2)
            hip.layernorm_fwd(f2, gamma, beta, y, mean, rstd, res=x1)

        def bwd_ops():
            hip.layernorm_bwd(dy, f2, gamma, mean, rstd, ds2, None, None, lnws, res=x1)
            if not hip.linear_dgrad_wt(ds2, w2t, dz1, act_below="relu", aux=f1):
                hip.linear_dgrad(ds2, w2, dz1, act_below="relu", aux=f1)
            if not hip.linear_dgrad_wt(dz1, w1t, dx1, addend=ds2):
                hip.linear_dgrad(dz1, w1, dx1, addend=ds2)
        t_fo, t_bo = timeit(fwd_ops), timeit(bwd_ops)
    fl = 2 * 2 * M * d * ffn
    print(f"M={M}: pack (4 layers) {t_pack:.1f} us | fused fwd {t_fwd:.1f} us ({fl / t_fwd / 1e6:.0f} TF/s) vs per-op {t_fo:.1f} us | "
          f"fused bwd {t_bwd:.1f} us ({fl / t_bwd / 1e6:.0f} TF/s) vs per-op {t_bo:.1f} us")


if __name__ == "__main__":
    main()
