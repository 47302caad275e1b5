"""In-kernel phase profile of the whole-layer launches with the attention inside (csrc/ffn_chain.hip / ffn_chain_bwd.hip,
measurement build): thread 0 of every workgroup stamps the 100 MHz wall clock at phase boundaries.
Usage: python tools/layer_prof.py [B] [T]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401
from inferbiomechanics_amd import hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    M, d, ffn, dev, bf = B * T, 512, 2048, "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    q = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, bf)
    v = lambda n: (0.1 * torch.randn(n, generator=g)).to(dev)
    x, attn, dy = q(M, d), q(M, d), q(M, d)
    w1, w2, wo, wq = q(ffn, d, sc=d ** -0.5), q(d, ffn, sc=ffn ** -0.5), q(d, d, sc=d ** -0.5), q(3 * d, d, sc=2 * d ** -0.5)
    b1, b2, bo, bq = v(ffn), v(d), v(d), v(3 * d)
    g1, be1, g2, be2 = 1 + v(d), v(d), 1 + v(d), v(d)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=dev)
    hip.ffn_chain_pack([(w1, w2, packed, wo, wq)])
    e = lambda *s: torch.empty(*s, dtype=bf, device=dev)
    f1, s2, y, s1, x1o, qkv, ao = e(M, ffn), e(M, d), e(M, d), e(M, d), e(M, d), e(M, 3 * d), e(M, d)
    lse = torch.empty(B, 8, T, device=dev)
    mean, rstd, mean1, rstd1 = (torch.empty(M, device=dev) for _ in range(4))
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn, T), dtype=torch.uint8, device=dev)
    nwg = hip.ffn_chain_workgroups(M, d, ffn, T)
    ds2, dz1, ds1, dqkv, dx = e(M, d), e(M, ffn), e(M, d), e(M, 3 * d), e(M, d)
    part = torch.empty(4 * nwg, d, device=dev)
    fwd = lambda: hip.ffn_chain_fwd(x, packed, b1, b2, g2, be2, f1, s2, y, mean, rstd, mask,
                                    attn_out=(attn, bo, g1, be1, s1, x1o, mean1, rstd1), qkv_next=(packed, bq, qkv),
                                    attn_next=(ao, lse, T), panel_T=T)
    bwd = lambda: hip.ffn_chain_bwd(dy, s2, mean, rstd, g2, packed, mask, ds2, dz1, None, part,
                                    attn_out=(s1, mean1, rstd1, g1, ds1, None), attn_bwd=(qkv, lse, dqkv, dx, T))
    nc = ffn // 512
    fnames = ["x, attn rows -> LDS; out-proj GEMM; LayerNorm1"]
    for c in range(nc):
        fnames += [f"c{c}.gemm1", f"c{c}.sync(skew)", f"c{c}.relu+mask+sync", f"c{c}.gemm2(+f1 rows)"]
    fnames += ["s2 exchange, LayerNorm2 rows", "Q gemm", "Q epilogue", "K gemm (+Q rows out)", "K epilogue + S^T + softmax",
               "V gemm (+K rows out)", "V epilogue", "P.V, attn + V rows out"]
    bnames = ["LayerNorm2 backward (+dgamma/dbeta)", "feed-forward dgrads (8 GEMM phases)", "dx1 -> image D, LayerNorm1 backward",
              "out-projection dgrad GEMM", "barrier", "dO -> SB, q/k/v loads, K -> SA", "attention phase 1 (dQ)",
              "attention phase 2 (dK, dV)", "barrier", "in-projection dgrad (3 GEMM phases) + ds1", "dx rows out"]
    for name, launch, names in (("forward", fwd, fnames), ("backward", bwd, bnames)):
        stamps = torch.zeros(nwg, 64, dtype=torch.int64, device=dev)
        for _ in range(20):
            fwd(); bwd()
        torch.cuda.synchronize()
        hip.lib().ib_debug_set_ffn_prof(ctypes.c_void_p(stamps.data_ptr()))
        n = len(names)
        acc = None
        for _ in range(8):
            for _ in range(30):
                launch()
            torch.cuda.synchronize()
            s = stamps.cpu().double()
            dd = (s[:, 1:n + 1] - s[:, :n]) * 0.01
            row = torch.cat([dd.mean(0), ((s[:, n] - s[:, 0]) * 0.01).mean().view(1), ((s[:, n].max() - s[:, 0].min()) * 0.01).view(1)])
            acc = row if acc is None else acc + row
        hip.lib().ib_debug_set_ffn_prof(None)
        acc /= 8
        print(f"---- {name} (B = {B}, T = {T}, {nwg} workgroups; wave 0's timeline)")
        for nm, val in zip(names, acc[:n].tolist()):
            print(f"{nm:52s} {val:7.2f} us")
        print(f"per-WG mean {acc[-2]:.2f} us, first start -> last end {acc[-1]:.2f} us")


if __name__ == "__main__":
    main()
