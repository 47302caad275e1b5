"""In-kernel phase profile + back-to-back launch time of the fused token-local FORWARD launch in its three forms (plain,
+ attention epilogue, + attention epilogue and the next layer's QKV tail) -- csrc/ffn_chain.hip, measurement build.
Usage: python tools/ffn_prof2.py [M]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401
from inferbiomechanics_amd import hip  # noqa: E402


def timeit(fn, n=40):
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
    z = lambda *s: torch.zeros(*s, device=dev)
    x, attn = q(M, d), q(M, d)
    w1, w2, wo, wqkv = q(ffn, d, sc=d ** -0.5), q(d, ffn, sc=ffn ** -0.5), q(d, d, sc=d ** -0.5), q(3 * d, d, sc=d ** -0.5)
    b1, b2, bo, bq = z(ffn), z(d), z(d), z(3 * d)
    gamma, beta, g1, be1 = torch.ones(d, device=dev), z(d), torch.ones(d, device=dev), z(d)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=dev)
    hip.ffn_chain_pack([(w1, w2, packed, wo, wqkv)])
    e = lambda *s: torch.empty(*s, dtype=bf, device=dev)
    f1, s2, y, s1, x1o, qkv = e(M, ffn), e(M, d), e(M, d), e(M, d), e(M, d), e(M, 3 * d)
    mean, rstd, m1, r1 = z(M), z(M), z(M), z(M)
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device=dev)
    nwg = hip.ffn_chain_workgroups(M, d, ffn)
    ao = (attn, bo, g1, be1, s1, x1o, m1, r1)
    forms = {
        "plain": lambda: hip.ffn_chain_fwd(x, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask),
        "out": lambda: hip.ffn_chain_fwd(x, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask, attn_out=ao),
        "out+qkv": lambda: hip.ffn_chain_fwd(x, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask, attn_out=ao,
                                             qkv_next=(packed, bq, qkv)),
    }
    nc = ffn // 512
    names = ["rows in (+ out-projection, LayerNorm1)"]
    for c in range(nc):
        names += [f"c{c}.gemm1", f"c{c}.sync(skew)", f"c{c}.relu+mask+sync", f"c{c}.gemm2(+f1 rows)"]
    names += ["s2 exchange + LayerNorm2 rows"]
    tail = []
    for c in range(3):
        tail += [f"q{c}.gemm(+rows of q{c - 1})", f"q{c}.sync+bias+image"]
    tail += ["last qkv rows out"]
    stamps = torch.zeros(nwg, 64, dtype=torch.int64, device=dev)
    for form, launch in forms.items():
        us = timeit(launch)
        nm = names + (tail if form == "out+qkv" else [])
        n = len(nm)
        hip.lib().ib_debug_set_ffn_prof(ctypes.c_void_p(stamps.data_ptr()))
        acc = None
        for _ in range(6):
            for _ in range(20):
                launch()
            torch.cuda.synchronize()
            s = stamps.cpu().double()
            dd = (s[:, 1:n + 1] - s[:, :n]) * 0.01
            row = torch.cat([dd.mean(0), ((s[:, n] - s[:, 0]) * 0.01).mean().view(1),
                             ((s[:, n].max() - s[:, 0].min()) * 0.01).view(1)])
            acc = row if acc is None else acc + row
        hip.lib().ib_debug_set_ffn_prof(None)
        acc /= 6
        print(f"== {form}: {us:.1f} us per launch back to back; per-WG mean {acc[-2]:.1f} us, first start -> last end {acc[-1]:.1f} us")
        for k, v in zip(nm, acc[:n].tolist()):
            print(f"   {k:42s} {v:7.2f} us")


if __name__ == "__main__":
    main()
