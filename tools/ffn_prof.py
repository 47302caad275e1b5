"""In-kernel phase profile of the fused feed-forward forward kernel (csrc/ffn_chain.hip; measurement build): thread 0 of
every workgroup stamps the 100 MHz wall clock at every phase boundary.  Usage: python tools/ffn_prof.py [M]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401
from inferbiomechanics_amd import hip  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
    d, ffn, dev, bf = 512, 2048, "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    q = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, bf)
    x1, w1, w2 = q(M, d), q(ffn, d, sc=d ** -0.5), q(d, ffn, sc=ffn ** -0.5)
    b1, b2 = torch.zeros(ffn, device=dev), torch.zeros(d, device=dev)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=bf, device=dev)
    hip.ffn_chain_pack([(w1, w2, packed)])
    f1 = torch.empty(M, ffn, dtype=bf, device=dev)
    s2, y = torch.empty(M, d, dtype=bf, device=dev), torch.empty(M, d, dtype=bf, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device=dev)
    nwg = hip.ffn_chain_workgroups(M, d, ffn)
    stamps = torch.zeros(nwg, 64, dtype=torch.int64, device=dev)
    launch = lambda: hip.ffn_chain_fwd(x1, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask)
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    hip.lib().ib_debug_set_ffn_prof(ctypes.c_void_p(stamps.data_ptr()))
    nc = ffn // 512
    names = ["x rows -> LDS"]
    for c in range(nc):
        names += [f"c{c}.gemm1", f"c{c}.sync(skew)", f"c{c}.relu+mask+sync", f"c{c}.gemm2(+f1 rows)"]
    names += ["final: sync, s2 exchange, LayerNorm rows"]
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
    for nm, v in zip(names, acc[:n].tolist()):
        print(f"{nm:42s} {v:7.2f} us")
    print(f"per-WG mean {acc[-2]:.2f} us, first start -> last end {acc[-1]:.2f} us ({nwg} workgroups)")


if __name__ == "__main__":
    main()
