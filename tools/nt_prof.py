#!/usr/bin/env python3
"""Per-tile phase timing of the NT GEMM (csrc/gemm_nt.hip): thread 0 of every persistent workgroup stamps the 100 MHz
wall clock at kernel start, at the end of each tile's K loop and at the end of each tile's epilogue.
Usage (GPU box): python tools/nt_prof.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402


def run(M, N, K, act="none"):
    dev, bf = "cuda", torch.bfloat16
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=bf)
    launch = lambda: hip.linear_fwd(x, w, b, y, act=act)
    stamps = torch.zeros(256, 16, dtype=torch.int64, device=dev)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    hip.lib().ib_debug_set_nt_prof(ctypes.c_void_p(stamps.data_ptr()))
    rows = []
    for _ in range(10):
        stamps.zero_()
        launch()
        torch.cuda.synchronize()
        s = stamps.cpu().double() * 0.01
        t0 = s[:, 0].min()
        rows.append(s - t0)
    hip.lib().ib_debug_set_nt_prof(None)
    s = torch.stack(rows).median(0).values            # [256, 16]
    line = f"fwd [{M},{N},{K}] "
    for r in range(5):
        ke, ee = s[:, 1 + 2 * r], s[:, 2 + 2 * r]
        live = ke > 0
        if not live.any():
            break
        start = s[:, 0] if r == 0 else s[:, 2 * r]
        line += (f"| tile {r}: {int(live.sum())} WGs, K loop {float((ke - start)[live].mean()):.2f} us, "
                 f"epilogue {float((ee - ke)[live].mean()):.2f} us ")
    line += f"| first start spread {float(s[:, 0].max()):.2f} us, last end {float(s[:, 1:].max()):.2f} us"
    print(line, flush=True)


if __name__ == "__main__":
    for (m, n, k, a) in [(12800, 2048, 512, "relu"), (12800, 512, 2048, "none"), (12800, 1536, 512, "none"), (12800, 512, 512, "none")]:
        run(m, n, k, a)
