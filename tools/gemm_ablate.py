#!/usr/bin/env python3
"""Timing-only ablation of the GEMM K-loop phases (results are wrong while a mask is set): which of
global loads / LDS stores / LDS reads+MFMA / epilogue bounds a K-step.  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dt = torch.bfloat16
dev = "cuda"
L = hip.lib()
for (m, n, k) in [(12800, 512, 512), (12800, 512, 300), (256, 1024, 512), (12800, 2048, 512)]:
    x = torch.randn(m, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt)
    b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev, dtype=dt)
    dz = torch.randn(m, n, device=dev).to(dt)
    dx = torch.empty(m, k, device=dev, dtype=dt)
    dw = torch.empty(n, k, device=dev)
    ws = torch.empty(max(hip.linear_wgrad_workspace_bytes(m, n, k), 16), dtype=torch.uint8, device=dev)
    for mask, label in [(0, "full"), (1, "no steady-state global loads"), (2, "no LDS stores"), (3, "no loads, no stores"),
                        (4, "no LDS reads / MFMA"), (8, "no epilogue"), (15, "nothing (launch + prologue)")]:
        L.ib_debug_set_ablate(mask)
        r = {"shape": [m, n, k], "ablate": label,
             "fwd_us": round(timeit(lambda: hip.linear_fwd(x, w, b, y), 20), 2),
             "dgrad_us": round(timeit(lambda: hip.linear_dgrad(dz, w, dx), 20), 2),
             "wgrad_us": round(timeit(lambda: hip.linear_wgrad(dz, x, dw, ws), 20), 2)}
        print(json.dumps(r), flush=True)
    L.ib_debug_set_ablate(0)
