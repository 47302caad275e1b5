#!/usr/bin/env python3
"""Forward / dgrad Linear GEMMs at the transformer denoiser's training shapes: the 256 x 128 NT kernel (csrc/gemm_nt.hip)
against the 128 x 128 ring kernel of gemm.hip.  The switch is read once per process (IB_NO_NT), so run it twice:

    python tools/nt_bench.py ; IB_NO_NT=1 python tools/nt_bench.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    hip.lib()
    dev, dt = "cuda", torch.bfloat16
    tag = "ring128" if os.environ.get("IB_NO_NT") else "nt256"
    M = 12800
    rows = []
    for (n, k, act) in [(2048, 512, "relu"), (1536, 512, "none"), (512, 2048, "none"), (512, 512, "none")]:
        x = torch.randn(M, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt)
        b = torch.randn(n, device=dev)
        y = torch.empty(M, n, device=dev, dtype=dt)
        us = timeit(lambda: hip.linear_fwd(x, w, b, y, act=act), 20)
        rows.append({"kernel": tag, "op": "fwd", "MNK": [M, n, k], "us": round(us, 2), "TFLOP/s": round(2 * M * n * k / us / 1e6, 1)})
        print(json.dumps(rows[-1]), flush=True)
        # dgrad of the same layer: dx[M,k] = dz[M,n] w[n,k]
        dz = torch.randn(M, n, device=dev).to(dt)
        dx = torch.empty(M, k, device=dev, dtype=dt)
        wt = torch.empty(k, n, device=dev, dtype=dt)
        hip.transpose_multi([(w, wt)])
        add = torch.randn(M, k, device=dev).to(dt)

        def dgrad():
            if os.environ.get("IB_NO_NT") or not hip.linear_dgrad_wt(dz, wt, dx, addend=add):
                hip.linear_dgrad(dz, w, dx, addend=add)
        us = timeit(dgrad, 20)
        rows.append({"kernel": tag, "op": "dgrad+addend", "MNK": [M, k, n], "us": round(us, 2),
                     "TFLOP/s": round(2 * M * n * k / us / 1e6, 1)})
        print(json.dumps(rows[-1]), flush=True)
    # the relu'-gated dgrad of FFN2 (output 2048 wide)
    dz = torch.randn(M, 512, device=dev).to(dt)
    w = (torch.randn(512, 2048, device=dev) / 45).to(dt)
    wt = torch.empty(2048, 512, device=dev, dtype=dt)
    hip.transpose_multi([(w, wt)])
    aux = torch.randn(M, 2048, device=dev).to(dt)
    dx = torch.empty(M, 2048, device=dev, dtype=dt)

    def dgrad2():
        if os.environ.get("IB_NO_NT") or not hip.linear_dgrad_wt(dz, wt, dx, act_below="relu", aux=aux):
            hip.linear_dgrad(dz, w, dx, act_below="relu", aux=aux)
    us = timeit(dgrad2, 20)
    print(json.dumps({"kernel": tag, "op": "dgrad*relu'", "MNK": [M, 2048, 512], "us": round(us, 2),
                      "TFLOP/s": round(2 * M * 2048 * 512 / us / 1e6, 1)}), flush=True)
    for name, shapes in (("transformer layer", [(512, 2048), (2048, 512), (512, 512), (1536, 512)]),
                         ("mlp denoiser", [(304, 512), (512, 512), (512, 304)])):
        probs = []
        for (n, k) in shapes:
            dz = torch.randn(M, n, device=dev).to(dt)
            x = torch.randn(M, k, device=dev).to(dt)
            wsb = torch.empty(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, n, k)), dtype=torch.uint8, device=dev)
            probs.append((dz, x, wsb))
        parts = [torch.empty(32, n, device=dev) for (n, k) in shapes]
        us = timeit(lambda: hip.linear_wgrad_slabs_multi(probs, bias_parts=parts), 10)
        fl = sum(2 * M * n * k for n, k in shapes)
        print(json.dumps({"kernel": "tn256" if not os.environ.get("IB_NO_TN") else "ring128", "op": "wgrad group + bias: " + name,
                          "us": round(us, 2), "TFLOP/s": round(fl / us / 1e6, 1)}), flush=True)
    ws = [(torch.randn(r, c, device=dev).to(dt), torch.empty(c, r, device=dev, dtype=dt))
          for (r, c) in [(512, 2048), (2048, 512), (512, 512), (1536, 512)] * 4]
    us = timeit(lambda: hip.transpose_multi(ws), 20)
    print(json.dumps({"kernel": "transpose_multi", "op": "16 weight matrices (4 layers)", "us": round(us, 2)}), flush=True)


if __name__ == "__main__":
    main()
