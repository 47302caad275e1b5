#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the C-ABI entry points at the benchmark shapes: back-to-back launches
between two HIP events on the launch stream (host overhead hidden by queueing), reported as µs/launch,
algorithmic TFLOP/s (GEMMs) and algorithmic GB/s (row-wise kernels).  GPU only.

    python tools/kbench.py [--dtype bf16|f32] [--reps 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402


def timeit(fn, reps):
    """`reps` launches captured into ONE hipGraph (no host launch overhead between kernels), replayed
    3 times between two HIP events on the launch stream"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = hip.Graph()
        g.begin()
        for _ in range(reps):
            fn()
        g.end()
        g.launch()
        e0, e1 = hip.Event(), hip.Event()
        e0.record()
        for _ in range(3):
            g.launch()
        e1.record()
        ms = e0.elapsed_ms(e1)
    torch.cuda.synchronize()
    return ms * 1e3 / (3 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if a.dtype == "bf16" else 4
    dev = "cuda"
    hip.lib()
    rows = []

    def rec(name, shape, us, flops=None, bytes_=None):
        r = {"kernel": name, "shape": shape, "us": round(us, 2)}
        if flops:
            r["TFLOP/s"] = round(flops / us / 1e6, 1)
        if bytes_:
            r["GB/s"] = round(bytes_ / us / 1e3, 1)
        rows.append(r)
        print(json.dumps(r), flush=True)

    B, T = 256, 50
    M = B * T
    gemm_shapes = [(M, 512, 300), (M, 512, 512), (M, 300, 512), (256, 512, 128), (256, 1024, 512),
                   (M, 1536, 512), (M, 2048, 512), (M, 512, 2048)]
    for (m, n, k) in gemm_shapes:
        if a.only and a.only not in "gemm":
            continue
        x = torch.randn(m, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt)
        b = torch.randn(n, device=dev)
        y = torch.empty(m, n, device=dev, dtype=dt)
        z = torch.empty(m, n, device=dev, dtype=dt)
        fl = 2 * m * n * k
        rec("linear_fwd", [m, n, k], timeit(lambda: hip.linear_fwd(x, w, b, y), a.reps), fl, (m * k + n * k + m * n) * es)
        rec("linear_fwd+silu+z", [m, n, k], timeit(lambda: hip.linear_fwd(x, w, b, y, act="silu", z=z), a.reps), fl)
        dz = torch.randn(m, n, device=dev).to(dt)
        dx = torch.empty(m, k, device=dev, dtype=dt)
        rec("linear_dgrad", [m, n, k], timeit(lambda: hip.linear_dgrad(dz, w, dx), a.reps), fl, (m * k + n * k + m * n) * es)
        rec("linear_dgrad+silu", [m, n, k], timeit(lambda: hip.linear_dgrad(dz, w, dx, act_below="silu", aux=x), a.reps), fl)
        dw = torch.empty(n, k, device=dev)
        ws = torch.empty(max(hip.linear_wgrad_workspace_bytes(m, n, k), 16), dtype=torch.uint8, device=dev)
        rec("linear_wgrad", [m, n, k], timeit(lambda: hip.linear_wgrad(dz, x, dw, ws), a.reps), fl, (m * k + m * n) * es + n * k * 4)

    for N in (512,):
        x = torch.randn(M, N, device=dev).to(dt)
        g = torch.ones(N, device=dev)
        bb = torch.zeros(N, device=dev)
        y = torch.empty_like(x)
        mu = torch.empty(M, device=dev)
        rs = torch.empty(M, device=dev)
        rec("layernorm_fwd(silu)", [M, N], timeit(lambda: hip.layernorm_fwd(x, g, bb, y, mu, rs, act="silu"), a.reps),
            bytes_=2 * M * N * es)
        rec("layernorm_fwd(res)", [M, N], timeit(lambda: hip.layernorm_fwd(x, g, bb, y, mu, rs, res=x), a.reps),
            bytes_=3 * M * N * es)
        hip.layernorm_fwd(x, g, bb, y, mu, rs, act="silu")
        dx = torch.empty_like(x)
        dg = torch.empty(N, device=dev)
        db = torch.empty(N, device=dev)
        ws = torch.empty(hip.layernorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=dev)
        rec("layernorm_bwd(silu)", [M, N],
            timeit(lambda: hip.layernorm_bwd(y, x, g, mu, rs, dx, dg, db, ws, act="silu"), a.reps), bytes_=3 * M * N * es)
        out = torch.empty(B, N, device=dev)
        rec("segment_colsum(T)", [M, N], timeit(lambda: hip.segment_colsum(x, out, seg=T), a.reps), bytes_=M * N * es)
        o1 = torch.empty(1, N, device=dev)
        rec("segment_colsum(all of [256,N] f32)", [B, N], timeit(lambda: hip.segment_colsum(out, o1, seg=B), a.reps),
            bytes_=B * N * 4)

    D = 300
    x0 = torch.randn(B, T, D, device=dev).to(dt)
    eps = torch.randn(B, T, D, device=dev).to(dt)
    t = torch.randint(0, 1000, (B,), device=dev)
    from inferbiomechanics_amd.diffusion.schedule import DiffusionTables
    tabs = DiffusionTables(torch.device(dev))
    xt = torch.empty_like(x0)
    rec("q_sample", [B, T, D], timeit(lambda: hip.q_sample(x0, eps, t, tabs.sqrt_ab, tabs.sqrt_1mab, xt), a.reps),
        bytes_=3 * x0.numel() * es)
    res = torch.zeros(1, device=dev)
    dp = torch.empty_like(x0)
    ws = torch.empty(hip.mse_loss_workspace_bytes(x0.numel()), dtype=torch.uint8, device=dev)
    rec("mse_loss", [B, T, D], timeit(lambda: hip.mse_loss(x0, eps, res, ws, dpred=dp), a.reps), bytes_=3 * x0.numel() * es)
    n = 1_200_000
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev)
    s1 = torch.zeros(n, device=dev)
    sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
    rec("optim_step(rmsprop)", [n], timeit(lambda: hip.optim_step("rmsprop", p, g, s1, None, 1e-4, step=1, shadow=sh), a.reps),
        bytes_=n * (4 * 5 + 2))
    for (b_, t_, h_, dh_) in ((256, 50, 8, 64), (64, 200, 8, 64)):
        d = h_ * dh_
        qkv = torch.randn(b_, t_, 3 * d, device=dev).to(dt)
        o = torch.empty(b_, t_, d, device=dev, dtype=dt)
        lse = torch.empty(b_, h_, t_, device=dev)
        fl = 4 * b_ * h_ * t_ * t_ * dh_
        rec("attention_fwd", [b_, t_, h_, dh_], timeit(lambda: hip.attention_fwd(qkv, o, lse, h_), max(5, a.reps // 5)), fl)
        dq = torch.empty_like(qkv)
        rec("attention_bwd", [b_, t_, h_, dh_], timeit(lambda: hip.attention_bwd(qkv, o, o, lse, dq, h_), max(5, a.reps // 5)),
            2.5 * fl)
    print("TABLE " + json.dumps(rows))


if __name__ == "__main__":
    main()
