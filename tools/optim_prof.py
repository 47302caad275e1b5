"""Optimizer launch in isolation: ib_optim_step / ib_optim_step_sources on a flat buffer of the MLP denoiser's size
(1.16 M parameters) and the transformer's (13 M), back to back inside one hipGraph (tools/kbench.timeit).
Arms: plain / + bf16 shadow / + exit ticket / + 3 slab sources of 10 slabs (the MLP step's shape).  GPU only.
    python tools/optim_prof.py [opt]"""
import sys

import torch

sys.path.insert(0, ".")
from inferbiomechanics_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    opt = sys.argv[1] if len(sys.argv) > 1 else "rmsprop"
    dev = "cuda"
    print("lib", hip.LIB_PATH.split("/")[-1], "opt", opt)
    for n in (1_163_264, 13_000_000 // 4 * 4):
        p = torch.randn(n, device=dev)
        g = torch.randn(n, device=dev) * 1e-3
        s1 = torch.zeros(n, device=dev)
        s2 = torch.zeros(n, device=dev) if opt in ("adam", "adadelta", "adamax") else None
        sh = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        sd = torch.zeros(1, dtype=torch.int32, device=dev)
        tk = torch.zeros(hip.optim_ticket_words(), dtype=torch.int32, device=dev)
        arms = {
            "plain": dict(step=3),
            "shadow": dict(step=3, shadow=sh),
            "shadow+step_dev": dict(step=0, step_dev=sd, shadow=sh),
            "shadow+ticket": dict(step=0, step_dev=sd, ticket=tk, shadow=sh),
        }
        for name, kw in arms.items():
            us = timeit(lambda: hip.optim_step(opt, p, g, s1, s2, 1e-4, **kw), 50)
            byts = n * (4 * 2 + 4 + 8 + (8 if s2 is not None else 0) + (2 if "shadow" in kw else 0))
            print(f"n={n:9d} {name:18s} {us:7.2f} us  {byts / us / 1e3:7.0f} GB/s")
        if n < 2_000_000:
            # the MLP step's sources: W0 [512 x 300(304)], W1 [512 x 512], head [300 x 512], 10 slabs each
            items, off = [], 0
            for rows, cols in ((512, 304), (512, 512), (300, 512)):
                m = rows * cols
                ws = torch.randn(10, m, device=dev) * 1e-4
                items.append((ws, 10, g[off:off + m]))
                off += m
            part = torch.randn(200, 4096, device=dev)
            segs = [(0, 512, g[off:off + 512], None, 1.0), (512, 512, g[off + 512:off + 1024], None, 1.0)]
            us = timeit(lambda: hip.optim_step(opt, p, g, s1, s2, 1e-4, step=0, step_dev=sd, ticket=tk, shadow=sh,
                                               sources=(items, part, 200, segs)), 50)
            print(f"n={n:9d} {'sources(3x10 slabs)':18s} {us:7.2f} us")


def transformer_sources(opt="rmsprop"):
    """the transformer denoiser's optimizer launch: 13.26 M parameters, per layer 4 slab-backed weights (2 / 8 / 2 / 2
    slabs) and 8 column-sum ranges over 100 partial rows -- 48 sources"""
    dev = "cuda"
    n = 13_263_168
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev) * 1e-3
    s1 = torch.zeros(n, device=dev)
    sh = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    sd = torch.zeros(1, dtype=torch.int32, device=dev)
    tk = torch.zeros(hip.optim_ticket_words(), dtype=torch.int32, device=dev)
    items, segs, off = [], [], 100_000
    part = torch.randn(100, 8192, device=dev)
    for layer in range(4):
        for (rows, cols, ns) in ((1536, 512, 2), (512, 512, 8), (2048, 512, 2), (512, 2048, 2)):
            m = rows * cols
            items.append((torch.randn(ns, m, device=dev) * 1e-4, ns, g[off:off + m]))
            off += m
        c0 = 0
        for w in (1536, 512, 2048, 512, 512, 512, 512, 512):
            segs.append((c0, w, g[off:off + w], None, 1.0))
            off += w
            c0 += w
    base = timeit(lambda: hip.optim_step(opt, p, g, s1, None, 1e-4, step=0, step_dev=sd, ticket=tk, shadow=sh), 20)
    us = timeit(lambda: hip.optim_step(opt, p, g, s1, None, 1e-4, step=0, step_dev=sd, ticket=tk, shadow=sh,
                                       sources=(items, part, 100, segs)), 20)
    slab_mb = sum(it[0].numel() * 4 for it in items) / 1e6
    print(f"transformer-shaped: plain {base:.1f} us, with {len(items) + len(segs)} sources ({slab_mb:.0f} MB of slabs) {us:.1f} us")


if __name__ == "__main__":
    if "--transformer" in sys.argv:
        sys.argv.remove("--transformer")
        transformer_sources()
        sys.exit(0)
    main()
