#!/usr/bin/env python3
"""Cost of a kernel boundary inside a captured graph: N dependent launches of (a) a one-element cast, (b) the 12800 x 512
LayerNorm forward, replayed; prints us per node.  Run under different HIP runtime switches (one process each) to see which
of them move the boundary cost:  python tools/graph_gap.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    from inferbiomechanics_amd import hip
    dev = torch.device("cuda", 0)
    a = torch.zeros(64, device=dev)
    b = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    x = torch.randn(12800, 512, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    m, r = torch.empty(12800, device=dev), torch.empty(12800, device=dev)
    out = {}
    for name, fn in (("cast1", lambda: hip.cast(a, b)), ("layernorm", lambda: hip.layernorm_fwd(x, gam, bet, y, m, r))):
        fn(); torch.cuda.synchronize()
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps / n * 1e6
    # fork / join: N x [main kernel; fork a side stream with one kernel beside one main kernel; join]
    side = torch.cuda.Stream()
    ev_f, ev_j = [torch.cuda.Event() for _ in range(n)], [torch.cuda.Event() for _ in range(n)]
    a2, b2 = torch.zeros(64, device=dev), torch.zeros(64, dtype=torch.bfloat16, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                hip.cast(a, b)
                ev_f[i].record(s)
                with torch.cuda.stream(side):
                    side.wait_event(ev_f[i])
                    hip.cast(a2, b2)
                    ev_j[i].record(side)
                hip.cast(a, b)
                s.wait_event(ev_j[i])
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    out["fork_join(3 casts)"] = (time.perf_counter() - t0) / 20 / n * 1e6
    keys = ("HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "AMD_OPT_FLUSH", "DEBUG_HIP_GRAPH_BATCH_SIZE",
            "DEBUG_HIP_FORCE_GRAPH_QUEUES", "GPU_FLUSH_ON_EXECUTION", "DEBUG_CLR_KERNARG_HDP_FLUSH_WA")
    env = {k: os.environ[k] for k in keys if k in os.environ}
    print(f"{env}: " + ", ".join(f"{k} {v:.2f} us/node" for k, v in out.items()), flush=True)


if __name__ == "__main__":
    main()
