"""Per-phase timing of the fused MLP chain kernel (csrc/chain.hip): workgroup thread 0 stamps s_memrealtime
(100 MHz) at every phase boundary into a debug buffer; prints the mean / max per phase over all workgroups.
Usage (GPU box): python tools/chain_prof.py [B T D H L]"""
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402
import ctypes  # noqa: E402


def main():
    B, T, D, H, L = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (256, 50, 300, 512, 2)
    dev, bf = "cuda", torch.bfloat16
    M = B * T
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(B, T, D, generator=g).to(dev, bf)
    eps = torch.randn(B, T, D, generator=g).to(dev, bf)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    sab = torch.rand(1000, generator=g).to(dev)
    s1m = torch.rand(1000, generator=g).to(dev)
    e = (torch.randn(B, L * H, generator=g) * 0.5).to(dev, bf)
    dims = [D] + [H] * L
    W = [(torch.randn(dims[i + 1], dims[i], generator=g) * dims[i] ** -0.5).to(dev, bf) for i in range(L)]
    W.append((torch.randn(D, H, generator=g) * H ** -0.5).to(dev, bf))
    bias = [torch.zeros(H, device=dev) for _ in range(L)] + [torch.zeros(D, device=dev)]
    gamma = [torch.ones(H, device=dev) for _ in range(L)]
    beta = [torch.zeros(H, device=dev) for _ in range(L)]
    packed = torch.zeros(hip.mlp_chain_packed_elems(D, H, L), dtype=bf, device=dev)
    hip.mlp_chain_pack(W, packed, D, H)
    Dp = (D + 7) // 8 * 8
    xt = torch.zeros(M, Dp, dtype=bf, device=dev)[:, :D]
    dpred = torch.zeros(M, Dp, dtype=bf, device=dev)[:, :D]
    u = [torch.zeros(M, H, dtype=bf, device=dev) for _ in range(L)]
    h = [torch.zeros(M, H, dtype=bf, device=dev) for _ in range(L)]
    dz = [torch.zeros(M, H, dtype=bf, device=dev) for _ in range(L)]
    nwg = hip.mlp_chain_workgroups(M)
    part = torch.zeros(nwg, hip.mlp_chain_partial_width(D, H, L), device=dev)
    stamps = torch.zeros(nwg, 64, dtype=torch.int64, device=dev)

    def launch():
        hip.mlp_chain_train(x0, eps, t, sab, s1m, e, packed, bias, gamma, beta, xt, u, h, dz, dpred, part, T)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    # un-instrumented launch time: 200 back-to-back launches between two events, best of 5
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print(f"launch (no stamps, back to back): {best:.2f} us   lib={hip.LIB_PATH.split('/')[-1]}")
    if "--no-phases" in sys.argv:
        return
    hip.lib().ib_debug_set_chain_prof(ctypes.c_void_p(stamps.data_ptr()))
    import os
    names = ["q_sample"]
    if os.environ.get("IB_CHAIN_V1"):
        for i in range(L):
            names += [f"fwd{i}.gemm", f"fwd{i}.epi.sync+u+silu", f"fwd{i}.epi.reduce", f"fwd{i}.epi.copy_u+h+sync"]
        names += ["head.gemm", "head.epi"]
        for i in range(L - 1, -1, -1):
            names += [f"bwd{i}.gemm", f"bwd{i}.u_image+sync", f"bwd{i}.pass1a", f"bwd{i}.reduce_a", f"bwd{i}.pass2a+pass1b",
                      f"bwd{i}.reduce_b", f"bwd{i}.pass2b", f"bwd{i}.colsums", f"bwd{i}.sync(+copy_dz0)"]
    else:                                # v2: row-wise epilogues
        for i in range(L):
            names += [f"fwd{i}.gemm", f"fwd{i}.sync+u_exchange+sync", f"fwd{i}.rows(silu,LN,h)+sync"]
        names += ["head.gemm", "head.epi+dpred_rows"]
        for i in range(L - 1, -1, -1):
            names += [f"bwd{i}.gemm", f"bwd{i}.dh_half0+sync", f"bwd{i}.rows_half0+sync", f"bwd{i}.dh_half1+sync",
                      f"bwd{i}.rows_half1+sync", f"bwd{i}.colsums"]
    n = len(names)
    acc = None
    reps = 8
    for _ in range(reps):
        for _ in range(40):                 # back-to-back launches: warm clocks, steady state; the last one is read
            launch()
        torch.cuda.synchronize()
        s = stamps.cpu().double()
        d = (s[:, 1:n + 1] - s[:, :n]) * 0.01            # us (100 MHz)
        tot = (s[:, n] - s[:, 0]) * 0.01
        span = (s[:, n].max() - s[:, 0].min()) * 0.01
        row = torch.cat([d.mean(0), tot.mean().view(1), tot.max().view(1), span.view(1)])
        acc = row if acc is None else acc + row
    hip.lib().ib_debug_set_chain_prof(None)
    acc /= reps
    for nm, v in zip(names, acc[:n].tolist()):
        print(f"{nm:24s} {v:7.2f} us")
    print(f"per-WG total mean {acc[-3]:.2f} us, max {acc[-2]:.2f} us, first-start -> last-end {acc[-1]:.2f} us  ({nwg} workgroups)")


if __name__ == "__main__":
    main()
