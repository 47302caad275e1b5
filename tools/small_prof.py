"""Back-to-back event timing of the small launches around the chain kernel (pack, time-MLP forward, both in one launch,
multi-segment column sums, slab reduction).  Usage (GPU box): python tools/small_prof.py"""
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402


def timeit(name, fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hg = hip.Graph()
        hg.begin()
        for _ in range(reps):
            fn()
        hg.end()
        hg.launch()
        torch.cuda.synchronize()
        e0.record(s)
        hg.launch()
        e1.record(s)
    torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) * 1000 / reps:7.2f} us / launch")


def main():
    dev, bf = "cuda", torch.bfloat16
    B, T, D, H, L = 256, 50, 300, 512, 2
    g = torch.Generator().manual_seed(0)
    dims = [D] + [H] * L
    W = [(torch.randn(dims[i + 1], dims[i], generator=g) * 0.05).to(dev, bf) for i in range(L)]
    W.append((torch.randn(D, H, generator=g) * 0.05).to(dev, bf))
    packed = torch.zeros(hip.mlp_chain_packed_elems(D, H, L), dtype=bf, device=dev)
    table = torch.randn(1000, 128, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    w1 = (torch.randn(512, 128, generator=g) * 0.05).to(dev, bf)
    w2 = (torch.randn(L * H, 512, generator=g) * 0.05).to(dev, bf)
    b1 = torch.zeros(512, device=dev); b2 = torch.zeros(L * H, device=dev)
    s = torch.zeros(B, 128, dtype=bf, device=dev); zu = torch.zeros(B, 512, dtype=bf, device=dev)
    u = torch.zeros(B, 512, dtype=bf, device=dev); e = torch.zeros(B, L * H, dtype=bf, device=dev)
    timeit("mlp_chain_pack", lambda: hip.mlp_chain_pack(W, packed, D, H))
    timeit("time_mlp_fwd", lambda: hip.time_mlp_fwd(table, t, w1, b1, w2, b2, s, zu, u, e))
    timeit("mlp_chain_prep (both)", lambda: hip.time_mlp_fwd(table, t, w1, b1, w2, b2, s, zu, u, e, pack=(W, packed, D, H)))
    import ctypes
    stamps = torch.zeros(1024, 8, dtype=torch.int64, device=dev)
    hip.lib().ib_debug_set_chain_prof(ctypes.c_void_p(stamps.data_ptr()))
    for _ in range(20):
        hip.time_mlp_fwd(table, t, w1, b1, w2, b2, s, zu, u, e)
    torch.cuda.synchronize()
    hip.lib().ib_debug_set_chain_prof(None)
    st = stamps.cpu().double()
    st = st[st[:, 0] > 0]                 # the time-MLP workgroups that stamped (16 or 64 windows each)
    print('time workgroups:', st.shape[0])
    d = (st[:, 1:5] - st[:, 0:4]) * 0.01
    print("time_mlp_fwd phases (us): gather", d[:, 0].mean().item(), "stage1", d[:, 1].mean().item(), "stage2 gemm",
          d[:, 2].mean().item(), "store", d[:, 3].mean().item(), "| per-WG", ((st[:, 4] - st[:, 0]) * 0.01).mean().item(),
          "span", ((st[:, 4].max() - st[:, 0].min()) * 0.01).item())
    if st[:, 5].max() > 0:               # -DTF_EXP build: stamps inside the gather (t arrived | weights arrived | table rows arrived)
        print("gather split (us): t", ((st[:, 5] - st[:, 0]) * 0.01).mean().item(), "weights", ((st[:, 6] - st[:, 5]) * 0.01).mean().item(),
              "table", ((st[:, 7] - st[:, 6]) * 0.01).mean().item(), "stores+barrier", ((st[:, 1] - st[:, 7]) * 0.01).mean().item())
    part = torch.randn(256, 3464, generator=g).to(dev)
    outs = [torch.zeros(512, device=dev) for _ in range(7)]
    segs = [(512 * i, 512, outs[i], None, 1.0) for i in range(6)] + [(3072, 300, outs[6], None, 1.0)]
    timeit("colsum_segments (7 segs)", lambda: hip.colsum_segments(part, 256, segs))


if __name__ == "__main__":
    main()
