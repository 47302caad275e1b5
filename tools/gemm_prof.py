"""Per-phase timing of the ring GEMM kernel (csrc/gemm.hip): thread 0 of every workgroup stamps the 100 MHz wall clock
at: start, prologue loads issued, first stage landed (+ barrier), main loop done, epilogue done.
Usage (GPU box): python tools/gemm_prof.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402


def run(kind, M, N, K):
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    if kind == "fwd":
        x = torch.randn(M, K, generator=g).to(dev, bf); w = torch.randn(N, K, generator=g).to(dev, bf)
        y = torch.zeros(M, N, device=dev, dtype=bf); b = torch.zeros(N, device=dev)
        launch = lambda: hip.linear_fwd(x, w, b, y)
    elif kind == "dgrad":
        dz = torch.randn(M, N, generator=g).to(dev, bf); w = torch.randn(N, K, generator=g).to(dev, bf)
        dx = torch.zeros(M, K, device=dev, dtype=bf)
        launch = lambda: hip.linear_dgrad(dz, w, dx)
    else:
        dz = torch.randn(M, N, generator=g).to(dev, bf); x = torch.randn(M, K, generator=g).to(dev, bf)
        ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=dev)
        launch = lambda: hip.linear_wgrad_slabs(dz, x, ws)
    nwg = 4096
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    hip.lib().ib_debug_set_gemm_prof(ctypes.c_void_p(stamps.data_ptr()))
    acc = None
    for _ in range(8):
        stamps.zero_()
        for _ in range(30):
            launch()
        torch.cuda.synchronize()
        s = stamps.cpu().double()
        s = s[s[:, 4] > 0]
        d = (s[:, 1:5] - s[:, 0:4]) * 0.01
        row = torch.cat([d.mean(0), ((s[:, 4] - s[:, 0]) * 0.01).mean().view(1),
                         ((s[:, 4].max() - s[:, 0].min()) * 0.01).view(1), torch.tensor([float(s.shape[0])])])
        acc = row if acc is None else acc + row
    hip.lib().ib_debug_set_gemm_prof(None)
    acc /= 8
    print(f"{kind:6s} [{M},{N},{K}]  issue {acc[0]:.2f}  first-stage {acc[1]:.2f}  main loop {acc[2]:.2f}  epilogue {acc[3]:.2f}"
          f"  | per-WG {acc[4]:.2f} us, kernel span {acc[5]:.2f} us, {int(acc[6])} workgroups")


if __name__ == "__main__":
    for kind, M, N, K in [("fwd", 12800, 512, 512), ("dgrad", 12800, 512, 512), ("wgrad", 12800, 512, 512),
                          ("fwd", 12800, 2048, 512), ("fwd", 12800, 512, 2048), ("fwd", 12800, 1536, 512),
                          ("dgrad", 12800, 512, 2048), ("dgrad", 12800, 2048, 512), ("wgrad", 12800, 2048, 512)]:
        run(kind, M, N, K)
