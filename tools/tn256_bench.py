"""The transformer layer's grouped weight-gradient launch (M = 12800; [1536,512], [512,512], [2048,512], [512,2048], with
bias partial sums): 256 x 256 kernel (gemm_tn256.hip, one split count) against the 256 x 128 kernel (IB_NO_TN256=1 in a
second process).  Usage (GPU box): python tools/tn256_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401  (the -DIB_AB measurement build: A/B switches + stamp hooks)
from inferbiomechanics_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    dev, M = "cuda", 12800
    g = torch.Generator().manual_seed(0)
    probs, parts = [], []
    for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
        dz = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
        x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=dev)
        probs.append((dz, x, ws))
        parts.append(torch.zeros(32, N, device=dev))
    hip.lib().ib_debug_last_path()
    ns = hip.linear_wgrad_slabs_multi(probs, bias_parts=parts)
    path = hip.PATH_NAMES[int(hip.lib().ib_debug_last_path())]
    us = timeit(lambda: hip.linear_wgrad_slabs_multi(probs, bias_parts=parts), 20)
    flops = 2 * M * sum(p[0].shape[1] * p[1].shape[1] for p in probs)
    print(f"{path}: slabs {ns}, {us:.1f} us, {flops / us / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
