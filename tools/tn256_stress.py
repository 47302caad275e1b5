"""Run-to-run stress of the grouped 256 x 256 weight-gradient launch (transformer layer group, random operands): every
launch must reproduce the first one bit for bit, alone and with a memory-heavy stream beside it.
Usage (GPU box): python tools/tn256_stress.py [launches]"""
import sys

import torch

sys.path.insert(0, ".")
import tools._ab  # noqa: E402,F401
from inferbiomechanics_amd import hip  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev, M = "cuda", 12800
    g = torch.Generator().manual_seed(0)
    probs, parts = [], []
    for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
        dz = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
        x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=dev)
        probs.append((dz, x, ws))
        parts.append(torch.zeros(32, N, device=dev))
    ns = hip.linear_wgrad_slabs_multi(probs, bias_parts=parts)
    torch.cuda.synchronize()
    first = [p[2].clone() for p in probs] + [q.clone() for q in parts]
    for j, (dz, x, ws) in enumerate(probs):
        N, K = dz.shape[1], x.shape[1]
        got = ws[:ns[j] * N * K * 4].view(torch.float32).view(ns[j], N, K).sum(0)
        ref = dz.float().t() @ x.float()
        print(j, "max err vs fp32 matmul", float((got - ref).abs().max()), "nan" if torch.isnan(got).any() else "")
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)
    bad = 0
    for mode in ("alone", "beside a copy stream"):
        for i in range(n):
            if mode != "alone":
                with torch.cuda.stream(side):
                    big2.copy_(big)
            for p in probs:
                p[2].fill_(0x7F)
            for q in parts:
                q.fill_(float("nan"))
            hip.linear_wgrad_slabs_multi(probs, bias_parts=parts)
            torch.cuda.synchronize()
            now = [p[2] for p in probs] + list(parts)
            for j, (a, b) in enumerate(zip(first, now)):
                if j < 4:
                    N, K = probs[j][0].shape[1], probs[j][1].shape[1]
                    a, b = a[:ns[j] * N * K * 4], b[:ns[j] * N * K * 4]
                else:
                    a, b = a[:ns[j - 4]], b[:ns[j - 4]]
                if not torch.equal(a.view(torch.uint8), b.view(torch.uint8)):
                    bad += 1
                    d = (a.view(torch.uint8) != b.view(torch.uint8)).nonzero()
                    print(f"{mode}: launch {i} tensor {j}: {d.numel()} bytes differ, first at {int(d[0])}", flush=True)
                    if bad > 12:
                        print("giving up")
                        return 1
        print(mode, "done; mismatching tensors so far:", bad, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
