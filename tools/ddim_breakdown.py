#!/usr/bin/env python3
"""Per-launch device time of ONE captured DDIM denoise step (transformer T = 200): every C-ABI call of an eager step is
recorded, each distinct call re-issued back-to-back inside a hipGraph and timed (bench.roofline_leg's method).
usage: python tools/ddim_breakdown.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    dev = torch.device("cuda", 0)
    for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
        model = bench.build_model("transformer", 200, 300, torch.bfloat16, dev)
        s = DDIMSampler(model, 100, use_graph=False)
        xT = torch.randn(B, 200, 300, device=dev)
        s.sample(xT, steps=2)
        torch.cuda.synchronize()
        s2 = DDIMSampler(model, 100, use_graph=False)
        with hip.record_launches() as rec:
            s2.sample(xT, steps=1)
            torch.cuda.synchronize()
        # the loop prologue (prepare_inference: table of time embeddings, frame projection) is not part of a step: keep
        # the calls after the last fill of the timestep vector
        names = [n for n, _ in rec.calls]
        start = max(i for i, n in enumerate(names) if n == "ib_fill_i64") + 1 if "ib_fill_i64" in names else 0
        rec.calls = rec.calls[start:]
        _, rows, total = bench.roofline_leg(rec, "bf16", gemm_family=True)
        n = sum(r["launches_per_step"] for r in rows)
        print(f"B = {B}: {n} launches (top 16 entries), {total:.1f} us of kernels per step")
        for r in rows:
            print(f"  {r['entry']:28s} x{r['launches_per_step']:2d} {r['avg_launch_us']:7.2f} us  {r['us_per_step']:8.2f}  {r['dims'][-6:]}  {r['tflops']}")


if __name__ == "__main__":
    main()
