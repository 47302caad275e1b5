"""Per-launch breakdown of one DDIM denoise step (BASELINE config 5): every distinct C-ABI call of an eager step is
re-timed back-to-back inside a hipGraph.  Usage (GPU box): python tools/sampler_prof.py [B T]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd import hip  # noqa: E402
from inferbiomechanics_amd.diffusion.sampler import DDIMSampler  # noqa: E402


def main():
    B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 200)
    dev = torch.device("cuda", 0)
    model = bench.build_model("transformer", T, 300, torch.bfloat16, dev)
    sampler = DDIMSampler(model, 100, use_graph=False)
    xT = torch.randn(B, T, 300, device=dev)
    sampler.sample(xT, steps=2)
    torch.cuda.synchronize()
    with hip.record_launches() as rec:
        sampler.sample(xT, steps=1)
        torch.cuda.synchronize()
    uniq = {}
    for name, args in rec.calls:
        key = (name, bench._ints(args))
        uniq.setdefault(key, [args, 0])[1] += 1
    rows = []
    for (name, ints), (args, count) in uniq.items():
        if name in hip._RecordingLib.SKIP:
            continue
        us = hip.time_recorded_call(name, args)
        rows.append((us * count, name, count, us, ints[-5:]))
    rows.sort(reverse=True)
    print(f"{len(rec.calls)} launches per denoise step, sum of kernels {sum(r[0] for r in rows):.1f} us")
    for tot, name, count, us, ints in rows:
        print(f"{tot:8.1f} us  {count:2d} x {us:7.2f}  {name[3:]:18s} {list(ints)}")


if __name__ == "__main__":
    main()
