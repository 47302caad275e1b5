"""Every distinct launch of ONE training step of a bench workload, re-timed back to back (bench.roofline_leg lists the top
16 only).  Usage (GPU box): python tools/step_entries.py [mlp_denoiser_T50 | transformer_denoiser_T50]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd import hip  # noqa: E402
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "transformer_denoiser_T50"
    kind, T, D, B = bench.WORKLOADS[workload][:4]
    dev = torch.device("cuda", 0)
    model = bench.build_model(kind, T, D, torch.bfloat16, dev)
    trainer = HipTrainer(model, "diffusion", "rmsprop", 1e-4)
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randn(B, T, D, generator=g).to(dev, torch.bfloat16), torch.randint(0, 1000, (B,), generator=g).to(dev),
                torch.randn(B, T, D, generator=g).to(dev, torch.bfloat16))]
    for _ in range(3):
        trainer.step(batches[0])
    rec = bench.record_eager_step(trainer, batches)
    uniq = {}
    for i, (name, args) in enumerate(rec.calls):
        if name in hip._RecordingLib.SKIP:
            continue
        key = (name, bench._ints(args))
        uniq.setdefault(key, [args, 0])[1] += 1
    rows = []
    for (name, ints), (args, count) in uniq.items():
        us = hip.time_recorded_call(name, args)
        rows.append((us * count, name, count, us, ints[-6:]))
    rows.sort(reverse=True)
    print(f"{len(rec.calls)} calls per step, sum of kernels {sum(r[0] for r in rows):.1f} us")
    for tot, name, count, us, ints in rows:
        print(f"{tot:8.1f} us  {count:2d} x {us:7.2f}  {name[3:]:30s} {list(ints)}")


if __name__ == "__main__":
    main()
