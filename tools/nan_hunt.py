"""Step the bench's transformer trainer and look at the loss / parameters after EVERY step (first non-finite value, which
parameters hold it).  Usage (GPU box): python tools/nan_hunt.py [steps]"""
import math
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build_model("transformer", 50, 300, torch.bfloat16, dev)
    tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4, use_graph=True)
    batches = bench.make_batches(8, 256, 50, 300, torch.bfloat16, dev, seed=0)
    prev = tr.adopt_stream()
    tr.pin_batches(batches)
    losses = []
    for i in range(steps):
        tr.step(batches[i % len(batches)])
        torch.cuda.synchronize()
        v = tr.loss_value()
        losses.append(v)
        bad_p = not bool(torch.isfinite(tr.flat).all())
        if not math.isfinite(v) or bad_p:
            print(f"step {i}: loss {v}, parameters finite: {not bad_p}, captured: {tr._rec is not None}", flush=True)
            for name, (lo, n_) in tr.layout.items():
                hi = lo + n_
                seg = tr.flat[lo:hi]
                if not bool(torch.isfinite(seg).all()):
                    nb = int((~torch.isfinite(seg)).sum())
                    print(f"   {name}: {nb} of {hi - lo} non-finite (first at {int((~torch.isfinite(seg)).nonzero()[0])})")
            gr = tr.grad
            for name, (lo, n_) in tr.layout.items():
                hi = lo + n_
                seg = gr[lo:hi]
                if not bool(torch.isfinite(seg).all()):
                    nb = int((~torch.isfinite(seg)).sum())
                    print(f"   grad {name}: {nb} of {hi - lo} non-finite (first at {int((~torch.isfinite(seg)).nonzero()[0])})")
            return 1
    print("all finite;", [round(x, 5) for x in losses[:4]], "...", round(losses[-1], 6), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
