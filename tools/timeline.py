"""Device-side timeline of one captured training step: every C-ABI launch bracketed by wall-clock stamp kernels
(hip._StampLib).  The stamps add ~2 launches per call, so absolute times are inflated; the ORDER, the overlaps and
the relative sizes are what to read.  Usage (GPU box): python tools/timeline.py [workload]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd import hip  # noqa: E402
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "mlp_denoiser_T50"
    only = tuple("ib_" + n for n in sys.argv[2].split(",")) if len(sys.argv) > 2 else None
    kind, T, D, B = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    import os
    if os.environ.get("IB_DDP_SELFTEST") == "1":     # the data-parallel launch sequence on one rank (RCCL, world = 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    model = bench.build_model(kind, T, D, torch.bfloat16, dev)
    batches = bench.make_batches(4, B, T, D, torch.bfloat16, dev, seed=0)
    with hip.stamp_launches(only=only) as sl:
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4, use_graph=True)
        for i in range(2):
            tr.step(batches[i])            # eager warm-ups
        sl.calls.clear()
        tr.step(batches[2])                # capture (+ first replay)
        ncap = len(sl.calls)
        for i in range(5):
            tr.step(batches[i % 4])        # replays
        tl = sl.timeline()[:ncap]
    streams = {}
    for name, ints, st, a, b in tl:
        streams.setdefault(st, len(streams))
    tl.sort(key=lambda r: r[3])
    end = max(r[4] for r in tl)
    print(f"{len(tl)} launches, {len(streams)} streams, first stamp -> last stamp {end:.1f} us")
    for name, ints, st, a, b in tl:
        print(f"s{streams[st]:<2d} {a:8.1f} -> {b:8.1f}  ({b - a:6.1f})  {name[3:]:22s} {list(ints[-6:])}")


if __name__ == "__main__":
    main()
