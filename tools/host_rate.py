"""How fast can the host ENQUEUE training steps (no synchronisation inside the loop) against how fast the GPU retires
them?  If the two times agree the step is host-bound.  IB_DDP_SELFTEST=1: the data-parallel launch sequence (graph
segments cut around the RCCL all-reduce) on one rank.  Usage (GPU box): python tools/host_rate.py [workload]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "mlp_denoiser_T50"
    kind, T, D, B = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if os.environ.get("IB_DDP_SELFTEST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    model = bench.build_model(kind, T, D, torch.bfloat16, dev)
    batches = bench.make_batches(8, B, T, D, torch.bfloat16, dev, seed=0)
    tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4)
    for i in range(50):
        tr.step(batches[i % 8])
    torch.cuda.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(batches[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # bursts of 4 steps into an EMPTY queue: the host cost without back-pressure from a full queue
    burst = []
    for r in range(50):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for i in range(4):
            tr.step(batches[i % 8])
        burst.append((time.perf_counter() - a) / 4)
    torch.cuda.synchronize()
    burst.sort()
    print(f"{wl}: host cost per step into an empty queue: median {1e3 * burst[25]:.4f} ms, min {1e3 * burst[0]:.4f} ms")
    print(f"{wl}: host enqueue {1e3 * (t1 - t0) / n:.4f} ms/step, device retire {1e3 * (t2 - t0) / n:.4f} ms/step "
          f"(ddp={tr.ddp}, overlap_comm={tr.overlap_comm})")


if __name__ == "__main__":
    main()
