#!/usr/bin/env python3
"""Loss trajectory of the fused transformer step (B = 128, T = 50, bf16, 4 steps) as ONE JSON line -- run it under different
kernel-selection switches (IB_NO_NT / IB_NO_TN / IB_NO_WGRAD_BIAS, IB_DDP_SELFTEST) and compare: the switches are read once
per process, so an A/B needs one process per arm (tests/test_trainer_gpu.py::test_large_batch_kernel_paths_agree)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if os.environ.get("IB_DDP_SELFTEST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    from inferbiomechanics_amd.engine import HipTrainer
    model = bench.build_model("transformer", 50, 300, torch.bfloat16, dev)
    tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4, overlap_comm=True if os.environ.get("IB_DDP_SELFTEST") == "1" else None)
    batches = bench.make_batches(2, 128, 50, 300, torch.bfloat16, dev, seed=3)
    losses = []
    for i in range(4):
        tr.step(batches[i % 2])
        losses.append(tr.loss_value())
    sys.stdout.write(json.dumps({"losses": losses, "psum": float(tr.flat.double().abs().sum())}) + "\n")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
