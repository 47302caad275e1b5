"""DDIM loop rate (bench.ddim_leg) over a list of batch sizes, for same-box A/B runs under environment switches.
Usage (GPU box): python tools/ddim_ab.py [B ...]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(v) for v in sys.argv[1:]] or (1, 2, 4, 8, 16, 32):
    r = bench.ddim_leg(dev, torch.bfloat16, B=B)
    print(B, r["steps_per_sec"], flush=True)
