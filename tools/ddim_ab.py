import sys, torch, json
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for B in (1, 2, 4, 16):
    r = bench.ddim_leg(dev, torch.bfloat16, B=B)
    print(B, r["steps_per_sec"], flush=True)
