import os, sys, socket
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
mode = sys.argv[1]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", IB_DDP_SELFTEST="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from inferbiomechanics_amd.engine import HipTrainer
from inferbiomechanics_amd._tuning import tuning as TU
from test_ddp_rccl_gpu import _transformer
B, T, D = 128, 32, 48
dt = torch.bfloat16
g = torch.Generator().manual_seed(3)
batches = [(torch.randn(B, T, D, generator=g).to("cuda", dt), torch.randint(0, 1000, (B,), generator=g).cuda(), torch.randn(B, T, D, generator=g).to("cuda", dt)) for _ in range(3)]
import time
t00 = time.time()
def run(captured, bopt=True, mb=0.5, lag=True, ddp=True):
    TU.no_bucket_opt = not bopt
    TU.no_lag_group = not lag
    os.environ["IB_DDP_SELFTEST"] = "1" if ddp else "0"
    print("start", round(time.time() - t00, 2), captured, bopt, lag, ddp, flush=True)
    os.environ["IB_GRAPH_COLLECTIVES"] = "1" if captured else "0"
    tr = HipTrainer(_transformer(dt, T, D), "diffusion", "sgd", 1e-2, bucket_mb=mb, overlap_comm=True)
    for i in range(6):
        tr.step(batches[i % 3])
    torch.cuda.synchronize()
    print("ok", captured, bopt, mb, len(tr.buckets.ranges), tr.loss_value(), flush=True)
if mode == "cap_only": run(True)
if mode == "cap_nobopt": run(True, bopt=False)
if mode == "cut_then_cap": run(False); run(True)
if mode == "cap_big": run(True, mb=8.0)
if mode == "full": run(False); run(False, lag=False); run(False, ddp=False); run(False, bopt=False); run(True)
if mode == "nosingle": run(False); run(False, lag=False); run(False, bopt=False); run(True)
if mode == "single_cap": run(False, ddp=False); run(True)
if mode == "nobopt_cap": run(False, bopt=False); run(True)
dist.destroy_process_group()
