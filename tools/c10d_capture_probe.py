"""Which c10d Work makes the watchdog's event query fail during a hipGraph capture?  Each case runs in a child process on a
1-rank RCCL group (the failure is a process abort) and holds the capture open for 0.4 s so that watchdog passes (every
~100 ms) land inside it.  Usage (GPU box): python tools/c10d_capture_probe.py"""
import os
import subprocess
import sys
import time

CASES = ("retired_flag_over_time", "captured_collective_only", "pending_eager_work_capture_without_collective",
         "pending_eager_work_capture_with_collective", "retired_eager_work_capture_with_collective",
         "captured_collective_short_capture_then_idle")


def child(case):
    import pickle
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PROBE_PORT", "29581"), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_NCCL_CUDA_EVENT_CACHE=os.environ.get("EVC", "0"),
                      TORCH_FR_BUFFER_SIZE="2000", TORCH_NCCL_TRACE_BUFFER_SIZE="2000")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    c = torch._C._distributed_c10d

    def active(show=False):
        e = pickle.loads(c._dump_nccl_trace(True, False, True)).get("entries") or []
        if show:
            allx = pickle.loads(c._dump_nccl_trace(True, False, False)).get("entries") or []
            print(f"[{case}]   FR:", [(x.get("record_id"), x.get("collective_seq_id"), x.get("state"), x.get("retired")) for x in allx],
                  flush=True)
        return sum(1 for x in e if not x.get("retired", False))

    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t)                       # communicator up
    torch.cuda.synchronize()
    while active():
        time.sleep(0.01)
    print(f"[{case}] start: active works {active()}", flush=True)
    if case == "retired_flag_over_time":
        w = dist.all_reduce(t, async_op=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            allx = pickle.loads(c._dump_nccl_trace(True, False, False)).get("entries") or []
            print(f"[{case}] +{(time.perf_counter() - t0) * 1e3:5.0f} ms:",
                  [(x.get("record_id"), x.get("state"), x.get("retired")) for x in allx], flush=True)
            time.sleep(0.05)
        dist.destroy_process_group()
        return
    cap = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    pending = case.startswith("pending")
    if pending:
        torch.cuda._sleep(int(2.0e9))        # ~1 s of GPU time ahead of the eager collective: its Work stays with the watchdog
        w = dist.all_reduce(t, async_op=True)
        print(f"[{case}] eager work issued: active {active()}", flush=True)
    if case.startswith("retired"):
        w = dist.all_reduce(t, async_op=True)
        torch.cuda.synchronize()
        while active():
            time.sleep(0.005)
        print(f"[{case}] eager work retired: {active(True)}", flush=True)
    with torch.cuda.stream(cap):
        g.capture_begin()
        t.mul_(1.0)
        if "with_collective" in case or case.startswith("captured"):
            w2 = dist.all_reduce(t, async_op=True)
            w2.wait()
            print(f"[{case}] collective captured: active {active(True)}", flush=True)
        if case != "captured_collective_short_capture_then_idle":
            time.sleep(0.4)
        g.capture_end()
    print(f"[{case}] capture ended", flush=True)
    time.sleep(0.4)                          # watchdog passes AFTER the capture
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print(f"[{case}] SURVIVED (active {active()})", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for i, case in enumerate(CASES):
            env = dict(os.environ, PROBE_PORT=str(29581 + i))
            p = subprocess.run([sys.executable, __file__, case], env=env, capture_output=True, text=True, timeout=120)
            out = [l for l in (p.stdout + p.stderr).splitlines()
                   if l.startswith("[" + case) or "rror" in l and "frame" not in l][:14]
            print(f"== {case}: exit code {p.returncode}")
            for l in out:
                print("   ", l[:200])
