"""Start-up probe of the data-parallel step with its gradient all-reduces CAPTURED inside the step's hipGraph.

The captured form (one graph per step: no graph cut and no host action per collective) is the faster one, but whether RCCL
kernels enqueued by c10d can be captured and replayed depends on the RCCL / driver / topology at hand -- and a failure
there is a hang or an abort, not an exception.  So the question is never asked in a process that matters:

    parent (every rank, HipTrainer.__init__ -> decide())
        rank 0 starts a FRESH child job -- `python -m torch.distributed.run --nproc-per-node min(world, 2)
        -m inferbiomechanics_amd.ddp_probe` (subprocess; nothing that has touched a GPU is ever exec'd) -- bounded by a
        timeout, its process group killed as a whole when it expires;
        the child ranks build two small models, run the cut-graph form and the captured form on the same batches
        (bucketed + overlapped policy and the one-bucket policy) and compare the trajectories bit for bit;
        exit code 0 + the OK line -> every parent rank uses the captured form; anything else -> the cut-graph form.
    The verdict reaches the other parent ranks through the process group's key-value store: no collective is issued and
    their GPUs stay idle while the child runs.

The captured form is only considered when c10d's event cache is off (TORCH_NCCL_CUDA_EVENT_CACHE=0 at process-group
creation: `prepare_env()`, called by bench.py and `main.py train`) -- see decide().
IB_GRAPH_COLLECTIVES=1 / =0 in the environment skips the probe (forces the form); the verdict of a probe is remembered in
that variable for the rest of the process.  Reference: the step this is about is DDP's bucketed all-reduce inside
backward, src/cli/train.py:99-102,175,281."""
import os
import socket
import subprocess
import sys
import time

OK_LINE = "IB_DDP_PROBE_OK"


def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_child(world: int, timeout_s: float = 150.0) -> dict:
    """rank 0 of the parent job: launch the probe as `world` fresh ranks, wait at most timeout_s"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
              "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
              "IB_GRAPH_COLLECTIVES", "IB_BENCH_REHEARSAL"):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=env.get("OMP_NUM_THREADS", "4"),
               IB_DDP_SELFTEST="1" if world == 1 else "0", IB_DDP_PROBE_CHILD="1", TORCH_NCCL_CUDA_EVENT_CACHE="0",
               TORCH_NCCL_TRACE_BUFFER_SIZE=env.get("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000"),
               TORCH_FR_BUFFER_SIZE=env.get("TORCH_FR_BUFFER_SIZE", "2000"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "-m", "inferbiomechanics_amd.ddp_probe"]
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(p.pid, signal.SIGKILL)          # the launcher's own process group: exactly the children started here
            p.communicate()
            return {"ok": False, "why": f"timed out after {timeout_s:.0f} s", "seconds": round(time.perf_counter() - t0, 1)}
        ok = p.returncode == 0 and OK_LINE in out.decode(errors="replace")
        why = "captured and cut-graph trajectories agree bit for bit" if ok else \
            f"child exit code {p.returncode}: " + err.decode(errors="replace")[-300:].replace("\n", " | ")
        return {"ok": ok, "why": why, "seconds": round(time.perf_counter() - t0, 1)}
    except Exception as exc:                          # nothing in here may cost the run
        return {"ok": False, "why": repr(exc)[:300], "seconds": round(time.perf_counter() - t0, 1)}


PROBE_RANKS = 2       # the child job's size (at most the parent's world): two ranks put RCCL's inter-GPU transport inside
                      # the captured graph, which is what can fail; a world-sized child would double the processes on the node
_probe_no = 0
_verdict = None      # the last decision of this process: {"captured": bool, "source": ..., "why": ...}


def verdict():
    return _verdict


def event_cache_off() -> bool:
    return os.environ.get("TORCH_NCCL_CUDA_EVENT_CACHE", "1").strip().lower() in ("0", "false", "off", "n", "no")


def prepare_env():
    """call BEFORE dist.init_process_group("nccl"): what the captured form needs from the process group (and dmabuf IPC)"""
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
    # c10d's flight recorder on: engine._drain_c10d_watchdog reads it to know when the watchdog has let go of every eager
    # collective (both spellings: the variable was renamed between torch releases)
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def decide(world: int, rank: int, backend: str) -> bool:
    """collective over the parent job's ranks: True = capture the all-reduces inside the step graph"""
    global _verdict
    import torch.distributed as dist
    forced = os.environ.get("IB_GRAPH_COLLECTIVES")
    if forced in ("0", "1"):
        if _verdict is None or _verdict.get("source") == "environment":
            _verdict = {"captured": forced == "1", "source": "environment", "why": f"IB_GRAPH_COLLECTIVES={forced}"}
        return forced == "1"
    if backend == "nccl" and not event_cache_off():
        # A precaution kept from the hunt for the watchdog abort ("event last recorded in a capturing stream"): c10d recycles
        # the completion events of its Works through a cache, so a captured Work and a later eager one can share an event
        # object.  The two causes that were found and fixed are elsewhere (engine._drain_c10d_watchdog waits on the flight
        # recorder's `retired` flags; hip.new_stream keeps the library's streams out of the pool c10d's stream comes from);
        # with both, 24 stress runs with the cache on and 30 with it off passed (profiles/r05_rccl_stress.txt).  The cache
        # is a property of the process group, fixed when it was created.
        _verdict = {"captured": False, "source": "event-cache",
                    "why": "TORCH_NCCL_CUDA_EVENT_CACHE was not 0 when the process group was created: captured collectives "
                           "are only used with c10d's event cache off (bench.py / main.py train set it)"}
        os.environ["IB_GRAPH_COLLECTIVES"] = "0"
        return False
    if backend != "nccl" or os.environ.get("IB_DDP_PROBE_CHILD") == "1":
        _verdict = {"captured": False, "source": "backend", "why": f"backend {backend}: collectives run on the host"}
        os.environ["IB_GRAPH_COLLECTIVES"] = "0"
        return False
    # the verdict travels through the process group's own key-value store (rank 0 sets a key, the others block on it):
    # no extra group, no collective, and the other ranks' GPUs stay idle while the child job runs
    import json
    global _probe_no
    _probe_no += 1
    key = f"ib_ddp_probe_verdict_{_probe_no}"
    res = [None]
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if rank == 0:
        res[0] = run_child(min(world, PROBE_RANKS))
        if world > 1 and store is not None:
            store.set(key, json.dumps(res[0]))
    elif store is not None:
        store.wait([key])
        res[0] = json.loads(store.get(key).decode())
    if world > 1 and store is None:                   # no store to be had: a host-side group for one broadcast
        g = dist.new_group(backend="gloo")
        dist.broadcast_object_list(res, src=0, group=g)
        dist.destroy_process_group(g)
    r = res[0]
    _verdict = {"captured": bool(r["ok"]), "source": "probe", "why": r["why"], "seconds": r["seconds"]}
    os.environ["IB_GRAPH_COLLECTIVES"] = "1" if r["ok"] else "0"
    return bool(r["ok"])


def main() -> int:
    """the child job's ranks"""
    import torch
    import torch.distributed as dist
    prepare_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    from .engine import HipTrainer
    from .models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    dt = torch.bfloat16

    def model(kind):
        torch.manual_seed(11)
        if kind == "mlp":
            return DiffusionMLP(48, [128, 128], temb_dim=32, temb_hidden=64, device=dev, compute_dtype=dt)
        return DiffusionTransformer(48, 16, d_model=128, num_heads=2, dim_feedforward=256, num_layers=2, temb_dim=32,
                                    temb_hidden=64, device=dev, compute_dtype=dt)

    g = torch.Generator().manual_seed(100 + rank)
    batches = [(torch.randn(8, 16, 48, generator=g).to(dev, dt), torch.randint(0, 1000, (8,), generator=g).to(dev),
                torch.randn(8, 16, 48, generator=g).to(dev, dt)) for _ in range(3)]
    for kind, overlap in (("transformer", True), ("mlp", False)):
        traj = {}
        for captured in ("0", "1"):
            os.environ["IB_GRAPH_COLLECTIVES"] = captured
            tr = HipTrainer(model(kind), "diffusion", "sgd", 1e-2, bucket_mb=0.05, overlap_comm=overlap, use_graph=True)
            losses = []
            for i in range(8):                        # two eager warm-ups, the capture, replays
                tr.step(batches[i % 3])
                losses.append(tr.loss_value())
            torch.cuda.synchronize()
            if tr._rec is None:
                raise SystemExit(f"probe: the {kind} step was not captured")
            traj[captured] = (losses, tr.flat.detach().clone())
            del tr
        if traj["0"][0] != traj["1"][0] or not torch.equal(traj["0"][1], traj["1"][1]):
            raise SystemExit(f"probe: captured collectives changed the {kind} trajectory: {traj['0'][0]} vs {traj['1'][0]}")
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    if rank == 0:
        print(OK_LINE, flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
