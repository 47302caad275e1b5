"""Data-parallel wiring on CPU with the gloo backend, world_size 2 (the N > 1 path cannot be exercised on
the one-GPU box): ready-order gradient buckets all-reduce to the SUM, the optimizer's grad_scale turns it
into DDP's mean, parameters are broadcast from rank 0, every bucket is launched exactly once per step, and
the reference's DistributedSampler(shuffle=False, drop_last=True) sharding (train.py:143) is rank::world."""
import os
import socket
from collections import OrderedDict

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from inferbiomechanics_amd import hip
        from inferbiomechanics_amd.engine import GradBuckets, HipTrainer, broadcast_parameters
        from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
        from inferbiomechanics_amd.module import flat_layout

        # 1) buckets: contiguous ready-order slices, SUM across ranks
        shapes = OrderedDict(a=(100,), b=(7, 9), c=(300,), d=(64,), e=(5,))
        lay, total = flat_layout(shapes)
        g = torch.zeros(total)
        for i, (k, (off, n)) in enumerate(lay.items()):
            g[off:off + n] = (rank + 1) * (i + 1)
        bk = GradBuckets(g, lay, bucket_bytes=1200)
        assert len(bk.ranges) >= 2 and bk.ranges[0][0] == 0 and bk.ranges[-1][1] == total
        assert all(bk.ranges[i][1] == bk.ranges[i + 1][0] for i in range(len(bk.ranges) - 1))
        launched = []
        for k in lay:
            b = bk.mark_ready(k)
            if b is not None:
                launched.append(b)
                bk.launch(b)
        bk.finish()
        assert launched == list(range(len(bk.ranges)))
        for i, (k, (off, n)) in enumerate(lay.items()):
            assert torch.all(g[off:off + n] == 3 * (i + 1)), k           # (1 + 2) * (i + 1)
        bk.reset()
        # 2) parameter broadcast from rank 0
        p = torch.full((10,), float(rank + 5))
        broadcast_parameters(p)
        assert torch.all(p == 5.0)
        # 3) the fused trainer's step sequence under world_size 2 (kernels in dry-run: host logic only)
        hip.set_dry_run(True)
        torch.manual_seed(rank)             # different init per rank -> must be equalised by the broadcast
        m = DiffusionMLP(12, [16, 24], temb_dim=8, temb_hidden=16)
        for overlap in (True, False):      # large-model policy (bucket overlap) / small-model policy (one all-reduce)
            tr = HipTrainer(m, "diffusion", "adam", 1e-3, bucket_mb=0.001, overlap_comm=overlap)
            assert tr.world == 2 and tr.ddp and tr.overlap_comm == overlap
            assert (len(tr.buckets.ranges) > 1) == overlap
            ref = tr.flat.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, tr.flat)
            for _ in range(3):
                tr.grad.fill_(float(rank + 1))
                tr.step((torch.randn(2, 5, 12), torch.tensor([1, 2]), torch.randn(2, 5, 12)))
                assert tr.buckets._works == [] and all(v == 0 for v in tr.buckets._pending)
                assert torch.all(tr.grad == 3.0)      # dry-run kernels write nothing: the SUM of the fills remains
        # 4) transformer denoiser, overlapped policy: the layers' side streams stay on and completed buckets are launched
        #    at the plan's flush points (every layer boundary) -- every bucket exactly once, all gradients summed
        from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
        torch.manual_seed(rank)
        mt = DiffusionTransformer(12, 5, d_model=16, num_heads=2, dim_feedforward=32, num_layers=3)
        tr = HipTrainer(mt, "diffusion", "adam", 1e-3, bucket_mb=0.004, overlap_comm=True)
        assert tr._flush_mode and len(tr.buckets.ranges) > 3 and all(lp.flush_on_exit for lp in tr.plan.layers)
        launches = []
        orig = tr.buckets.launch
        tr.buckets.launch = lambda b, inline=False: (launches.append(b), orig(b, inline))[1]
        for _ in range(2):
            launches.clear()
            tr.grad.fill_(float(rank + 1))
            tr.step((torch.randn(2, 5, 12), torch.tensor([1, 2]), torch.randn(2, 5, 12)))
            assert sorted(launches) == list(range(len(tr.buckets.ranges))), launches
            assert tr.buckets._works == [] and all(v == 0 for v in tr.buckets._pending)
            assert torch.all(tr.grad == 3.0)
        hip.set_dry_run(False)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_ddp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_distributed_sampler_sharding_rule():
    from torch.utils.data.distributed import DistributedSampler
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    ds = SyntheticWindowDataset(11)
    s0 = list(DistributedSampler(ds, num_replicas=2, rank=0, shuffle=False, drop_last=True))
    s1 = list(DistributedSampler(ds, num_replicas=2, rank=1, shuffle=False, drop_last=True))
    assert s0 == [0, 2, 4, 6, 8] and s1 == [1, 3, 5, 7, 9]
    a, la, _, _ = ds[3]
    b, lb, _, _ = ds[3]
    assert all(torch.equal(a[k], b[k]) for k in a) and a['pos'].shape == (10, 23)
    assert la['groundContactWrenchesInRootFrame'].shape == (10, 12)


def _probe_worker(rank, world, port, q):
    """ddp_probe.decide() across two ranks: rank 0 runs the (here: stubbed) child job, the verdict reaches rank 1 through the
    process group's store -- no collective -- and both remember it"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        os.environ.pop("IB_GRAPH_COLLECTIVES", None)
        os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "0"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sys
        import time
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from inferbiomechanics_amd import ddp_probe
        calls = []

        def stub(n, timeout_s=150.0):
            calls.append(n)
            time.sleep(0.5)                       # the other rank must WAIT for the verdict, not run ahead
            return {"ok": True, "why": "stub", "seconds": 0.5}
        ddp_probe.run_child = stub
        t0 = time.monotonic()
        got = ddp_probe.decide(world, rank, "nccl")        # (the backend NAME selects the probe; the store is the group's)
        waited = time.monotonic() - t0
        assert got is True and os.environ["IB_GRAPH_COLLECTIVES"] == "1" and ddp_probe.verdict()["source"] == "probe"
        assert calls == ([2] if rank == 0 else []) and waited >= 0.4, (rank, calls, waited)
        # a second decision in the same process is answered from the environment: no second child job
        assert ddp_probe.decide(world, rank, "nccl") is True and calls == ([2] if rank == 0 else [])
        # a failed probe: every rank falls back to the cut-graph form
        os.environ.pop("IB_GRAPH_COLLECTIVES")
        ddp_probe.run_child = lambda n, timeout_s=150.0: {"ok": False, "why": "timed out after 150 s", "seconds": 150.0}
        assert ddp_probe.decide(world, rank, "nccl") is False and os.environ["IB_GRAPH_COLLECTIVES"] == "0"
        assert "timed out" in ddp_probe.verdict()["why"]
        os.environ.pop("IB_GRAPH_COLLECTIVES")
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_probe_verdict_reaches_every_rank_through_the_store():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
