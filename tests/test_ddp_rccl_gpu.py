"""The data-parallel code paths over REAL RCCL on a one-rank group (a fresh process each: `nccl` backend, IB_DDP_SELFTEST=1
makes the trainer issue its collectives although world == 1).  What two ranks sharing the card over gloo
(tests/test_ddp_numerics_gpu.py) cannot reach:

* the LAGGED weight-gradient path of the overlapped transformer step (plans.TransformerLayerPlan.lag_group / take_lagged:
  bf16, B*T >= 4096, buckets all-reduced at the layer flush points, a layer's grouped launch + reduction forked beside
  the next layer's backward, graph cuts with that side stream in flight) -- bitwise against the same run with
  IB_NO_LAG_GROUP=1 and against the plain one-GPU step; gradients reported in layout order;
* `torch.nn.parallel.DistributedDataParallel(HipModule)` (cli/train.py `--eager`): one reference-style
  step (autograd node -> DDP's bucketed all-reduce -> torch.optim) against the fused trainer's step on the same batch.
  Reference: src/cli/train.py:99-102,175,281.  -m gpu."""
import os
import socket

import pytest
import torch

from inferbiomechanics_amd._tuning import tuning as TU

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _transformer(dtype, T, D, seed=7):
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    torch.manual_seed(seed)
    # d_model = 512 with a 1024-wide feed-forward: the fused token-local launches (csrc/ffn_chain.hip) are part of the path
    return DiffusionTransformer(D, T, d_model=512, num_heads=8, dim_feedforward=1024, num_layers=3, temporal_embedding_dim=6,
                                temb_dim=32, temb_hidden=64, device="cuda", compute_dtype=dtype)


def _lag_worker(port, q):
    try:
        import faulthandler
        import sys
        # a hang shows WHERE (all threads) instead of a silent timeout: into a file the GPU box hands back
        root_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root_, "gpurun_out"), exist_ok=True)
        _fh = open(os.path.join(root_, "gpurun_out", f"rccl_worker_hang_{os.getpid()}.txt"), "w")
        faulthandler.dump_traceback_later(150, exit=True, file=_fh)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0", IB_DDP_SELFTEST="1",
                          TORCH_NCCL_CUDA_EVENT_CACHE=os.environ.get("TORCH_NCCL_CUDA_EVENT_CACHE", "0"),
                          TORCH_NCCL_TRACE_BUFFER_SIZE=os.environ.get("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000"),
                          TORCH_FR_BUFFER_SIZE=os.environ.get("TORCH_FR_BUFFER_SIZE", "2000"))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from inferbiomechanics_amd.engine import HipTrainer
        B, T, D, steps = 128, 32, 48, 6                        # B * T = 4096 rows: the grouped / lagged path's threshold
        dt = torch.bfloat16
        g = torch.Generator().manual_seed(3)
        batches = [(torch.randn(B, T, D, generator=g).to("cuda", dt), torch.randint(0, 1000, (B,), generator=g).cuda(),
                    torch.randn(B, T, D, generator=g).to("cuda", dt)) for _ in range(3)]

        def run(lag: bool, ddp: bool, bopt: bool = True, captured: bool = False):
            TU.no_lag_group = not lag
            TU.no_bucket_opt = not bopt
            os.environ["IB_GRAPH_COLLECTIVES"] = "1" if captured else "0"      # forced: no start-up probe in this test
            os.environ["IB_DDP_SELFTEST"] = "1" if ddp else "0"
            print(f"[arm] lag={lag} ddp={ddp} bucket_opt={bopt} captured={captured}", file=sys.stderr, flush=True)
            model = _transformer(dt, T, D)
            tr = HipTrainer(model, "diffusion", "sgd", 1e-2, bucket_mb=0.5, overlap_comm=True if ddp else None)
            from inferbiomechanics_amd import hip
            nopt = [0]
            real = hip.optim_step

            def counted(*a, **k):
                nopt[0] += 1
                return real(*a, **k)
            hip.optim_step = counted
            info = {"ddp": tr.ddp, "overlap": tr.overlap_comm, "buckets": len(tr.buckets.ranges), "bucket_opt": tr.bucket_opt,
                    "fused_ffn": all(lp.ffn_fused(B * T) for lp in tr.plan.layers),
                    "lagging": [bool(lp.lag_group and lp.parent_flushes) for lp in tr.plan.layers]}
            losses = []
            for i in range(steps):
                tr.step(batches[i % len(batches)])
                if i == 0:
                    info["ready_is_layout_order"] = tr._ready_seen == list(tr.layout.keys())
                    info["optimizer_launches_first_step"] = nopt[0]
                losses.append(tr.loss_value())
            hip.optim_step = real
            torch.cuda.synchronize()
            info["captured"] = tr._rec is not None
            from inferbiomechanics_amd import engine as _eng
            info["drain"] = dict(_eng._drain_report)
            info["graph_cuts"] = sum(1 for k, _ in tr._rec.actions if k == "host") if tr._rec is not None else -1
            return tr.flat.detach().cpu().numpy().copy(), losses, info

        out = {"lag": run(True, True), "nolag": run(False, True), "single": run(True, False),
               "one_opt_launch": run(True, True, bopt=False), "captured": run(True, True, captured=True)}
        TU.no_bucket_opt = False
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
        _fh.close()
        os.unlink(_fh.name)                           # no hang: nothing to hand back
        q.put(("ok", out))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("FAIL: " + traceback.format_exc(), None))


def _spawn(target):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=target, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=200)
    p.join(timeout=120)
    assert status == "ok", status
    return out


def test_lagged_weight_gradient_path_is_bitwise_the_unlagged_and_the_single_gpu_step():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    out = _spawn(_lag_worker)
    (fa, la, ia), (fb, lb, ib), (fc, lc, ic) = out["lag"], out["nolag"], out["single"]
    assert ia["ddp"] and ia["overlap"] and ia["buckets"] > 1 and ia["captured"] and ia["graph_cuts"] >= ia["buckets"]
    assert ia["lagging"] == [True, True, True], ia        # every layer (round 5: the last-run one lags beside the step's tail)
    assert ia["fused_ffn"] and ib["fused_ffn"] and ic["fused_ffn"]
    assert ib["lagging"] == [False, False, False] and not ic["ddp"]
    assert ia["ready_is_layout_order"] and ib["ready_is_layout_order"] and ic["ready_is_layout_order"]
    assert la == lb and np.array_equal(fa, fb), "lagging a layer's weight-gradient launch changed the numbers"
    # the one-GPU step sums the same slabs inside the optimizer launch instead of ib_step_reduce: same fixed order
    assert la == lc and np.array_equal(fa, fc), (la, lc, float(np.abs(fa - fc).max()))
    # round 5: the optimizer runs per bucket behind that bucket's all-reduce (one launch per bucket + the self-counting last
    # launch, eager warm-up, graph replay with the launches as host actions of the cuts) -- bitwise the single launch over
    # the whole flat buffer behind the last all-reduce
    fd, ld, idd = out["one_opt_launch"]
    assert ia["bucket_opt"] and ia["optimizer_launches_first_step"] == ia["buckets"] + 1, ia
    assert not idd["bucket_opt"] and idd["optimizer_launches_first_step"] == 1, idd
    assert la == ld and np.array_equal(fa, fd), (la, ld, float(np.abs(fa - fd).max()))
    # ... and with the all-reduces (and the per-bucket optimizer branches behind them) captured inside ONE graph per step
    fe, le, ie = out["captured"]
    assert ie["captured"] and ie["graph_cuts"] == 0 and ie["bucket_opt"], ie
    # the capture waited until c10d's flight recorder showed no eager collective left with the watchdog (not a timed sleep)
    assert ie["drain"]["mode"] == "flight-recorder" and ie["drain"]["polls"] >= 1, ie["drain"]
    assert la == le and np.array_equal(fa, fe), (la, le, float(np.abs(fa - fe).max()))


def _eager_ddp_worker(port, q):
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import argparse
        from inferbiomechanics_amd import hip
        from inferbiomechanics_amd.engine import HipTrainer
        from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
        from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
        from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
        from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
        from oracle.fixture_inputs import ff_inputs, ff_labels
        res = {}
        # ---- regression model, the reference's loop body (train.py:240-284) under DDP
        args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                                  predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
        inputs = {k: v.cuda() for k, v in ff_inputs(6, 10, 23, 5).items()}
        labels = {k: v.cuda() for k, v in ff_labels(6, 10).items()}

        def ff():
            torch.manual_seed(11)
            return FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[64, 48], device="cuda")
        m1 = ff()
        m1.ensure_packed()
        ddp = DDP(m1, device_ids=[0], output_device=0)
        opt = torch.optim.RMSprop(m1.parameters(), lr=1e-3)
        ev = RegressionLossEvaluator(dataset=None, split="train", device="cuda")
        l_eager = []
        for _ in range(3):
            opt.zero_grad()
            loss = ev(dict(inputs), ddp(dict(inputs)), dict(labels), [], [], args)
            loss.backward()
            opt.step()
            l_eager.append(float(loss.detach().cpu()))
        m2 = ff()
        tr = HipTrainer(m2, "regression", "rmsprop", 1e-3, args=args)
        l_fused = []
        for _ in range(3):
            tr.step((inputs, labels))
            l_fused.append(tr.loss_value())
        res["ff"] = (l_eager, l_fused, max(float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12))
                                           for (_, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters())))
        # ---- diffusion MLP under DDP, fp32
        g = torch.Generator().manual_seed(5)
        x0, t, eps = torch.randn(8, 10, 44, generator=g).cuda(), torch.randint(0, 1000, (8,), generator=g).cuda(), \
            torch.randn(8, 10, 44, generator=g).cuda()

        def dm():
            torch.manual_seed(12)
            return DiffusionMLP(44, [64, 96], temb_dim=32, temb_hidden=48, device="cuda")
        m3 = dm()
        m3.ensure_packed()
        ddp3 = DDP(m3, device_ids=[0], output_device=0)
        opt3 = torch.optim.Adam(m3.parameters(), lr=1e-3)
        ev3 = DiffusionLossEvaluator("train")
        tabs = m3.tables(torch.device("cuda", 0))
        l_eager3 = []
        for _ in range(3):
            opt3.zero_grad()
            xt = torch.empty_like(x0)
            hip.q_sample(x0, eps, t, tabs.sqrt_ab, tabs.sqrt_1mab, xt)
            loss = ev3(ddp3(xt, t), eps)
            loss.backward()
            opt3.step()
            l_eager3.append(float(loss.detach().cpu()))
        m4 = dm()
        tr4 = HipTrainer(m4, "diffusion", "adam", 1e-3)
        l_fused3 = []
        for _ in range(3):
            tr4.step((x0, t, eps))
            l_fused3.append(tr4.loss_value())
        res["mlp"] = (l_eager3, l_fused3, max(float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12))
                                              for (_, a), (_, b) in zip(m3.named_parameters(), m4.named_parameters())))
        res["ddp_buckets_built"] = bool(ddp.reducer is not None)
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", res))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("FAIL: " + traceback.format_exc(), None))


def test_hipmodule_under_torch_ddp_matches_the_fused_trainer():
    """`DistributedDataParallel(model)` works over a HipModule (its parameters are ordinary leaves that
    are views of one flat buffer; the plan is ONE autograd node, so DDP's hooks fire when its gradients land)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = _spawn(_eager_ddp_worker)
    for key in ("ff", "mlp"):
        eager, fused, perr = res[key]
        for a, b in zip(eager, fused):
            assert abs(a - b) <= 2e-5 * abs(b), (key, eager, fused)
        assert perr <= 5e-4, (key, perr)            # three optimizer steps; RMSprop / Adam amplify 1e-7 gradient differences
    assert res["ddp_buckets_built"]
