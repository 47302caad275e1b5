"""The fused training step (engine.HipTrainer: flat buffers, forked branches, hipGraph replay, self-counting
optimizer) against the CPU oracle's training trajectory on identical seeded batches: per-step losses and the final
parameters must match ("matched diffusion loss", north_star).  fp32 mode <= 1e-3 relative; bf16 mode follows the
fp32 loss curve within 2 %.  Also: graph replay == eager launches bit-for-bit, regression task vs the oracle."""
import argparse

import pytest
import torch

from inferbiomechanics_amd._tuning import tuning as TU

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle.fixture_inputs import det_state, ff_inputs, ff_labels  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def load_det(module):
    sd = module.state_dict()
    new = det_state({k: tuple(v.shape) for k, v in sd.items()})
    module.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})


def batches(n, B, T, D, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, T, D, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randn(B, T, D, generator=g))
            for _ in range(n)]


def oracle_run(model, bs, hidden, opt, lr, steps):
    p = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    st = {k: R.optim_init_state(opt, v.detach()) for k, v in p.items()}
    tabs = R.schedule_tables()
    losses = []
    for i in range(steps):
        x0, t, eps = bs[i % len(bs)]
        for v in p.values():
            v.grad = None
        xt = R.q_sample(x0.double(), t, eps.double(), tabs)
        loss = R.eps_mse(R.denoiser_mlp_forward(p, xt, t, hidden, temb_dim=model.temb_dim), eps.double())
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            for k, v in p.items():
                v.copy_(R.optim_step(opt, v, v.grad, st[k], lr, i + 1))
    return losses, {k: v.detach() for k, v in p.items()}


@pytest.mark.parametrize("opt", ["rmsprop", "adam"])
def test_fused_trainer_matches_oracle_trajectory_fp32(opt):
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    hidden, B, T, D, steps, lr = [64, 96], 8, 10, 44, 8, 1e-3
    model = DiffusionMLP(D, hidden, temb_dim=32, temb_hidden=48, device=DEV)
    load_det(model)
    bs = batches(4, B, T, D)
    ref_losses, ref_p = oracle_run(model, bs, hidden, opt, lr, steps)
    tr = HipTrainer(model, "diffusion", opt, lr, use_graph=True)
    got = []
    for i in range(steps):                       # steps 0,1 eager, step 2 captured, 3.. replayed
        x0, t, eps = bs[i % len(bs)]
        tr.step((x0.to(DEV), t.to(DEV), eps.to(DEV)))
        got.append(tr.loss_value())
    assert tr._rec is not None, "the step was never captured into a hipGraph"
    for a, e in zip(got, ref_losses):
        assert abs(a - e) <= 1e-3 * abs(e), (got, ref_losses)
    assert int(tr.step_dev.cpu()) == steps
    for k, v in model.state_dict().items():
        e = ref_p[k]
        err = (v.detach().cpu().double() - e).abs().max().item()
        assert err <= 2e-3 * max(e.abs().max().item(), 1e-6) + 2e-5, (k, err)


def test_graph_replay_is_bitwise_equal_to_eager_launches():
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    bs = batches(3, 6, 10, 44, seed=3)
    finals = []
    for use_graph in (False, True):
        model = DiffusionMLP(44, [64, 64], temb_dim=32, temb_hidden=48, device=DEV, compute_dtype=torch.bfloat16)
        load_det(model)
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-3, use_graph=use_graph)
        for i in range(7):
            x0, t, eps = bs[i % 3]
            tr.step((x0.to(DEV, torch.bfloat16), t.to(DEV), eps.to(DEV, torch.bfloat16)))
        torch.cuda.synchronize()
        finals.append((tr.flat.clone(), tr.loss_value()))
    assert torch.equal(finals[0][0], finals[1][0]) and finals[0][1] == finals[1][1]


def test_bf16_trainer_tracks_fp32_loss_curve():
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    bs = batches(8, 32, 10, 44, seed=5)
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        model = DiffusionMLP(44, [64, 64], temb_dim=32, temb_hidden=48, device=DEV, compute_dtype=dt)
        load_det(model)
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-3)
        c = []
        for i in range(24):
            x0, t, eps = bs[i % 8]
            tr.step((x0.to(DEV, dt), t.to(DEV), eps.to(DEV, dt)))
            c.append(tr.loss_value())
        curves[dt] = c
    for a, b in zip(curves[torch.float32], curves[torch.bfloat16]):
        assert abs(a - b) <= 0.02 * abs(a), (curves[torch.float32], curves[torch.bfloat16])
    assert curves[torch.float32][-1] < curves[torch.float32][0]          # it learns


def test_regression_trainer_matches_oracle():
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    model = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[64, 48], device=DEV)
    load_det(model)
    inputs, labels = ff_inputs(6, 10, 23, 5), ff_labels(6, 10)
    p = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    st = {k: R.optim_init_state("rmsprop", v.detach()) for k, v in p.items()}
    ref = []
    for i in range(5):
        for v in p.values():
            v.grad = None
        layers = [(p[f"net.{2 * j}.weight"], p[f"net.{2 * j}.bias"]) for j in range(3)]
        out = R.feedforward_forward(layers, {k: v.double() for k, v in inputs.items()}, "sigmoid", 10)
        loss, _, _ = R.regression_loss(out, {k: v.double() for k, v in labels.items()}, range(6), range(6), range(6), range(12))
        loss.backward()
        ref.append(float(loss))
        with torch.no_grad():
            for k, v in p.items():
                v.copy_(R.optim_step("rmsprop", v, v.grad, st[k], 1e-3, i + 1))
    tr = HipTrainer(model, "regression", "rmsprop", 1e-3, args=args)
    got = []
    for i in range(5):
        tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
        got.append(tr.loss_value())
    for a, e in zip(got, ref):
        assert abs(a - e) <= 1e-3 * abs(e), (got, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_recurring_dict_batches_are_read_in_place_bitwise(dtype, monkeypatch):
    """a dict batch that comes back at the same device addresses (a loader recycling its buffers) gets, from its second
    appearance, a captured graph that reads the caller's tensors where they lie -- no staging copies.  Same steps, bit for
    bit, as a trainer that stages every batch into its static buffers (IB_NO_PINNED_GRAPHS), two alternating batches, and a
    tensor that does NOT qualify (non-contiguous) falls back to staging."""
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    batches = []
    for sd in (0, 1):
        i_, l_ = ff_inputs(8, 10, 23, 5), ff_labels(8, 10)
        g = torch.Generator().manual_seed(40 + sd)
        batches.append(({k: (v + 0.1 * torch.randn(v.shape, generator=g)).to(DEV) for k, v in i_.items()},
                        {k: (v + 0.1 * torch.randn(v.shape, generator=g)).to(DEV) for k, v in l_.items()}))
    res = {}
    for mode in ("in_place", "staged"):
        model = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[64, 48], device=DEV,
                                    compute_dtype=dtype)
        load_det(model)
        tr = HipTrainer(model, "regression", "rmsprop", 1e-3, args=args)
        if mode == "staged":
            monkeypatch.setattr(TU, "no_pinned_graphs", True)
        else:
            monkeypatch.setattr(TU, "no_pinned_graphs", False)
        losses = []
        for i in range(10):
            inp, lab = batches[i % 2]
            tr.step((inp, lab))
            losses.append(tr.result[0].clone())
        torch.cuda.synchronize()
        res[mode] = (torch.stack(losses).cpu(), tr.flat.detach().cpu().clone(), len(tr._pinned))
    assert res["in_place"][2] == 2 and res["staged"][2] == 0       # one graph per recurring batch / none
    assert torch.equal(res["in_place"][0], res["staged"][0])
    assert torch.equal(res["in_place"][1], res["staged"][1])
    # a batch with a non-contiguous tensor is staged (and still right)
    inp, lab = batches[0]
    k0 = next(iter(inp))
    wide = torch.zeros(inp[k0].shape[:-1] + (inp[k0].shape[-1] + 3,), device=DEV)
    wide[..., :inp[k0].shape[-1]] = inp[k0]
    odd = dict(inp)
    odd[k0] = wide[..., :inp[k0].shape[-1]]
    monkeypatch.setattr(TU, "no_pinned_graphs", False)
    before = len(tr._pinned)
    for _ in range(3):
        tr.step((odd, lab))
    assert len(tr._pinned) == before


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_cache_step_is_bitwise_the_dict_batch_step(dtype):
    """training from the on-device window cache (one gather launch per batch) must be EXACTLY
    training from the reference-layout dict batches of the same windows (gather + cast vs concat + cast: same values)"""
    from torch.utils.data import DataLoader
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    from inferbiomechanics_amd.data.WindowCache import DeviceWindowCache, PackedWindows
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    ds = SyntheticWindowDataset(96, history_len=50, stride=5, seed=11)

    def fresh():
        m = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, device=DEV, compute_dtype=dtype)
        load_det(m)
        return m, HipTrainer(m, "regression", "rmsprop", 1e-3, args=args)

    ma, ta = fresh()
    la = []
    for inputs, labels, _, _ in DataLoader(ds, batch_size=32, shuffle=False, drop_last=True):
        ta.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
        la.append(ta.loss_value())
    mb, tb = fresh()
    cache = DeviceWindowCache(PackedWindows.from_windows(ds), DEV)
    lb = []
    for idx in cache.batches(32):
        tb.step_windows(cache, idx)
        lb.append(tb.loss_value())
    assert len(la) == len(lb) == 3 and la == lb, (la, lb)
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k
    # a ragged gather (indices out of order, repeated) through the kernel alone
    from inferbiomechanics_amd import hip
    idx = torch.tensor([95, 0, 17, 17, 3], device=DEV)
    x = torch.zeros(5, 1470, dtype=dtype, device=DEV)
    labs = [torch.zeros(5, 10, c, device=DEV) for c in (6, 6, 6, 12)]
    hip.gather_windows(cache.table, idx, x, labs)
    torch.cuda.synchronize()
    for j, w in enumerate(idx.tolist()):
        inputs, labels, _, _ = ds[w]
        from inferbiomechanics_amd.data.AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER
        xe = torch.cat([inputs[k] for k in INPUT_KEY_ORDER], dim=-1).reshape(-1).to(dtype)
        assert torch.equal(x[j].cpu(), xe)
        for t, k in zip(labs, LOSS_KEY_ORDER):
            assert torch.equal(t[j].cpu(), labels[k])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transformer_fused_reductions_match_separate_reductions(dtype, monkeypatch):
    """one GPU: the optimizer launch adds up the weight-gradient slabs and the per-block partial sums of every bias /
    LayerNorm gradient of the transformer denoiser (<= 64 sources); under data parallelism each is reduced by its own
    launch.  Same numbers up to fp32 summation order: compared after 3 SGD steps on the parameter MOVEMENT."""
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    B, T, D = 24, 30, 44                                     # 720 tokens: bias sums go through partial-sum matrices
    bs = batches(3, B, T, D, seed=11)
    res = []
    for fuse in (True, False):
        if not fuse:
            monkeypatch.setattr(TU, "no_opt_fuse", True)
        torch.manual_seed(0)
        model = DiffusionTransformer(D, T, d_model=128, num_heads=4, dim_feedforward=256, num_layers=3, device=DEV,
                                     compute_dtype=dtype)
        p0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        tr = HipTrainer(model, "diffusion", "sgd", 1e-2)
        losses = []
        for i in range(4):
            x0, t, eps = bs[i % 3]
            tr.step((x0.to(DEV, dtype), t.to(DEV), eps.to(DEV, dtype)))
            losses.append(tr.loss_value())
        assert tr.plan.fuse_reduce_into_optimizer == fuse and tr.plan.pending_sources is None
        res.append((losses, {k: v.detach() - p0[k] for k, v in model.state_dict().items()}))
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 1e-5 * abs(b), (res[0][0], res[1][0])
    for k, mv in res[1][1].items():
        ref = mv.abs().max().item()
        assert ref > 0, k
        assert (res[0][1][k] - mv).abs().max().item() <= 2e-4 * ref, k


@pytest.mark.parametrize("cfg", [("adam", torch.bfloat16, 128, 32, 48, 512, 1024), ("rmsprop", torch.float32, 24, 30, 44, 128, 256)])
def test_layers_updated_early_are_bitwise_the_single_optimizer_launch(cfg, monkeypatch):
    """one GPU: every transformer layer's parameters are updated by a launch over that layer's range of the flat buffers on
    the layer's side stream (step number *step_dev + 1 without publishing), the step's last, self-counting launch skips
    those ranges (source kind 3).  Same sources summed in the same order: BITWISE the one-launch step (IB_NO_EARLY_OPT=1),
    eager steps and graph replays, with a step-dependent optimizer (Adam's bias corrections)."""
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    opt, dtype, B, T, D, dm, ff = cfg
    bs = [(x.to(DEV, dtype), t.to(DEV), e.to(DEV, dtype)) for x, t, e in batches(3, B, T, D, seed=13)]
    res = []
    for early in (True, False):
        if not early:
            monkeypatch.setattr(TU, "no_early_opt", True)
        torch.manual_seed(0)
        model = DiffusionTransformer(D, T, d_model=dm, num_heads=4, dim_feedforward=ff, num_layers=3, device=DEV,
                                     compute_dtype=dtype)
        tr = HipTrainer(model, "diffusion", opt, 1e-3)
        losses = []
        for i in range(7):                                   # warm-up (eager) steps, the capture, replays
            tr.step(bs[i % 3])
            losses.append(tr.loss_value())
        torch.cuda.synchronize()
        assert (tr.plan.early_optimizer is not None) == early and tr._early_done == []
        assert int(tr.step_dev.item()) == 7
        res.append((losses, tr.flat.detach().clone(), None if tr.s1 is None else tr.s1.detach().clone(),
                    None if model._shadow is None else model._shadow.detach().clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


def test_bench_data_parallel_launch_sequence_on_one_rank():
    """bench.py end to end in a child process, twice: the single-GPU step (reductions fused into the optimizer, one graph)
    and -- IB_DDP_SELFTEST=1, world size 1 -- the data-parallel sequence (gradient materialised, graph cut around an RCCL
    all-reduce, plain optimizer).  Both must print exactly ONE stdout line (the JSON; RCCL's banner goes to stderr) and,
    because every reduction has a fixed order, end on the bitwise same loss."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for ddp in (False, True):
        env = dict(os.environ)
        env.pop("IB_DDP_SELFTEST", None)
        if ddp:
            env.update(IB_DDP_SELFTEST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1",
                       LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "4",
                            "--no-cpu-baseline", "--no-ddim"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[:500]
        outs.append(json.loads(lines[0]))
    a, b = outs
    # the data-parallel run chose the form of its collectives by the start-up probe (ddp_probe.py: a fresh 1-rank child job
    # that runs both forms side by side); either verdict is fine here, but it must be the probe's and be reported
    assert "collectives" not in a and b["collectives"]["source"] == "probe", b.get("collectives")
    assert b["summary"]["collectives"]["captured_in_step_graph"] == b["collectives"]["captured"]
    # headline = the transformer denoiser (BASELINE configs[2]): bucketed all-reduces under data parallelism; the MLP
    # denoiser rides as `mlp_T50`: one bucket = the whole gradient
    assert a["config"]["grad_buckets"] == 0 and b["config"]["grad_buckets"] >= 2
    assert a["mlp_T50"]["config"]["grad_buckets"] == 0 and b["mlp_T50"]["config"]["grad_buckets"] == 1
    assert "configs[2]" in a["config"]["workload"] and "configs[1]" in a["mlp_T50"]["workload"]
    assert a["mlp_T50"]["final_loss"] == b["mlp_T50"]["final_loss"]
    assert abs(a["final_loss"] - b["final_loss"]) <= 5e-3 * abs(a["final_loss"])      # (other summation order of the slabs)
    for o in outs:
        assert o["unit"] == "windows/s" and o["n_gpus"] == 1 and o["steps"] == 40 and o["value"] > 0
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(o["roofline"])
        assert list(o)[-1] == "summary" and o["summary"]["headline"]["ms_per_step"] == o["ms_per_step"]
        assert o["summary"]["mlp_T50"]["ms_per_step"] == o["mlp_T50"]["ms_per_step"]


def test_bench_two_rank_control_flow_rehearsal():
    """`python bench.py --gpus 2` BARE (no launcher, WORLD_SIZE unset): bench.py starts its own two ranks under
    torch.distributed.run as a fresh child process and relays rank 0's line.  Here on ONE GPU (IB_BENCH_REHEARSAL=1: gloo
    instead of RCCL, both ranks on device 0).  Guards the multi-rank control flow -- every collective (parameter broadcast,
    gradient all-reduce inside every step INCLUDING the recorded eager step of the roofline leg, barriers, the max-reduce
    of the elapsed time) must be entered by every rank, or the run hangs -- and the one-line stdout contract at n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IB_BENCH_REHEARSAL="1")
    for k in ("IB_DDP_SELFTEST", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--no-ddim", "--no-cpu-baseline", "--batches", "16"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:1000]
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["rccl_world"] == 2 and o["backend"] == "gloo"
    assert o["config"]["global_batch"] == 512 and o["config"]["parallelism"] == "dp2"
    assert o["config"]["grad_buckets"] >= 2 and o["value"] > 0 and "roofline" in o and "cpu_baseline" not in o
    assert "configs[3]" in o["config"]["workload"]
    assert o["mlp_T50"]["config"]["parallelism"] == "dp2" and o["mlp_T50"]["config"]["grad_buckets"] == 1
    assert list(o)[-1] == "summary"


def test_nested_branch_fork_is_refused():
    """plans.Branch: a fork issued while another enabled branch's launches are being issued (a fork nested inside a forked
    stream) crashed hipStreamEndCapture in round 1; it is refused with HipError in eager mode already.  Sibling forks and
    disabled (inline) branches nest freely."""
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.plans import Branch
    a, b = Branch(DEV, name="outer"), Branch(DEV, name="inner")
    assert a.on and b.on
    x = torch.zeros(8, device=DEV)
    with pytest.raises(hip.HipError, match="nested"):
        a.run(lambda: b.run(lambda: x.add_(1)))
    assert Branch._depth == 0                       # the guard leaves the class-wide depth balanced
    a.join()
    # siblings: fine
    a.run(lambda: x.add_(1))
    b.run(lambda: x.add_(1))
    a.join()
    b.join()
    # a disabled branch runs inline, also inside an enabled one
    c = Branch(DEV, enabled=False, name="inline")
    a.run(lambda: c.run(lambda: x.add_(1)))
    a.join()
    torch.cuda.synchronize()
    assert Branch._depth == 0


def test_library_streams_are_not_torch_pool_streams():
    """hip.new_stream(): the side branches / capture stream / trainer stream come from ib_stream_create, not from torch's
    round-robin pool of 32 streams per device (which c10d's communication stream shares: the 33rd torch stream of a process
    IS an earlier one); handles of dead wrappers are handed out again, never destroyed"""
    import gc
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.plans import Branch
    pool = {torch.cuda.Stream(device=DEV).cuda_stream for _ in range(40)}
    assert len(pool) <= 32                                  # the pool wraps: torch.cuda.Stream() aliases after 32
    mine = [hip.new_stream(DEV) for _ in range(40)]
    ptrs = [s.cuda_stream for s in mine]
    assert len(set(ptrs)) == 40 and not (set(ptrs) & pool)
    b = Branch(DEV, name="t")
    assert b.stream.cuda_stream not in pool
    x = torch.zeros(8, device=DEV)
    with torch.cuda.stream(mine[0]):
        x.add_(1)
    mine[0].synchronize()
    assert float(x.sum()) == 8.0
    gone = ptrs[-1]
    del mine[-1]
    gc.collect()
    free = list(hip._free_streams.get(torch.cuda.current_device(), []))
    assert gone in free                                     # recycled, not destroyed: a dangling current-stream reference
    assert hip.new_stream(DEV).cuda_stream in free          # stays valid, and the next stream is a recycled one


_CAPTURE_GUARD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from collections import OrderedDict
from inferbiomechanics_amd import hip
from inferbiomechanics_amd.engine import GradBuckets, _Recorder
flat = torch.ones(1024, device=dev)
bk = GradBuckets(flat, OrderedDict(w=(0, 1024)), 1 << 62, active=True)
side, clean = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    bk.launch(0, inline=True)                 # a synchronous c10d collective: its completion event sits on `side`
torch.cuda.synchronize()
assert side.cuda_stream in bk.collective_streams and clean.cuda_stream not in bk.collective_streams
refused = False
with torch.cuda.stream(side):
    try:
        _Recorder(bk.collective_streams).begin()
    except hip.HipError as e:
        refused = "carried collectives" in str(e)
assert refused, "capture on a stream that carried a collective was not refused"
with torch.cuda.stream(clean):                # a stream that never carried one captures fine
    slots = torch.zeros(4, dtype=torch.int64, device=dev)
    rec = _Recorder(bk.collective_streams)
    rec.begin()
    hip.set_ptrs(slots, [flat])
    rec.end()
    torch.cuda.synchronize()
    assert int(slots[0]) == 0                 # the capture executed nothing
    rec.replay()
torch.cuda.synchronize()
assert int(slots[0]) == flat.data_ptr(), int(slots[0])
dist.destroy_process_group()
print("capture-guard ok")
"""


def test_capture_on_a_stream_that_carried_a_collective_is_refused():
    """engine._Recorder.begin(): c10d's watchdog polls the completion events of earlier collectives; a poll that lands while
    the event's stream is being captured aborts the process (round 1).  A 1-rank RCCL collective is issued on a stream, then
    capture on THAT stream must raise HipError; capture on a clean stream works.  Child process: own process group."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", _CAPTURE_GUARD, root], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "capture-guard ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_captured_collectives_on_one_rank_match_the_cut_graphs():
    """IB_GRAPH_COLLECTIVES=1: the gradient all-reduce is captured INTO the step's hipGraph (no graph cut, no host work per
    collective) -- run on the 1-rank RCCL self-test group for the one-bucket policy (MLP denoiser) and the overlapped
    bucket policy (transformer denoiser); the final loss must equal the cut-graph run bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for workload, port in (("mlp_denoiser_T50", "29561"), ("transformer_denoiser_T50", "29563")):
        outs = []
        for captured in (False, True):
            env = dict(os.environ, IB_DDP_SELFTEST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1",
                       LOCAL_RANK="0")
            env["IB_GRAPH_COLLECTIVES"] = "1" if captured else "0"          # forced forms: no start-up probe here
            r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "30",
                                "--warmup", "4", "--no-cpu-baseline", "--no-ddim", "--no-mlp", "--overlap-comm",
                                "on" if "transformer" in workload else "off"],
                               capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0]))
        a, b = outs
        assert a["final_loss"] == b["final_loss"], (workload, a["final_loss"], b["final_loss"])
        assert a["config"]["grad_buckets"] == b["config"]["grad_buckets"] >= 1
        print(workload, "cut graphs", a["ms_per_step"], "ms/step; captured collectives", b["ms_per_step"], "ms/step")


def test_large_batch_kernel_paths_agree():
    """the fused transformer step at M = 6400 token rows through (a) the 256 x 128 NT / TN kernels with the bias partial sums
    (default), (b) the 128 x 128 ring kernels with column-sum launches (IB_NO_NT / IB_NO_TN / IB_NO_WGRAD_BIAS) and (c) the
    data-parallel launch sequence on a 1-rank RCCL group: same losses up to bf16 summation order (5e-3), four steps"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arms = {"default": {}, "ring": {"IB_NO_NT": "1", "IB_NO_TN": "1", "IB_NO_WGRAD_BIAS": "1"},
            "ddp": {"IB_DDP_SELFTEST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "IB_GRAPH_COLLECTIVES": "0"}}
    out = {}
    for name, extra in arms.items():
        env = dict(os.environ)
        for k in ("IB_NO_NT", "IB_NO_TN", "IB_NO_WGRAD_BIAS", "IB_DDP_SELFTEST", "IB_GRAPH_COLLECTIVES"):
            env.pop(k, None)
        env.update(extra)
        if name == "ring":         # IB_NO_TN is a C-level switch: it exists only in the measurement build of the library
            from inferbiomechanics_amd import hip as _hip
            env["IB_HIP_LIB"] = _hip.AB_LIB_PATH
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "path_ab.py")], capture_output=True, text=True, env=env,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    ref = out["ring"]["losses"]
    for name in ("default", "ddp"):
        for a, e in zip(out[name]["losses"], ref):
            assert abs(a - e) <= 5e-3 * abs(e), (name, out[name]["losses"], ref)
        assert abs(out[name]["psum"] - out["ring"]["psum"]) <= 1e-5 * out["ring"]["psum"], name


def test_adopt_stream_same_trajectory_no_handover():
    """a loop that works on the trainer's stream (HipTrainer.adopt_stream(), as cli/train.py and bench.py do) takes the same
    steps as one that calls step() from the default stream -- bitwise the same losses and parameters -- and the values read
    back afterwards on the adopted stream are ordered after the last step"""
    import bench
    from inferbiomechanics_amd.engine import HipTrainer
    dev = torch.device("cuda", 0)
    outs = []
    for adopt in (False, True):
        model = bench.build_model("mlp", 50, 300, torch.bfloat16, dev)
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-3)
        batches = bench.make_batches(4, 32, 50, 300, torch.bfloat16, dev, seed=5)
        prev = tr.adopt_stream() if adopt else None
        if adopt:
            assert torch.cuda.current_stream() == tr.stream and prev is not None
        losses = []
        for i in range(12):
            tr.step(batches[i % 4])
            losses.append(tr.loss_value())
        flat = tr.flat.detach().clone()
        torch.cuda.synchronize()
        if adopt:
            torch.cuda.set_stream(prev)
            assert torch.cuda.current_stream() == prev
        outs.append((losses, flat.cpu()))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
