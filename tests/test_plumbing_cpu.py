"""Host-side plumbing of every launch plan, run on CPU tensors with the C-ABI in DRY-RUN mode
(inferbiomechanics_amd.hip._DryRunLib: arguments are marshalled through the real ctypes signatures and
shape/stride checks run, but nothing is launched and nothing is computed).  This catches glue errors
(wrong arity, shape mismatches between plan steps, missing buffers) without a GPU; numerical parity is
the job of the -m gpu tests."""
import argparse
import os

import pytest
import torch

from inferbiomechanics_amd._tuning import tuning as TU


@pytest.fixture()
def dry():
    from inferbiomechanics_amd import hip
    hip.set_dry_run(True)
    yield hip
    hip.set_dry_run(False)


def targs():
    return argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))


def test_feedforward_plan_plumbing(dry):
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from oracle.fixture_inputs import ff_inputs, ff_labels
    for dt in (torch.float32, torch.bfloat16):
        for actn in ("sigmoid", "silu"):
            m = FeedForwardBaseline(23, 2, 50, 'all_frames', actn, 5, 10, compute_dtype=dt)
            out = m(ff_inputs(3, 10, 23, 5))
            assert out['groundContactWrenchesInRootFrame'].shape == (3, 10, 12)
            ev = RegressionLossEvaluator(None, 'train', device='cpu')
            loss = ev({}, out, ff_labels(3, 10), [], [], targs())
            loss.backward()
            assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    names = dry.lib().calls
    # (few-row batches take the one-launch weight + bias gradient in fp32 too: csrc/gemm_f32_small.hip)
    assert "ib_concat_keys" in names and "ib_linear_wgrad_bias" in names and "ib_regression_loss_strided" in names


def test_transformer_layer_plumbing(dry):
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    for dt in (torch.float32, torch.bfloat16):
        layer = TransformerLayer(64, 4, 128, 0.0, dtype=dt)
        x = torch.randn(2, 9, 64, requires_grad=True)
        y = layer(x)
        y.float().sum().backward()
        assert x.grad.shape == x.shape
        assert all(p.grad is not None for p in layer.parameters())


def test_denoiser_plumbing(dry):
    from inferbiomechanics_amd.loss.DiffusionLossEvaluator import DiffusionLossEvaluator
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    for dt in (torch.float32, torch.bfloat16):
        for model in (DiffusionMLP(30, [32, 48], temb_dim=16, temb_hidden=24, compute_dtype=dt),
                      DiffusionTransformer(30, 7, d_model=32, num_heads=4, dim_feedforward=64, num_layers=2,
                                           temporal_embedding_dim=6, temb_dim=16, temb_hidden=24, compute_dtype=dt)):
            x = torch.randn(3, 7, 30)
            t = torch.tensor([0, 5, 999])
            pred = model(x, t)
            assert pred.shape == x.shape
            loss = DiffusionLossEvaluator()(pred, torch.randn(3, 7, 30))
            loss.backward()
            assert all(p.grad is not None and p.grad.shape == p.shape for p in model.parameters()), \
                [k for k, p in model.named_parameters() if p.grad is None]


def test_cli_train_and_analyze_plumbing(dry, tmp_path, capsys):
    """whole `main.py train` / `analyze` control flow (dataset -> sampler -> fused trainer -> report ->
    checkpoint -> resume -> analyze CSV) with the kernels in dry-run"""
    import os
    from inferbiomechanics_amd.main import main
    ck = str(tmp_path / "ck")
    base = ['--no-wandb', '--synthetic-windows', '24', '--batch-size', '8', '--checkpoint-dir', ck,
            '--data-loading-workers', '0']
    assert main(['train', '--epochs', '2', '--max-steps', '2'] + base)
    files = sorted(os.listdir(os.path.join(ck, 'feedforward')))
    assert files == ['epoch_0_batch_1.pt', 'epoch_1_batch_1.pt']
    sd = torch.load(os.path.join(ck, 'feedforward', files[-1]))
    assert set(sd) == {'epoch', 'model_state_dict', 'optimizer_state_dict'} and sd['epoch'] == 1
    assert list(sd['model_state_dict'])[0] == 'net.0.weight'            # no DDP `module.` prefix
    assert main(['train', '--epochs', '3', '--max-steps', '1', '--eager', '--opt-type', 'adam'] + base[:-4] +
                ['--checkpoint-dir', str(tmp_path / "ck2"), '--data-loading-workers', '0'])
    assert main(['train', '--epochs', '3', '--max-steps', '1'] + base)   # resumes at epoch 2
    assert 'epoch_2_batch_0.pt' in os.listdir(os.path.join(ck, 'feedforward'))
    assert main(['analyze', '--no-wandb', '--synthetic-windows', '5', '--checkpoint-dir', ck,
                 '--data-loading-workers', '0'])
    rows = open(os.path.join(ck, 'feedforward', 'dev_analysis.csv')).read().strip().splitlines()
    assert len(rows) == 5 and rows[0].startswith('synthetic_subject_0,window_')
    assert main(['train', '--model-type', 'diffusion-mlp', '--epochs', '1', '--max-steps', '2', '--feat-dim', '24',
                 '--hidden-dims', '32', '32', '--stride', '1', '--history-len', '6', '--compute-dtype', 'bf16'] + base)
    assert main(['visualize', '--synthetic-windows', '4', '--checkpoint-dir', ck, '--num-frames', '2'])


def test_checkpoint_loader_accepts_ddp_prefix_and_orders_files(dry, tmp_path):
    import os
    from inferbiomechanics_amd.cli.abstract_command import AbstractCommand
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    m = FeedForwardBaseline(23, 2, 50, 'all_frames', 'sigmoid', 5, 10, hidden_dims=[8])
    d = str(tmp_path)
    for e, b, fill in ((0, 999, 1.0), (1, 5, 2.0), (0, 1000, 3.0)):
        sd = {('module.' + k): torch.full_like(v, fill) for k, v in m.state_dict().items()}
        torch.save({'epoch': e, 'model_state_dict': sd, 'optimizer_state_dict': None}, os.path.join(d, f'epoch_{e}_batch_{b}.pt'))
    epoch, batch = AbstractCommand().load_latest_checkpoint(m, checkpoint_dir=d)
    assert (epoch, batch) == (1, 5) and float(m.state_dict()['net.0.bias'][0]) == 2.0
    assert AbstractCommand().load_latest_checkpoint(m, checkpoint_dir=os.path.join(d, 'nope')) == (-1, 0)


def test_chain_step_plumbing(dry):
    """the fused chain path of HipTrainer (prep -> chain -> deferred weight-gradient slabs -> one slab reduction), every
    call marshalled through the real ctypes signatures with the kernels in dry-run"""
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP
    for T in (13, 16):                                   # ragged panels / panel == window
        m = DiffusionMLP(48, [128, 128], temb_dim=32, temb_hidden=128, compute_dtype=torch.bfloat16)
        tr = HipTrainer(m, "diffusion", "adam", 1e-3, use_graph=False)
        assert tr.plan.chain_ok(48)
        x0, eps = torch.randn(5, T, 48), torch.randn(5, T, 48)
        t = torch.randint(0, 1000, (5,))
        dry.lib().calls.clear()
        tr.step((x0, t, eps))
        names = dry.lib().calls
        assert "ib_mlp_chain_train" in names and "ib_mlp_chain_prep" in names
        assert "ib_linear_wgrad_slabs_multi" in names            # every independent dW GEMM: one launch
        assert "ib_q_sample" not in names and "ib_layernorm_fwd" not in names
        assert names[-1] == "ib_optim_step_sources"              # single GPU: the optimizer sums the partials itself


def test_packed_windows_roundtrip_and_reference_pickle_blocks(tmp_path):
    """the packed row reproduces the reference tuple layout exactly; file round trip (memory-mapped); the reference's
    `pickle-data` blocks (torch.save of a list of window tuples, pickle_data.py:52-64) pack to the same rows"""
    import numpy as np
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import (INPUT_KEY_ORDER, LOSS_KEY_ORDER,
                                                                   SyntheticWindowDataset)
    from inferbiomechanics_amd.data.WindowCache import DeviceWindowCache, PackedWindows
    ds = SyntheticWindowDataset(37, history_len=50, stride=5, seed=3)
    pack = PackedWindows.from_windows(ds)
    assert len(pack) == 37 and pack.frames == 10 and pack.x_elems == 1470 and pack.rows.shape[1] == 1472 + 300
    for i in (0, 5, 36):
        inputs, labels, s, t = pack.window(i)
        ri, rl, rs, rt = ds[i]
        assert (s, t) == (rs, rt)
        for k in INPUT_KEY_ORDER:
            assert torch.equal(inputs[k], ri[k])
        for k in LOSS_KEY_ORDER:
            assert torch.equal(labels[k], rl[k])
        # the model-input part of the row is exactly the model's own concatenation
        x = torch.cat([ri[k] for k in INPUT_KEY_ORDER], dim=-1).reshape(-1)
        assert torch.equal(torch.from_numpy(np.array(pack.rows[i][:1470])), x)
    path = str(tmp_path / "w.ibw")
    pack.save(path)
    back = PackedWindows.load(path)
    assert np.array_equal(np.asarray(back.rows), pack.rows) and np.array_equal(back.trials, pack.trials)
    with open(path, "r+b") as f:            # a truncated file is refused
        f.truncate(4096 + 100)
    with pytest.raises(ValueError):
        PackedWindows.load(path)
    blocks = []
    for b, (lo, hi) in enumerate(((0, 20), (20, 37))):
        bp = str(tmp_path / f"train_{b}.pkl")
        torch.save([ds[j] for j in range(lo, hi)], bp)
        blocks.append(bp)
    assert np.array_equal(PackedWindows.from_pickled_blocks(blocks).rows, pack.rows)
    # sampler semantics: DistributedSampler(shuffle=False, drop_last=True), then whole batches only
    cache = DeviceWindowCache(pack, "cpu")
    got = [b.tolist() for b in cache.batches(4, rank=1, world=2)]
    assert got == [[1, 3, 5, 7], [9, 11, 13, 15], [17, 19, 21, 23], [25, 27, 29, 31]]


def test_window_cache_trainer_plumbing(dry):
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    from inferbiomechanics_amd.data.WindowCache import DeviceWindowCache, PackedWindows
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    import argparse
    cache = DeviceWindowCache(PackedWindows.from_windows(SyntheticWindowDataset(16, 50, 5)), "cpu")
    m = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10)
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=[],
                              predict_moment_components=[], predict_wrench_components=[])
    tr = HipTrainer(m, "regression", "rmsprop", 1e-4, args=args, use_graph=False)
    dry.lib().calls.clear()
    tr.step_windows(cache, next(cache.batches(8)))
    names = dry.lib().calls
    assert names[0] == "ib_gather_windows" and "ib_concat_keys" not in names and "ib_regression_loss_strided" in names


def test_groundlink_plumbing_registry_and_state_dict(dry):
    """reference constructor / parameter names (src/models/Groundlink.py:20,34-62), both output formats, autograd path and
    fused-trainer path (dropout keyed on the device step counter)"""
    from inferbiomechanics_amd.cli.abstract_command import AbstractCommand
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    from inferbiomechanics_amd.models.Groundlink import Groundlink
    from oracle.fixture_inputs import gl_inputs
    from oracle.ref_cpu import groundlink_param_shapes
    for fmt, Fo in (("all_frames", 10), ("last_frame", 1)):
        for dt in (torch.float32, torch.bfloat16):
            m = Groundlink(23, 12, 10, output_data_format=fmt, compute_dtype=dt)
            assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == groundlink_param_shapes(23, 12, 10)
            m.train()
            out = m(gl_inputs(3, 10, 23, 10))
            assert out['groundContactWrenchesInRootFrame'].shape == (3, Fo, 12)
            labels = {k: torch.zeros_like(v, dtype=torch.float32) for k, v in out.items()}
            RegressionLossEvaluator(None, 'train', device='cpu')({}, out, labels, [], [], targs()).backward()
            assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    names = dry.lib().calls
    assert "ib_im2col_replicate" in names and "ib_col2im_replicate" in names and "ib_dropout" in names
    m = AbstractCommand().get_model(23, 2, model_type='groundlink', root_history_len=10)
    assert isinstance(m, Groundlink) and m.channels == 23 * 3 + 12 + 36 + 60
    tr = HipTrainer(m, "regression", "adam", 1e-4, args=targs(), use_graph=False)
    dry.lib().calls.clear()
    inputs = gl_inputs(4, 10, 23, 10)
    labels = {k: torch.zeros(4, 10, c) for k, c in zip(
        ('groundContactCenterOfPressureInRootFrame', 'groundContactForceInRootFrame', 'groundContactTorqueInRootFrame',
         'groundContactWrenchesInRootFrame'), (6, 6, 6, 12))}
    tr.step((inputs, labels))
    names = dry.lib().calls
    assert names.count("ib_im2col_replicate") == 4 and names.count("ib_col2im_replicate") == 3
    assert names.count("ib_dropout") == 6 and names[-1].startswith("ib_optim_step")


def test_feedforward_state_dict_keys_follow_the_reference_with_every_flag_combination(golden_dir):
    """`net.{j}` indices shift when --dropout / --batchnorm insert modules (FeedForwardRegressionBaseline.py:67-77): the
    key lists of the real reference class are in the golden file"""
    import numpy as np
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    g = np.load(os.path.join(golden_dir, "ff_options.npz"))
    hip.set_dry_run(True)
    try:
        for bn in (False, True):
            for dr in (False, True):
                m = FeedForwardBaseline(23, 2, 50, "all_frames", "relu", 5, 10, hidden_dims=[64, 48], batchnorm=bn, dropout=dr,
                                        dropout_prob=0.3)
                assert list(m.state_dict().keys()) == list(g[f"keys/bn{int(bn)}_drop{int(dr)}"])
    finally:
        hip.set_dry_run(False)


def test_bench_self_launch_relays_the_ranks_return_code():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (torch.distributed.run as a child process, before
    any GPU call) and relays their return code.  Without a GPU every rank exits with "bench.py needs a GPU": the relayed
    code is non-zero and stdout carries no result line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (on a GPU box tests/test_trainer_gpu.py runs the real two-rank rehearsal)")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr


def test_mask_seed_mixes_torch_seed_rank_and_layer(monkeypatch):
    """plans.mask_seed: torch.manual_seed selects the dropout stream, data-parallel ranks and stacked transformer layers
    draw different masks (the reference's nn.Dropout draws independently per module and per process)"""
    import torch.distributed as dist
    from inferbiomechanics_amd import plans
    torch.manual_seed(7)
    a = plans.mask_seed(0x3A7)
    assert a == plans.mask_seed(0x3A7) and 0 <= a <= 0x7FFFFFF0
    assert a != plans.mask_seed(0x3A8)
    torch.manual_seed(8)
    assert plans.mask_seed(0x3A7) != a
    torch.manual_seed(7)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_rank", lambda: 1)
    assert plans.mask_seed(0x3A7) != a
    monkeypatch.undo()
    l0 = plans.TransformerLayerPlan("a.", 64, 4, 128, torch.bfloat16, "cpu", tag="tl0", dropout_p=0.1)
    l1 = plans.TransformerLayerPlan("b.", 64, 4, 128, torch.bfloat16, "cpu", tag="tl1", dropout_p=0.1)
    assert l0.seed != l1.seed and abs(l0.seed - l1.seed) > 2
    assert plans.TransformerLayerPlan("a.", 64, 4, 128, torch.bfloat16, "cpu", seed=5).seed == 5


def test_wait_for_file_polls_and_times_out(tmp_path):
    import threading
    import time
    from inferbiomechanics_amd.data.WindowCache import wait_for_file
    p = tmp_path / "cache.ibw"
    with pytest.raises(TimeoutError):
        wait_for_file(str(p), timeout_s=0.05, poll_s=0.01)
    threading.Timer(0.1, lambda: p.write_bytes(b"x")).start()
    t0 = time.monotonic()
    wait_for_file(str(p), timeout_s=5.0, poll_s=0.01)
    assert time.monotonic() - t0 < 2.0


def test_cli_diffusion_trains_from_the_motion_cache(dry, tmp_path):
    """`main.py train --model-type diffusion-mlp --window-cache f.npy`: windows written once, loaded into the (here: CPU
    stand-in of the) HBM table, every step = index copy + ONE draw launch + the fused step; no host random numbers"""
    import numpy as np
    from inferbiomechanics_amd.cli.train import TrainCommand
    from inferbiomechanics_amd.main import main
    cache = str(tmp_path / "motion.npy")
    base = ['--no-wandb', '--synthetic-windows', '24', '--batch-size', '8', '--checkpoint-dir', str(tmp_path / "ck"),
            '--data-loading-workers', '0', '--model-type', 'diffusion-mlp', '--feat-dim', '24', '--hidden-dims', '32', '32',
            '--stride', '1', '--history-len', '6', '--compute-dtype', 'bf16', '--seed', '7']
    assert main(['train', '--epochs', '1', '--window-cache', cache, '--loss-every', '2'] + base)
    assert np.load(cache).shape == (24, 6, 24)
    assert TrainCommand.last_run_stats["steps"] == 3 and TrainCommand.last_run_stats["windows_per_s"] > 0
    calls = dry.lib().calls
    assert calls.count("ib_diffusion_draw") >= 3 + 3            # dev evaluation + training steps
    assert main(['train', '--epochs', '2', '--max-steps', '1'] + base)                       # DataLoader x0, device-drawn t / eps
    assert main(['train', '--epochs', '3', '--max-steps', '1', '--eager'] + base)


def test_eager_diffusion_loop_draws_fresh_noise_every_epoch(dry, tmp_path, monkeypatch):
    """`--eager`: the training loop's draw counter is the global step (epoch * batches + i) -- epoch 1 draws with other
    counters than epoch 0 -- while the dev evaluation before every epoch starts at 0 again (its own stream id)"""
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.main import main
    seen = []
    real = hip.diffusion_draw

    def rec(seed, step=0, step_dev=None, stream_id=0, **kw):
        seen.append((int(stream_id), int(step), step_dev is not None))
        return real(seed, step=step, step_dev=step_dev, stream_id=stream_id, **kw)
    monkeypatch.setattr(hip, "diffusion_draw", rec)
    base = ['--no-wandb', '--synthetic-windows', '24', '--batch-size', '8', '--checkpoint-dir', str(tmp_path / "ck"),
            '--data-loading-workers', '0', '--model-type', 'diffusion-mlp', '--feat-dim', '24', '--hidden-dims', '32', '32',
            '--stride', '1', '--history-len', '6', '--compute-dtype', 'bf16', '--seed', '7']
    assert main(['train', '--epochs', '2', '--eager'] + base)
    dev = [s for sid, s, _ in seen if sid & 0x80000000]
    train = [s for sid, s, _ in seen if not sid & 0x80000000]
    assert dev == [0, 1, 2, 0, 1, 2]                               # the same dev noise before every epoch
    assert train == [0, 1, 2, 3, 4, 5]                             # epoch 1 = steps 3..5, not 0..2 again
    assert not any(d for _, _, d in seen)                          # host-side counters only on this path
    # resumed run: the counter starts at the global step of the first epoch it runs, not at 0
    del seen[:]
    assert main(['train', '--epochs', '3', '--eager'] + base)
    assert [s for sid, s, _ in seen if not sid & 0x80000000] == [6, 7, 8]
    with pytest.raises(SystemExit, match="loss-every"):
        main(['train', '--epochs', '1', '--loss-every', '0'] + base)
    with pytest.raises(SystemExit, match="synthetic-windows"):
        main(['train', '--epochs', '1', '--window-cache', 'hbm', '--model-type', 'feedforward', '--no-wandb',
              '--synthetic-windows', '0', '--checkpoint-dir', str(tmp_path / "ck2")])


def test_every_parameter_is_updated_exactly_once_per_step(dry, monkeypatch):
    """one GPU, transformer denoiser: each layer's range of the flat buffers is updated by its own optimizer launch (issued
    from the backward), the step's last launch names those ranges as done (source kind 3) -- together they cover the flat
    buffer exactly once, the early launches use `*step_dev + 1` without a ticket, the last one is the self-counting launch.
    bf16 at the fused launches' row threshold and fp32 at a small shape; IB_NO_EARLY_OPT=1: one launch, nothing marked."""
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionTransformer
    real = hip.optim_step
    for dtype, B, T, D, dm, ff in ((torch.bfloat16, 128, 32, 48, 512, 1024), (torch.float32, 6, 10, 44, 64, 128)):
        for early in (True, False):
            if early:
                monkeypatch.setattr(TU, "no_early_opt", False)
            else:
                monkeypatch.setattr(TU, "no_early_opt", True)
            m = DiffusionTransformer(D, T, d_model=dm, num_heads=4, dim_feedforward=ff, num_layers=3, compute_dtype=dtype)
            tr = HipTrainer(m, "diffusion", "adam", 1e-3, use_graph=False)
            seen = []

            def rec(opt, p, g, s1, s2, lr, step=1, step_dev=None, grad_scale=1.0, shadow=None, ticket=None, sources=None):
                lo = (p.data_ptr() - tr.flat.data_ptr()) // 4
                done = [((v.data_ptr() - tr.grad.data_ptr()) // 4, v.numel()) for v in (sources[4] if sources and len(sources) > 4 else ())]
                srcs = []
                if sources:
                    srcs = [(dw.data_ptr() - tr.grad.data_ptr()) // 4 for _, _, dw in sources[0]]
                    srcs += [(seg[2].data_ptr() - tr.grad.data_ptr()) // 4 for seg in sources[3]]
                seen.append({"lo": lo, "n": p.numel(), "step": step, "ticket": ticket is not None, "done": done, "srcs": srcs,
                             "g_lo": (g.data_ptr() - tr.grad.data_ptr()) // 4, "shadow": shadow is not None})
                return real(opt, p, g, s1, s2, lr, step=step, step_dev=step_dev, grad_scale=grad_scale, shadow=shadow,
                            ticket=ticket, sources=sources)
            monkeypatch.setattr(hip, "optim_step", rec)
            g = torch.Generator().manual_seed(1)
            tr.step((torch.randn(B, T, D, generator=g).to(dtype), torch.randint(0, 1000, (B,), generator=g),
                     torch.randn(B, T, D, generator=g).to(dtype)))
            monkeypatch.setattr(hip, "optim_step", real)
            n = tr.flat.numel()
            last = seen[-1]
            assert last["ticket"] and last["step"] == 0                   # the self-counting launch comes last
            if not early:
                assert len(seen) == 1 and last["done"] == [] and last["lo"] == 0 and last["n"] == n
                continue
            # the output projection (head of the flat buffers) + the two layers whose backward does not run last + the step's
            # last launch, which starts behind their ranges (nothing in the middle left to skip) and covers the last-run
            # layer, the projections' tail and the time-MLP
            assert len(seen) == 4 and last["done"] == [], [(e["lo"], e["n"]) for e in seen]
            cover = torch.zeros(n, dtype=torch.int32)
            for e in seen[:-1]:
                assert not e["ticket"] and e["step"] == 1 and e["g_lo"] == e["lo"] and e["shadow"] == (dtype == torch.bfloat16)
                assert e["srcs"] and all(e["lo"] <= s < e["lo"] + e["n"] for s in e["srcs"])     # only its own gradients
                cover[e["lo"]:e["lo"] + e["n"]] += 1
            assert last["lo"] == max(e["lo"] + e["n"] for e in seen[:-1]) and last["lo"] + last["n"] == n
            assert all(last["lo"] <= s for s in last["srcs"])
            cover[last["lo"]:] += 1
            assert torch.equal(cover, torch.ones(n, dtype=torch.int32))
            early_ranges = [(e["lo"], e["n"]) for e in seen[:-1]]
            lo_o, hi_o = tr._prefix_range("out_proj.")
            assert (lo_o, hi_o - lo_o) in early_ranges and lo_o == 0
            # the ranges are exactly the layers' parameters
            assert tr._prefix_range("transformer_layers.0.")[0] == last["lo"]
            for li in range(1, 3):
                lo, hi = tr._prefix_range(f"transformer_layers.{li}.")
                names = [k for k in tr.layout if k.startswith(f"transformer_layers.{li}.")]
                assert lo == min(tr.layout[k][0] for k in names) and (lo, hi - lo) in early_ranges
                assert all(not (lo <= tr.layout[k][0] < hi) for k in tr.layout if k not in names)


def test_ddp_probe_decision_rules(monkeypatch):
    """ddp_probe.decide(): a forced form (IB_GRAPH_COLLECTIVES=0/1) is taken as is; a host-side backend never captures; a
    probe whose child job fails (here: no GPU -- the child's first HIP call raises) yields the cut-graph form with the
    reason recorded, and the verdict is remembered in the environment for the trainers that follow"""
    from inferbiomechanics_amd import ddp_probe
    monkeypatch.setenv("IB_GRAPH_COLLECTIVES", "1")
    assert ddp_probe.decide(1, 0, "nccl") is True and ddp_probe.verdict()["source"] == "environment"
    monkeypatch.setenv("IB_GRAPH_COLLECTIVES", "0")
    assert ddp_probe.decide(1, 0, "nccl") is False
    monkeypatch.delenv("IB_GRAPH_COLLECTIVES")
    monkeypatch.setattr(ddp_probe, "_verdict", None)
    # c10d's event cache on (the default of a process group the caller created without prepare_env()): never captured
    monkeypatch.delenv("TORCH_NCCL_CUDA_EVENT_CACHE", raising=False)
    assert ddp_probe.decide(1, 0, "nccl") is False and ddp_probe.verdict()["source"] == "event-cache"
    monkeypatch.delenv("IB_GRAPH_COLLECTIVES")
    monkeypatch.setenv("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
    assert ddp_probe.decide(1, 0, "gloo") is False and ddp_probe.verdict()["source"] == "backend"
    assert os.environ["IB_GRAPH_COLLECTIVES"] == "0"
    monkeypatch.delenv("IB_GRAPH_COLLECTIVES")
    if torch.cuda.is_available():
        return                                   # (on a GPU box the real probe is exercised by tests/test_trainer_gpu.py)
    r = ddp_probe.run_child(1, timeout_s=120)
    assert r["ok"] is False and r["why"] and r["seconds"] >= 0
    calls = []
    monkeypatch.setattr(ddp_probe, "run_child", lambda world: (calls.append(world), {"ok": True, "why": "stub", "seconds": 0.0})[1])
    assert ddp_probe.decide(1, 0, "nccl") is True and calls == [1] and ddp_probe.verdict()["source"] == "probe"
    assert os.environ["IB_GRAPH_COLLECTIVES"] == "1"
    assert ddp_probe.decide(1, 0, "nccl") is True and calls == [1]          # remembered: no second probe
    monkeypatch.delenv("IB_GRAPH_COLLECTIVES")


def test_attention_inside_the_layer_launches_is_chosen_by_shape(dry):
    """plans.TransformerLayerPlan.attn_T: one-window panels (the attention inside the fused layer launches) for bf16,
    d = 512 = 8 heads x 64, 16 <= T <= 64, whole windows, at least 4096 rows -- and not where one-window panels would be much
    shorter than the token-count panels (many short windows multiply the rounds of workgroups)"""
    from inferbiomechanics_amd import plans
    from inferbiomechanics_amd._tuning import tuning as TU
    lp = plans.TransformerLayerPlan("l.", 512, 8, 2048, torch.bfloat16, "cpu", tag="tlx")
    assert lp.attn_T(12800, 50) == 50 and lp.attn_T(4096, 32) == 32 and lp.attn_T(25600, 50) == 50 and lp.attn_T(8192, 64) == 64
    assert lp.attn_T(32768, 16) == 0            # 2048 workgroups instead of 512
    assert lp.attn_T(51200, 200) == 0 and lp.attn_T(4000, 50) == 0 and lp.attn_T(12800, 48) == 0      # T > 64; < 4096 rows; M % T
    assert plans.TransformerLayerPlan("l.", 512, 4, 2048, torch.bfloat16, "cpu").attn_T(12800, 50) == 0          # 4 heads of 128
    assert plans.TransformerLayerPlan("l.", 512, 8, 2048, torch.float32, "cpu").attn_T(12800, 50) == 0           # parity mode
    assert plans.TransformerLayerPlan("l.", 512, 8, 2048, torch.bfloat16, "cpu", dropout_p=0.1).attn_T(12800, 50) == 0
    TU.no_attn_fuse = True
    try:
        assert lp.attn_T(12800, 50) == 0
    finally:
        TU.no_attn_fuse = False


def test_watchdog_drain_waits_for_the_flight_recorders_retired_flags(monkeypatch):
    """engine._drain_c10d_watchdog: before a capture with collectives it polls c10d's flight recorder until EVERY recorded
    Work is `retired` (the watchdog thread has let go of it: `state == completed` is not enough), gives up loudly, and
    falls back to a timed wait when the recorder is off"""
    import pickle
    import time
    from inferbiomechanics_amd import engine, hip
    c = torch._C._distributed_c10d
    calls = {"n": 0}

    def entries(retired_after):
        def dump(include_collectives=None, include_stack=None, only_active=None):
            calls["n"] += 1
            done = calls["n"] > retired_after
            return pickle.dumps({"entries": [{"record_id": 0, "state": "completed", "retired": True},
                                             {"record_id": 1, "state": "completed", "retired": done}]})
        return dump
    monkeypatch.setattr(c, "_dump_nccl_trace", entries(3), raising=False)
    engine._drain_c10d_watchdog("cpu")
    assert engine._drain_report["mode"] == "flight-recorder" and engine._drain_report["polls"] == 4 and calls["n"] == 4
    # a Work the watchdog never lets go of: an error that names the way out, not a capture beside it
    calls["n"] = 0
    monkeypatch.setattr(c, "_dump_nccl_trace", entries(10 ** 9), raising=False)
    with pytest.raises(hip.HipError, match="IB_GRAPH_COLLECTIVES=0"):
        engine._drain_c10d_watchdog("cpu", timeout_s=0.05)
    # recorder off (no entries) / a torch without the `retired` field: several watchdog periods of plain waiting
    slept = []
    monkeypatch.setattr(time, "sleep", lambda s: slept.append(s))
    monkeypatch.setattr(c, "_dump_nccl_trace", lambda *a: pickle.dumps({"entries": []}), raising=False)
    engine._drain_c10d_watchdog("cpu")
    assert engine._drain_report["mode"] == "sleep" and slept and max(slept) >= 0.5
    monkeypatch.setattr(c, "_dump_nccl_trace", lambda *a: pickle.dumps({"entries": [{"record_id": 0, "state": "completed"}]}),
                        raising=False)
    engine._drain_c10d_watchdog("cpu")
    assert engine._drain_report["mode"] == "sleep"


def test_sampler_row_split_rule():
    """plans.DenoiserTransformerPlan.side_windows: the windows beyond the fused launch's last FULL round of 256 panels take
    the per-op side stack when that round would fill at most 96 CUs (frozen bf16 weights, d = 512, >= 8193 rows)"""
    from inferbiomechanics_amd import plans

    class _On:
        on = True
    p = plans.DenoiserTransformerPlan.__new__(plans.DenoiserTransformerPlan)
    p.inference, p.dtype, p.d, p.br_side = True, torch.bfloat16, 512, _On()

    class _L:
        infer_packed = True
    p.layers = [_L(), _L()]
    assert p.side_windows(256, 200) == 11          # 800 panels = 3 rounds + 32: windows 245 ... 255
    assert p.side_windows(90, 200) == 9 and p.side_windows(84, 200) == 3
    assert p.side_windows(128, 200) == 0           # 400 panels: the second round is 144 panels wide
    assert p.side_windows(120, 200) == 0           # 119 panels: measured -3.4 %
    assert p.side_windows(110, 200) == 29          # 88 panels: measured +4.7 %
    assert p.side_windows(256, 50) == 0 and p.side_windows(16, 200) == 0           # one round / below the fused launch
    p.inference = False
    assert p.side_windows(256, 200) == 0
    p.inference, p.dtype = True, torch.float32
    assert p.side_windows(256, 200) == 0
