"""SURVEY.md §8f rank 4 on the GPU: a checkpoint in the REFERENCE's grammar (written by oracle/make_golden.py from the real
reference class + torch.optim.RMSprop: `module.`-prefixed keys, torch.optim state dict; train.py:272-278) is loaded through
`load_latest_checkpoint` (abstract_command.py:86-120), `main.py analyze` runs on it and reports what the reference's own
model + loss evaluator report on the same synthetic windows, and a resumed fused training step continues the reference's
optimizer trajectory.  -m gpu."""
import os
import re
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture()
def ckpt_dir(tmp_path, golden_dir):
    dst = tmp_path / "ck"
    shutil.copytree(os.path.join(golden_dir, "ref_checkpoint"), dst)
    return str(dst)


def test_analyze_on_a_reference_checkpoint_reports_the_reference_numbers(ckpt_dir, golden_dir, capsys):
    from inferbiomechanics_amd.main import main
    g = np.load(os.path.join(golden_dir, "ckpt_analyze.npz"))
    assert main(['analyze', '--no-wandb', '--synthetic-windows', '6', '--checkpoint-dir', ckpt_dir, '--hidden-dims', '32', '24',
                 '--data-loading-workers', '0'])
    text = capsys.readouterr().out
    assert "Loaded checkpoint from epoch 3, batch 7" in text
    names = ["Force Avg Err", "COM Acc Avg Err", "CoP Avg Err", "Moment Avg Err", "Wrench Avg Err", "Wrench Moment Avg Err"]
    for split in ("dev", "train"):
        block = text.split(f"Final {split} results:")[1]
        got = [float(re.search(re.escape(n) + r": ([-+0-9.eE]+)", block).group(1)) for n in names]
        np.testing.assert_allclose(got, g[f"{split}/metrics"], rtol=1e-4, err_msg=split)
        rows = open(os.path.join(ckpt_dir, "feedforward", f"{split}_analysis.csv")).read().strip().splitlines()
        assert rows == [f"synthetic_subject_0,window_{i}" for i in range(6)]       # sub_name, trial_name (analyze.py:166-174)


def test_per_window_losses_after_loading_match_the_reference(ckpt_dir, golden_dir):
    import argparse
    from inferbiomechanics_amd.cli.abstract_command import AbstractCommand
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    g = np.load(os.path.join(golden_dir, "ckpt_analyze.npz"))
    cmd = AbstractCommand()
    model = cmd.get_model(23, 2, 'feedforward', history_len=50, stride=5, hidden_dims=[32, 24], activation='sigmoid',
                          device='cpu')                      # the reference default: resolves to the local GPU
    assert next(model.parameters()).is_cuda
    assert cmd.load_latest_checkpoint(model, checkpoint_dir=os.path.join(ckpt_dir, "feedforward")) == (3, 7)
    model.eval()
    a1 = argparse.Namespace(predict_grf_components=[1], predict_cop_components=[], predict_moment_components=[],
                            predict_wrench_components=[])
    collate = torch.utils.data.default_collate
    for split, seed in (("dev", 1), ("train", 0)):
        ds = SyntheticWindowDataset(6, 50, 5, seed=seed)
        ev = RegressionLossEvaluator(None, split, device=DEV)
        with torch.no_grad():
            got = [float(ev({}, model(collate([ds[i]])[0]), collate([ds[i]])[1], [0], [i], a1)) for i in range(6)]
        np.testing.assert_allclose(got, g[f"{split}/losses"], rtol=1e-4)


def test_resuming_the_fused_trainer_from_a_reference_checkpoint_continues_its_trajectory(ckpt_dir, golden_dir):
    import argparse
    from inferbiomechanics_amd.cli.abstract_command import AbstractCommand
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    from inferbiomechanics_amd.engine import HipTrainer
    g = np.load(os.path.join(golden_dir, "ckpt_analyze.npz"))
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    cmd = AbstractCommand()
    model = cmd.get_model(23, 2, 'feedforward', history_len=50, stride=5, hidden_dims=[32, 24], activation='sigmoid', device='gpu')
    tr = HipTrainer(model, "regression", "rmsprop", 1e-4, args=args)
    d = os.path.join(ckpt_dir, "feedforward")
    assert cmd.load_latest_checkpoint(model, optimizer=tr, checkpoint_dir=d) == (3, 7)
    ck = torch.load(os.path.join(d, "epoch_3_batch_7.pt"), map_location="cpu")
    osd = ck["optimizer_state_dict"]
    # the torch.optim state landed in the flat buffers (square_avg -> s1, step -> the device counter)
    for i, (k, p) in enumerate(model.named_parameters()):
        off, n = tr.layout[k]
        assert torch.equal(tr.s1[off:off + n].view(p.shape).cpu(), osd["state"][i]["square_avg"]), k
    assert int(tr.step_dev.cpu()) == 2 and tr.steps_done == 2
    inputs, labels, _, _ = torch.utils.data.default_collate([SyntheticWindowDataset(8, 50, 5, seed=0)[i] for i in range(4)])
    tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
    np.testing.assert_allclose(tr.loss_value(), float(g["resume/loss"]), rtol=1e-4)
    for k, p in model.named_parameters():
        e = g["resume/param/" + k]
        np.testing.assert_allclose(p.detach().reshape(-1)[:64].cpu().numpy(), e, rtol=2e-4, atol=2e-6, err_msg=k)
    # and back: the trainer's state in torch.optim grammar loads into torch.optim.RMSprop
    tsd = tr.torch_optimizer_state_dict()
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-4)
    opt.load_state_dict(tsd)
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0
    # ... and torch.optim STEPS from it (load_state_dict replaces the live param_groups: they must carry alpha / eps / ...,
    # not only lr) exactly as the fused trainer's next step does: same gradient, same parameters before
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    tr.step(({k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in labels.items()}))
    torch.cuda.synchronize()
    q = {k: before[k].clone().requires_grad_(True) for k in before}
    for k, p in model.named_parameters():
        q[k].grad = p.grad.detach().clone()              # the gradient step 4 used (taken at `before`)
    opt2 = torch.optim.RMSprop(list(q.values()), lr=1e-4)
    opt2.load_state_dict(tsd)
    opt2.step()
    for k, p in model.named_parameters():
        upd = (p.detach() - before[k]).abs().max().item()
        err = (p.detach() - q[k].detach()).abs().max().item()
        assert err <= 1e-3 * upd + 1e-9, (k, err, upd)
