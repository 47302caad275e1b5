"""Generate golden vectors from the REAL reference classes (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Imports /root/reference/src with `nimblephysics` and `wandb` replaced by MagicMock (they are
imported at module top by the reference but are not used by model / loss arithmetic; neither is
installed here and there is no network).  Writes small .npz fixtures (data only: inputs are
closed-form `det_fill` tensors, so only expected outputs are stored) into tests/golden/.
The reference never travels to the GPU box; these fixtures do.

TEST INFRASTRUCTURE ONLY.
"""
import argparse
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.ref_cpu import K_COP, K_FORCE, K_TORQUE, K_WRENCH, det_fill  # noqa: E402
from oracle.fixture_inputs import (FF_CASES, FF_OPT_B, FF_OPT_CASES, FF_OPT_HIDDEN, FF_OPT_P, GL_CASES,  # noqa: E402
                                   LOSS_SUBSETS, TL_CASES, det_state, ff_inputs, ff_labels, ff_opt_state, gl_inputs,
                                   loss_case_outputs)

REF = os.environ.get("IB_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.modules["nimblephysics"] = MagicMock()
    sys.modules["wandb"] = MagicMock()
    sys.path.insert(0, os.path.join(REF, "src"))
    from models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from models.TransformerBaseline import TransformerLayer
    from loss.RegressionLossEvaluator import RegressionLossEvaluator
    return FeedForwardBaseline, TransformerLayer, RegressionLossEvaluator


def np_(t):
    return t.detach().cpu().numpy().copy()


def load_det_state(module, seed0=1.0):
    sd = module.state_dict()
    new = det_state({k: tuple(v.shape) for k, v in sd.items()}, seed0)
    new = {k: v.to(sd[k].dtype) for k, v in new.items()}
    module.load_state_dict(new)
    return new


def train_args(grf=range(6), cop=range(6), moment=range(6), wrench=range(12)):
    a = argparse.Namespace()
    a.predict_grf_components = list(grf)
    a.predict_cop_components = list(cop)
    a.predict_moment_components = list(moment)
    a.predict_wrench_components = list(wrench)
    return a


def gen_feedforward(FF, RLE):
    B, dofs, ncb = 4, 23, 2
    for name, hist, stride, actn in FF_CASES:
        F = hist // stride
        model = FF(dofs, ncb, hist, "all_frames", actn, stride, 10, hidden_dims=[512, 512])
        load_det_state(model)
        inputs = ff_inputs(B, F, dofs, stride)
        labels = ff_labels(B, F)
        out = model({k: v.clone() for k, v in inputs.items()})
        ev = RLE(dataset=None, split="train")
        loss = ev({}, dict(out), {k: v.clone() for k, v in labels.items()}, [], [], train_args())
        loss.backward()
        d = {"meta_torch": np.array(torch.__version__), "loss": np_(loss)}
        for k, v in out.items():
            d["out/" + k] = np_(v)
        d["metrics"] = np.array([ev.force_reported_metrics[0], ev.moment_reported_metrics[0],
                                 ev.cop_reported_metrics[0], ev.wrench_reported_metrics[0],
                                 ev.wrench_moment_reported_metrics[0], ev.com_acc_reported_metrics[0]])
        for k, p in model.named_parameters():
            d["gnorm/" + k] = np_(p.grad.norm())
            d["gslice/" + k] = np_(p.grad.reshape(-1)[:64])
        # one optimizer step of each kind the CLI offers (train.py:183-194), lr 1e-4 default (train.py:41)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        for opt_name, ctor in [("rmsprop", torch.optim.RMSprop), ("adam", torch.optim.Adam),
                               ("sgd", torch.optim.SGD)]:
            model.load_state_dict(sd0)
            for k, p in model.named_parameters():
                p.grad = grads[k].clone()
            opt = ctor(model.parameters(), lr=1e-4)
            opt.step()
            for k, p in model.named_parameters():
                d[f"step_{opt_name}/" + k] = np_(p.reshape(-1)[:64])
        np.savez_compressed(os.path.join(OUT, f"ff_{name}.npz"), **d)
        print("ff", name, float(loss.detach()))


def gen_feedforward_options(FF, RLE):
    """the optional layers of the feedforward model (FeedForwardRegressionBaseline.py:68-72; flags --batchnorm / --dropout
    --dropout-prob, train.py:43-47): state-dict key grammar for every flag combination, BatchNorm in train mode (batch
    statistics, running-statistics update -- deterministic) and eval mode, Dropout in eval mode (identity; its train-mode
    masks come from torch's global generator and cannot be pinned)."""
    dofs, ncb, hist, stride, B = 23, 2, 50, 5, FF_OPT_B
    F = hist // stride
    d = {"meta_torch": np.array(torch.__version__)}
    for bn in (False, True):
        for dr in (False, True):
            m = FF(dofs, ncb, hist, "all_frames", "relu", stride, 10, hidden_dims=list(FF_OPT_HIDDEN), batchnorm=bn,
                   dropout=dr, dropout_prob=FF_OPT_P)
            d[f"keys/bn{int(bn)}_drop{int(dr)}"] = np.array(list(m.state_dict().keys()))
    for name, bn, dr, train in FF_OPT_CASES:
        model = FF(dofs, ncb, hist, "all_frames", "relu", stride, 10, hidden_dims=list(FF_OPT_HIDDEN), batchnorm=bn,
                   dropout=dr, dropout_prob=FF_OPT_P)
        sd = model.state_dict()
        new = ff_opt_state({k: tuple(v.shape) for k, v in sd.items()})
        model.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})
        model.train(train)
        inputs, labels = ff_inputs(B, F, dofs, stride), ff_labels(B, F)
        out = model({k: v.clone() for k, v in inputs.items()})
        ev = RLE(dataset=None, split="train")
        loss = ev({}, dict(out), {k: v.clone() for k, v in labels.items()}, [], [], train_args())
        loss.backward()
        d[f"{name}/loss"] = np_(loss)
        for k, v in out.items():
            d[f"{name}/out/{k}"] = np_(v)
        for k, p in model.named_parameters():
            if p.dim() == 1:
                d[f"{name}/grad/{k}"] = np_(p.grad)                 # biases, BatchNorm gamma / beta: whole
            else:
                d[f"{name}/gnorm/{k}"] = np_(p.grad.norm())
                d[f"{name}/gslice/{k}"] = np_(p.grad.reshape(-1)[:64])
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                d[f"{name}/after/{k}"] = np_(v)
        print("ff options", name, float(loss.detach()))
    np.savez_compressed(os.path.join(OUT, "ff_options.npz"), **d)


def gen_transformer_layer(TL):
    for name, d, h, ffn, B, T, dt in TL_CASES:
        layer = TL(d, h, ffn, 0.0, dtype=dt)
        load_det_state(layer)
        x = det_fill((B, T, d), 7, 1.0, dt).requires_grad_(True)
        wout = det_fill((B, T, d), 8, 1.0, dt)
        y = layer(x)
        (y * wout).sum().backward()
        dd = {"meta_torch": np.array(torch.__version__),
              "y_sub": np_(y[:, ::7, ::5]), "y_sum": np_(y.sum()), "y_sq": np_((y * y).sum()),
              "dx_sub": np_(x.grad[:, ::7, ::5]), "dx_norm": np_(x.grad.norm())}
        if d <= 128:
            dd["y_full"] = np_(y)
            dd["dx_full"] = np_(x.grad)
        for k, p in layer.named_parameters():
            dd["gnorm/" + k] = np_(p.grad.norm())
            dd["gslice/" + k] = np_(p.grad.reshape(-1)[:64])
        np.savez_compressed(os.path.join(OUT, f"tl_{name}.npz"), **dd)
        print("tl", name, float(y.sum()))


def gen_transformer_layer_dropout(TL):
    """TransformerLayer(dropout=0.25) in TRAIN mode (TransformerBaseline.py:8-38): torch's generator draws the masks, hooks
    record them -- nn.MultiheadAttention is asked for its per-head weights (they are returned AFTER its dropout, so
    weights / eval-mode probabilities = the multipliers), dropout1 / dropout2 by output / input.  Saved: input, masks, output,
    all gradients.  Pins WHERE the three dropouts act (oracle/ref_cpu.py::transformer_layer_forward(masks=...))."""
    d, h, ffn, B, T, p = 32, 4, 64, 2, 9, 0.25
    layer = TL(d, h, ffn, p, dtype=torch.float64)
    load_det_state(layer)
    layer.train()
    rec = {}
    layer.multihead_attention.register_forward_pre_hook(
        lambda m, a, kw: (a, dict(kw, average_attn_weights=False)), with_kwargs=True)
    layer.multihead_attention.register_forward_hook(lambda m, a, out: rec.__setitem__("w", out[1].detach().clone()))
    for nm in ("dropout1", "dropout2"):
        getattr(layer, nm).register_forward_hook(
            lambda m, a, out, nm=nm: rec.__setitem__(nm, (a[0].detach().clone(), out.detach().clone())))
    torch.manual_seed(1234)
    x = det_fill((B, T, d), 7, 1.0, torch.float64).requires_grad_(True)
    wout = det_fill((B, T, d), 8, 1.0, torch.float64)
    y = layer(x)
    (y * wout).sum().backward()
    # eval-mode probabilities of the same input (no dropout) -> the attention multipliers
    layer.eval()
    with torch.no_grad():
        layer(x)
    probs = rec["w"].clone()                       # eval pass overwrote rec["w"] with the undropped probabilities
    layer.train()
    torch.manual_seed(1234)
    layer(x)                                       # same seed: the train pass's dropped weights again
    attn_mult = torch.where(probs > 0, rec["w"] / probs, torch.zeros_like(probs))
    keep = 1.0 / (1.0 - p)
    assert ((attn_mult - 0).abs() < 1e-9).logical_or((attn_mult - keep).abs() < 1e-9).all()
    attn_mult = torch.where(attn_mult > keep / 2, torch.full_like(probs, keep), torch.zeros_like(probs))   # exact values
    dd = {"meta_torch": np.array(torch.__version__), "p": np.array(p), "x": np_(x), "wout": np_(wout), "y": np_(y),
          "dx": np_(x.grad), "mask/attn": np_(attn_mult)}
    for nm, key in (("dropout1", "drop1"), ("dropout2", "drop2")):
        i, o = rec[nm]
        m = torch.where(o != 0, torch.full_like(o, keep), torch.zeros_like(o))
        assert torch.allclose(i * m, o, atol=1e-12)
        dd["mask/" + key] = np_(m)
    for k, q in layer.named_parameters():
        dd["param/" + k] = np_(q)
        dd["grad/" + k] = np_(q.grad)
    np.savez_compressed(os.path.join(OUT, "tl_dropout_train.npz"), **dd)
    print("tl dropout train", float(y.sum()), float(attn_mult.mean()))


def gen_groundlink(RLE):
    """Groundlink (src/models/Groundlink.py:19-156) in eval mode (its fc Dropout(0.2) draws from torch's generator in
    train mode and cannot be pinned); outputs, loss through the reference evaluator, gradient norms + slices."""
    from models.Groundlink import Groundlink
    B = 3
    for name, fmt, F in GL_CASES:
        model = Groundlink(23, 12, 10, fmt)
        model.eval()
        load_det_state(model, seed0=5.0)
        inputs = gl_inputs(B, F)
        Fo = F if fmt == "all_frames" else 1
        labels = ff_labels(B, Fo)
        out = model({k: v.clone() for k, v in inputs.items()})
        ev = RLE(dataset=None, split="train")
        loss = ev({}, dict(out), {k: v.clone() for k, v in labels.items()}, [], [], train_args())
        loss.backward()
        d = {"meta_torch": np.array(torch.__version__), "loss": np_(loss)}
        for k, v in out.items():
            d["out/" + k] = np_(v)
        for k, q in model.named_parameters():
            g = q.grad
            d["gnorm/" + k] = np_(g.norm())
            d["gslice/" + k] = np_(g.reshape(-1)[:64])
        np.savez_compressed(os.path.join(OUT, f"gl_{name}.npz"), **d)
        print("groundlink", name, float(loss))


def gen_loss(RLE):
    B, F = 5, 7
    outs = loss_case_outputs(B, F)
    labels = ff_labels(B, F)
    d = {"meta_torch": np.array(torch.__version__)}
    subsets = {k: train_args(*v) for k, v in LOSS_SUBSETS.items()}
    for name, a in subsets.items():
        ev = RLE(dataset=None, split="dev")
        o = {k: v.clone().requires_grad_(True) for k, v in outs.items()}
        loss = ev({}, dict(o), {k: v.clone() for k, v in labels.items()}, [], [], a)
        loss.backward()
        d[f"{name}/loss"] = np_(loss)
        d[f"{name}/force"] = np_(ev.force_losses[0])
        d[f"{name}/moment"] = np_(ev.moment_losses[0])
        d[f"{name}/wrench"] = np_(ev.wrench_losses[0])
        d[f"{name}/cop"] = np_(ev.cop_losses[0])
        d[f"{name}/metrics"] = np.array([ev.force_reported_metrics[0], ev.moment_reported_metrics[0],
                                         ev.cop_reported_metrics[0], ev.wrench_reported_metrics[0],
                                         ev.wrench_moment_reported_metrics[0], ev.com_acc_reported_metrics[0]])
        for k, v in o.items():
            d[f"{name}/grad/{k}"] = np_(v.grad if v.grad is not None else torch.zeros_like(v))
    d["mask"] = np_(RLE.get_mask_by_threes(labels[K_FORCE], threshold=10.0))
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **d)
    print("loss ok")


def gen_checkpoint_analyze(FF, RLE):
    """a checkpoint in the REFERENCE's grammar (train.py:272-278: {'epoch', 'model_state_dict' with
    DDP's `module.` prefix, 'optimizer_state_dict' = torch.optim state}) written from the real class after two RMSprop
    steps, what `analyze` (analyze.py:139-178,197-238: batch 1, eval mode, --predict-grf-components [1]) reports for it
    on synthetic windows, and the NEXT training step after a resume (exercises the imported optimizer state)."""
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset
    hidden, lr = [32, 24], 1e-4
    model = FF(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=hidden)
    load_det_state(model, seed0=9.0)
    train_ds = SyntheticWindowDataset(8, 50, 5, seed=0)
    collate = torch.utils.data.default_collate

    def batch(ds, idx):
        inputs, labels, _, _ = collate([ds[i] for i in idx])
        return inputs, labels
    opt = torch.optim.RMSprop(model.parameters(), lr=lr)
    model.train()
    for step in range(2):
        inputs, labels = batch(train_ds, range(4 * step, 4 * step + 4))
        opt.zero_grad()
        loss = RLE(dataset=None, split="train")({}, model(inputs), labels, [], [], train_args())
        loss.backward()
        opt.step()
    ck = os.path.join(OUT, "ref_checkpoint", "feedforward")
    os.makedirs(ck, exist_ok=True)
    torch.save({"epoch": 3, "model_state_dict": {"module." + k: v.clone() for k, v in model.state_dict().items()},
                "optimizer_state_dict": opt.state_dict()}, os.path.join(ck, "epoch_3_batch_7.pt"))
    d = {"meta_torch": np.array(torch.__version__), "hidden": np.array(hidden)}
    a1 = train_args([1], [], [], [])                    # analyze defaults (analyze.py:44-47)
    model.eval()
    for split, seed in (("dev", 1), ("train", 0)):
        ds = SyntheticWindowDataset(6, 50, 5, seed=seed)
        ev = RLE(dataset=None, split=split)
        losses = []
        with torch.no_grad():
            for i in range(6):
                inputs, labels = batch(ds, [i])
                losses.append(float(ev({}, model(inputs), labels, [0], [i], a1)))
        d[f"{split}/losses"] = np.array(losses)
        d[f"{split}/metrics"] = np.array([np.mean(ev.force_reported_metrics), np.mean(ev.com_acc_reported_metrics),
                                          np.mean(ev.cop_reported_metrics), np.mean(ev.moment_reported_metrics),
                                          np.mean(ev.wrench_reported_metrics), np.mean(ev.wrench_moment_reported_metrics)])
    model.train()
    inputs, labels = batch(train_ds, range(4))
    opt.zero_grad()
    loss = RLE(dataset=None, split="train")({}, model(inputs), labels, [], [], train_args())
    loss.backward()
    opt.step()
    d["resume/loss"] = np_(loss)
    for k, p in model.named_parameters():
        d["resume/param/" + k] = np_(p.reshape(-1)[:64])
    np.savez_compressed(os.path.join(OUT, "ckpt_analyze.npz"), **d)
    print("checkpoint + analyze ok", float(loss.detach()))


def gen_optim():
    d = {"meta_torch": np.array(torch.__version__)}
    ctors = {"sgd": torch.optim.SGD, "adam": torch.optim.Adam, "rmsprop": torch.optim.RMSprop,
             "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax}
    for name, ctor in ctors.items():
        p = torch.nn.Parameter(det_fill((257,), 3, 0.5, torch.float64))
        opt = ctor([p], lr=1e-2)
        traj = []
        for s in range(4):
            p.grad = det_fill((257,), 20 + s, 0.3 * (s + 1), torch.float64)
            opt.step()
            traj.append(np_(p).copy())
        d[name] = np.stack(traj)
    np.savez_compressed(os.path.join(OUT, "optim_traj.npz"), **d)
    print("optim ok")


def gen_loader():
    """the REAL reference window loader (src/data/AddBiomechanicsDataset.py:63-139 index, :161-285 __getitem__,
    :287-303 worker re-open) over oracle/fake_nimble.py's closed-form subjects: window index, contact-body order,
    `--short` slice and a spread of whole windows (10 input + 7 label tensors) per case"""
    import tempfile
    from oracle import fake_nimble
    from oracle.fixture_inputs import LOADER_CASES, loader_sample
    import data.AddBiomechanicsDataset as refds          # already imported (with the MagicMock) by the model modules
    refds.nimble = fake_nimble                           # the module-level name every nimble call goes through
    d = {"meta_torch": np.array(torch.__version__)}
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "train")
        paths = fake_nimble.make_tree(root)
        for name, window, stride, fmt, dt in LOADER_CASES:
            ds = refds.AddBiomechanicsDataset(root, window, None, dtype=getattr(torch, dt), stride=stride,
                                              output_data_format=fmt, skip_loading_skeletons=(name != LOADER_CASES[0][0]))
            assert ds.subject_paths == paths
            d[f"{name}/windows"] = np.array(ds.windows, dtype=np.int64).reshape(-1, 3)
            d[f"{name}/contact_bodies"] = np.array(ds.contact_bodies)
            d[f"{name}/num_dofs"] = np.array(ds.num_dofs)
            for i in loader_sample(len(ds)):
                inputs, labels, subj, trial = ds[i]
                d[f"{name}/{i}/meta"] = np.array([subj, trial])
                for k, v in inputs.items():
                    d[f"{name}/{i}/in/{k}"] = np_(v)
                for k, v in labels.items():
                    d[f"{name}/{i}/lab/{k}"] = np_(v)
            print("loader", name, len(ds), "windows")
        # a single-file data path, and the --short slice over a 13-subject tree
        one = refds.AddBiomechanicsDataset(paths[1], 50, None, stride=5, output_data_format="all_frames",
                                           skip_loading_skeletons=True)
        d["single/windows"] = np.array(one.windows, dtype=np.int64).reshape(-1, 3)
        d["single/contact_bodies"] = np.array(one.contact_bodies)
        inputs, labels, _, _ = one[7]
        d["single/7/force"] = np_(labels[K_FORCE])
        d["single/7/wrench"] = np_(labels[K_WRENCH])
        root2 = os.path.join(tmp, "short")
        names = ["alpha"] + [f"tiny{j:02d}" for j in range(12)]
        paths2 = fake_nimble.make_tree(root2, names)
        short = refds.AddBiomechanicsDataset(root2, 50, None, stride=5, output_data_format="last_frame",
                                             testing_with_short_dataset=True, skip_loading_skeletons=True)
        assert short.subject_paths == paths2[11:12]
        d["short/windows"] = np.array(short.windows, dtype=np.int64).reshape(-1, 3)
        d["short/subject"] = np.array(os.path.basename(short.subject_paths[0]))
    np.savez_compressed(os.path.join(OUT, "loader_windows.npz"), **d)
    print("loader ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None, help="stages to regenerate (default: all)")
    only = ap.parse_args().only
    want = lambda stage: only is None or stage in only
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    FF, TL, RLE = import_reference()
    if want("feedforward"):
        gen_feedforward(FF, RLE)
    if want("feedforward_options"):
        gen_feedforward_options(FF, RLE)
    if want("transformer_layer"):
        gen_transformer_layer(TL)
    if want("transformer_layer_dropout"):
        gen_transformer_layer_dropout(TL)
    if want("groundlink"):
        gen_groundlink(RLE)
    if want("loss"):
        gen_loss(RLE)
    if want("checkpoint_analyze"):
        gen_checkpoint_analyze(FF, RLE)
    if want("optim"):
        gen_optim()
    if want("loader"):
        gen_loader()


if __name__ == "__main__":
    main()
