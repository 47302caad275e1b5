"""The `.b3d` window loader (inferbiomechanics_amd/data/AddBiomechanicsDataset.py::AddBiomechanicsDataset) against
windows produced by the REAL reference class (src/data/AddBiomechanicsDataset.py:63-139, :161-285, :287-303) over the
same closed-form fake `nimblephysics` (oracle/fake_nimble.py; fixture: tests/golden/loader_windows.npz written by
oracle/make_golden.py::gen_loader).  Everything here is bit-exact: the loader only moves and rounds numbers.  CPU only."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from oracle import fake_nimble
from oracle.fixture_inputs import LOADER_CASES, loader_sample
from inferbiomechanics_amd.data.AddBiomechanicsDataset import (INPUT_KEY_ORDER, LOSS_KEY_ORDER, AddBiomechanicsDataset,
                                                               OutputDataKeys)
from inferbiomechanics_amd.data.WindowCache import PackedWindows


@pytest.fixture()
def nimble(monkeypatch):
    monkeypatch.setitem(sys.modules, "nimblephysics", fake_nimble)
    return fake_nimble


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "loader_windows.npz"))


def open_case(root, case, **kw):
    name, window, stride, fmt, dt = case
    return AddBiomechanicsDataset(root, window, None, dtype=getattr(torch, dt), stride=stride, output_data_format=fmt,
                                  skip_loading_skeletons=True, **kw)


@pytest.mark.parametrize("case", LOADER_CASES, ids=[c[0] for c in LOADER_CASES])
def test_index_and_windows_match_reference(tmp_path, nimble, golden, case):
    name = case[0]
    root = str(tmp_path / "train")
    paths = fake_nimble.make_tree(root)
    ds = open_case(root, case)
    assert ds.subject_paths == paths                                   # 'vander' and non-.b3d files skipped
    assert ds.subject_indices == {p: i for i, p in enumerate(paths)}
    assert ds.contact_bodies == list(golden[f"{name}/contact_bodies"]) and ds.num_contact_bodies == 2
    assert ds.num_dofs == int(golden[f"{name}/num_dofs"])
    assert np.array_equal(np.asarray(ds.windows), golden[f"{name}/windows"])
    assert len(ds) == golden[f"{name}/windows"].shape[0]
    for i in loader_sample(len(ds)):
        inputs, labels, subj, trial = ds[i]
        assert [subj, trial] == list(golden[f"{name}/{i}/meta"]) and type(subj) is int and type(trial) is int
        assert len(inputs) == 10 and len(labels) == 7
        for k, v in inputs.items():
            ref = golden[f"{name}/{i}/in/{k}"]
            assert v.dtype == getattr(torch, case[4]) and tuple(v.shape) == ref.shape
            assert np.array_equal(v.numpy(), ref), (name, i, k)
        for k, v in labels.items():
            ref = golden[f"{name}/{i}/lab/{k}"]
            assert v.dtype == getattr(torch, case[4]) and tuple(v.shape) == ref.shape
            assert np.array_equal(v.numpy(), ref), (name, i, k)


def test_window_index_rule(tmp_path, nimble):
    """the index restated from the reference loop (:131-139), independent of the fixture: every start below
    max(len - window - 1, 0) whose strided taps are all measured"""
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = open_case(root, ("x", 30, 4, "all_frames", "float32"))
    want = []
    for si, name in enumerate(("alpha", "beta", "gamma")):
        s = fake_nimble.SubjectOnDisk(os.path.join(root, name + ".b3d"))
        for t in range(s.getNumTrials()):
            miss = [r != fake_nimble.MissingGRFReason.notMissingGRF for r in s.getMissingGRF(t)]
            for w in range(max(s.getTrialLength(t) - 30 - 1, 0)):
                if not any(miss[w:w + 30:4]):
                    want.append((si, t, w))
    assert [tuple(r) for r in np.asarray(ds.windows)] == want
    w50 = np.asarray(open_case(root, LOADER_CASES[0]).windows)
    assert not np.any((w50[:, 0] == 0) & (w50[:, 1] == 2))            # alpha's 40-frame trial is shorter than a 50-frame window


def test_subject_without_a_contact_body_gets_zeros_and_mass_division(tmp_path, nimble):
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = open_case(root, LOADER_CASES[0])
    w = np.asarray(ds.windows)
    i = int(np.flatnonzero(w[:, 0] == 2)[0])                           # gamma: only calcn_l
    _, labels, _, _ = ds[i]
    for k, c in ((OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME, 3), (OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME, 3),
                 (OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME, 3), (OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME, 6)):
        assert torch.count_nonzero(labels[k][:, c:]) == 0 and torch.count_nonzero(labels[k][:, :c]) == labels[k][:, :c].numel()
    # beta lists (pelvis, calcn_r, calcn_l): calcn_l is ITS third body, divided by beta's mass; CoP is not divided
    j = int(np.flatnonzero(w[:, 0] == 1)[0])
    _, labels, _, trial = ds[j]
    start = int(w[j, 2])
    spec = fake_nimble.SUBJECTS["beta"]
    p0 = fake_nimble.FramePass(spec, trial, start, 0)
    f32 = lambda a: torch.from_numpy(np.asarray(a)).to(torch.float32)
    assert torch.equal(labels[OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME][0, 0:3],
                       f32(p0.groundContactForceInRootFrame[6:9]) / spec["mass"])
    assert torch.equal(labels[OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME][0, 3:6],
                       f32(p0.groundContactCenterOfPressureInRootFrame[3:6]))
    plast = fake_nimble.FramePass(spec, trial, start, spec["passes"] - 1)
    assert torch.equal(labels[OutputDataKeys.TAU][0], f32(plast.tau))  # tau from the LAST pass, inputs from the first


def test_single_file_and_short_slice(tmp_path, nimble, golden):
    paths = fake_nimble.make_tree(str(tmp_path / "train"))
    one = AddBiomechanicsDataset(paths[1], 50, None, stride=5, output_data_format="all_frames", skip_loading_skeletons=True)
    assert one.contact_bodies == list(golden["single/contact_bodies"])  # first subject decides the order; pelvis dropped
    assert np.array_equal(np.asarray(one.windows), golden["single/windows"])
    _, labels, _, _ = one[7]
    assert np.array_equal(labels[OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME].numpy(), golden["single/7/force"])
    assert np.array_equal(labels[OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME].numpy(), golden["single/7/wrench"])
    with pytest.raises(AssertionError):
        AddBiomechanicsDataset(str(tmp_path / "train" / "notes.txt"), 50, None)
    names = ["alpha"] + [f"tiny{j:02d}" for j in range(12)]
    paths2 = fake_nimble.make_tree(str(tmp_path / "short"), names)
    short = AddBiomechanicsDataset(str(tmp_path / "short"), 50, None, stride=5, testing_with_short_dataset=True,
                                   skip_loading_skeletons=True)
    assert short.subject_paths == paths2[11:12] and os.path.basename(short.subject_paths[0]) == str(golden["short/subject"])
    assert np.array_equal(np.asarray(short.windows), golden["short/windows"])


def test_skeletons_loaded_unless_skipped(tmp_path, nimble):
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = AddBiomechanicsDataset(root, 50, "geom/", stride=5)
    assert len(ds.skeletons) == 3 and [[b.getName() for b in bs] for bs in ds.skeletons_contact_bodies] == [ds.contact_bodies] * 3
    assert open_case(root, LOADER_CASES[0]).skeletons == []


def test_inspect_dof_indices(tmp_path, nimble, capsys):
    """AddBiomechanicsDataset.py:141-156: 23 DoFs, the same name at every index across the skeletons; the three failure
    messages of the reference for a short skeleton, a renamed coordinate and a 24-DoF set"""
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = AddBiomechanicsDataset(root, 50, "geom/", stride=5)
    ds.inspect_dof_indices()
    out = capsys.readouterr().out
    assert "Skeleton 3/3 joints:" in out and "  - Dof Index 22/22: dof_22" in out
    assert " - Set of names at dof index 0: {'dof_0'}" in out
    good = [f"dof_{j}" for j in range(23)]
    ds.skeletons[1] = fake_nimble._Skeleton(70.0, good[:22])
    with pytest.raises(AssertionError, match="2 entries found at dof index 22, expected 3"):
        ds.inspect_dof_indices()
    ds.skeletons[1] = fake_nimble._Skeleton(70.0, good[:5] + ["knee_r"] + good[6:])
    with pytest.raises(AssertionError, match="2 distinct dof names found at dof index 5, expected 1"):
        ds.inspect_dof_indices()
    ds.skeletons = [fake_nimble._Skeleton(70.0, good + ["extra"]) for _ in range(3)]
    with pytest.raises(AssertionError, match="24 unique dof indices found, expected 23"):
        ds.inspect_dof_indices()
    assert open_case(root, LOADER_CASES[0]).skeletons == []


def test_worker_copy_reopens_subjects(tmp_path, nimble):
    """DataLoader workers get a pickled copy: the SubjectOnDisk handles are dropped and re-opened (:287-303)"""
    root = str(tmp_path / "train")
    paths = fake_nimble.make_tree(root)
    ds = AddBiomechanicsDataset(root, 50, None, stride=5, output_data_format="all_frames")
    blob = pickle.dumps(ds)
    fake_nimble.SubjectOnDisk.opened.clear()
    twin = pickle.loads(blob)
    assert fake_nimble.SubjectOnDisk.opened == paths and twin.skeletons == []
    a, b = ds[40], twin[40]
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and all(torch.equal(a[1][k], b[1][k]) for k in a[1])
    # and through a real worker process
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=1)
    inputs, labels, subj, trial = next(iter(loader))
    assert inputs[INPUT_KEY_ORDER[0]].shape == (4, 10, 23) and torch.equal(inputs[INPUT_KEY_ORDER[0]][2], ds[2][0][INPUT_KEY_ORDER[0]])


@pytest.mark.parametrize("case", LOADER_CASES[:2], ids=[c[0] for c in LOADER_CASES[:2]])
def test_packed_rows_equal_the_tuple_path(tmp_path, nimble, golden, case):
    """window_row writes the packed row directly; it must hold the same bits as packing the reference tuple"""
    name = case[0]
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = open_case(root, case)
    pack = PackedWindows.from_dataset(ds)
    assert len(pack) == len(ds) and pack.frames == 10 and pack.out_frames == (10 if case[3] == "all_frames" else 1)
    assert pack.input_widths == [23, 23, 23, 3, 3, 3, 3, 36, 30, 30]
    w = np.asarray(ds.windows)
    assert np.array_equal(pack.subjects, w[:, 0]) and np.array_equal(pack.trials, w[:, 1])
    for i in loader_sample(len(ds)):                                  # against the REAL reference's tensors
        inputs = {k: torch.from_numpy(golden[f"{name}/{i}/in/{k}"]) for k in INPUT_KEY_ORDER}
        labels = {k: torch.from_numpy(golden[f"{name}/{i}/lab/{k}"]) for k in LOSS_KEY_ORDER}
        assert np.array_equal(pack.rows[i], PackedWindows.row_of(inputs, labels)), (name, i)
    for i in range(0, len(ds), 17):                                   # and against this loader's own tuples
        item = ds[i]
        assert np.array_equal(pack.rows[i], PackedWindows.row_of(item[0], item[1]))
        back = pack.window(i)
        assert all(torch.equal(back[0][k], item[0][k]) for k in INPUT_KEY_ORDER)
    limited = PackedWindows.from_dataset(ds, limit=5)
    assert len(limited) == 5 and np.array_equal(limited.rows, pack.rows[:5])


def test_packing_in_worker_processes(tmp_path, nimble):
    root = str(tmp_path / "train")
    fake_nimble.make_tree(root)
    ds = open_case(root, LOADER_CASES[0])
    a = PackedWindows.from_dataset(ds)
    b = PackedWindows.from_dataset(ds, workers=2)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.trials, b.trials)


def test_cli_trains_and_analyzes_b3d_trees(tmp_path, nimble, monkeypatch):
    """`main.py train / analyze --dataset-home <dir of .b3d trees>` end to end with the kernels in dry-run: the loader
    feeds the DataLoader path, the packed on-device window cache (odd frame count: 30-value label blocks) and the
    diffusion view; analyze names rows by subject file / trial and runs the inverse-dynamics report on the skeletons"""
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.main import main
    hip.set_dry_run(True)
    try:
        home = tmp_path / "data"
        for split in ("train", "dev"):
            fake_nimble.make_tree(str(home / split))
        ck = str(tmp_path / "ck")
        base = ['--no-wandb', '--dataset-home', str(home), '--batch-size', '8', '--checkpoint-dir', ck,
                '--data-loading-workers', '0', '--history-len', '50', '--stride', '10']      # histories: 30 = stride * 3
        assert main(['train', '--epochs', '1', '--max-steps', '2'] + base)
        assert os.listdir(os.path.join(ck, 'feedforward')) == ['epoch_0_batch_1.pt']
        cache = str(tmp_path / "train.ibw")
        assert main(['train', '--epochs', '2', '--max-steps', '2', '--window-cache', cache] + base)
        pack = PackedWindows.load(cache)
        assert pack.frames == 5 and pack.label_elems == [30, 30, 30, 60] and pack.label_pad == [32, 32, 32, 60]
        ds = AddBiomechanicsDataset(str(home / "train"), 50, None, stride=10, output_data_format='all_frames',
                                    skip_loading_skeletons=True)
        assert len(pack) == len(ds)
        item = ds[len(ds) - 1]
        assert np.array_equal(pack.rows[len(ds) - 1], PackedWindows.row_of(item[0], item[1]))
        assert main(['analyze', '--no-wandb', '--dataset-home', str(home), '--checkpoint-dir', ck, '--history-len', '50',
                     '--stride', '10', '--data-loading-workers', '0', '--max-windows', '3'])
        rows = open(os.path.join(ck, 'feedforward', 'dev_analysis.csv')).read().strip().splitlines()
        assert rows == ['alpha.b3d,trial_0'] * 3
        assert main(['train', '--model-type', 'groundlink', '--epochs', '1', '--max-steps', '1', '--stride', '5'] + base[:-2])
        assert main(['train', '--model-type', 'diffusion-mlp', '--epochs', '1', '--max-steps', '2', '--hidden-dims', '32', '32',
                     '--compute-dtype', 'bf16'] + base)
    finally:
        hip.set_dry_run(False)
