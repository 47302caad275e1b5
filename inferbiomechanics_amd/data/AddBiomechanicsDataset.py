"""Tensor-dict contract of the reference data layer, the `.b3d` window loader and a nimble-free synthetic source.

``InputDataKeys`` / ``OutputDataKeys`` are the string constants every model / loss / CLI of the
reference agrees on (src/data/AddBiomechanicsDataset.py:9-42); they are reproduced verbatim because
they ARE the drop-in boundary.

``AddBiomechanicsDataset`` is the reference's window loader (src/data/AddBiomechanicsDataset.py:45-303) with the same
constructor, attributes, window index and per-item tuple.  It reads ``.b3d`` files through ``nimblephysics``
(a third-party C++ wheel, imported lazily -- the module itself imports without it).  What differs is HOW a window is
assembled: one ``numpy`` stack per key instead of one ``torch.tensor`` per frame per key, the window index from
strided prefix sums instead of a Python loop per start frame, and ``window_row`` which writes a window straight into its
packed row (data/WindowCache.py) without building the 17 tensors at all.  Values are bit-identical to the reference's
(tests/golden/loader_windows.npz is produced by the REAL reference class over oracle/fake_nimble.py's closed-form
subjects).

``SyntheticWindowDataset`` yields windows with exactly the shapes and tuple layout of
``AddBiomechanicsDataset.__getitem__`` so the train / analyze commands and the benchmarks run without ``.b3d`` data.
"""
import os
import sys
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class InputDataKeys:
    POS = 'pos'
    VEL = 'vel'
    ACC = 'acc'
    JOINT_CENTERS_IN_ROOT_FRAME = 'jointCentersInRootFrame'
    ROOT_LINEAR_VEL_IN_ROOT_FRAME = 'rootLinearVelInRootFrame'
    ROOT_ANGULAR_VEL_IN_ROOT_FRAME = 'rootAngularVelInRootFrame'
    ROOT_LINEAR_ACC_IN_ROOT_FRAME = 'rootLinearAccInRootFrame'
    ROOT_ANGULAR_ACC_IN_ROOT_FRAME = 'rootAngularAccInRootFrame'
    ROOT_POS_HISTORY_IN_ROOT_FRAME = 'rootPosHistoryInRootFrame'
    ROOT_EULER_HISTORY_IN_ROOT_FRAME = 'rootEulerHistoryInRootFrame'


class OutputDataKeys:
    TAU = 'tau'
    GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME = 'groundContactWrenchesInRootFrame'
    RESIDUAL_WRENCH_IN_ROOT_FRAME = 'residualWrenchInRootFrame'
    CONTACT = 'contact'
    COM_ACC_IN_ROOT_FRAME = 'comAccInRootFrame'
    GROUND_CONTACT_COPS_IN_ROOT_FRAME = 'groundContactCenterOfPressureInRootFrame'
    GROUND_CONTACT_TORQUES_IN_ROOT_FRAME = 'groundContactTorqueInRootFrame'
    GROUND_CONTACT_FORCES_IN_ROOT_FRAME = 'groundContactForceInRootFrame'


# concat order of the model input (src/models/FeedForwardRegressionBaseline.py:97-107)
INPUT_KEY_ORDER: List[str] = [
    InputDataKeys.POS, InputDataKeys.VEL, InputDataKeys.ACC,
    InputDataKeys.ROOT_LINEAR_VEL_IN_ROOT_FRAME, InputDataKeys.ROOT_ANGULAR_VEL_IN_ROOT_FRAME,
    InputDataKeys.ROOT_LINEAR_ACC_IN_ROOT_FRAME, InputDataKeys.ROOT_ANGULAR_ACC_IN_ROOT_FRAME,
    InputDataKeys.JOINT_CENTERS_IN_ROOT_FRAME,
    InputDataKeys.ROOT_POS_HISTORY_IN_ROOT_FRAME, InputDataKeys.ROOT_EULER_HISTORY_IN_ROOT_FRAME,
]
# order the loss kernel takes outputs / labels in (cop, force, torque, wrench)
LOSS_KEY_ORDER: List[str] = [
    OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME, OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME,
    OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME, OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME,
]
LOSS_KEY_WIDTHS = [6, 6, 6, 12]


def packed_row_layout(frames: int, out_frames: int, input_widths) -> Tuple[int, int, List[int], List[int], int]:
    """layout of one packed window row (data/WindowCache.py): (x_elems, x_pad, label_elems, label_pad, row_elems).
    Every segment starts on a 16-byte boundary: the input block and each label block are zero-padded to 4 values."""
    x_elems = int(frames) * int(sum(input_widths))
    x_pad = (x_elems + 3) // 4 * 4
    label_elems = [int(out_frames) * c for c in LOSS_KEY_WIDTHS]
    label_pad = [(e + 3) // 4 * 4 for e in label_elems]
    return x_elems, x_pad, label_elems, label_pad, x_pad + sum(label_pad)


def input_key_widths(num_dofs: int, history_width: int) -> List[int]:
    """channel widths per input key; history_width = stride*3 for the feedforward model
    (FeedForwardRegressionBaseline.py:92-94), root_history_len*3 for groundlink (Groundlink.py:116-118)."""
    return [num_dofs, num_dofs, num_dofs, 3, 3, 3, 3, 36, history_width, history_width]


def _nimble():
    """`nimblephysics`, imported on first use (the wheel is optional: everything synthetic runs without it)"""
    mod = sys.modules.get('nimblephysics')
    if mod is None:
        import nimblephysics as mod  # noqa: F811
    return mod


# attribute of a frame's processing pass behind every input key (AddBiomechanicsDataset.py:181-210: the attribute names
# ARE the key strings)
_INPUT_FIELDS: List[str] = list(INPUT_KEY_ORDER)


class AddBiomechanicsDataset(Dataset):
    """Sliding windows over the trials of every ``.b3d`` subject below ``data_path``.

    Same constructor and attributes as the reference class (AddBiomechanicsDataset.py:63-139):
    ``subject_paths`` (``os.walk`` order, ``.b3d`` files without 'vander' in the name; ``testing_with_short_dataset``
    keeps ``[11:12]``), ``subject_indices``, ``num_dofs`` / ``contact_bodies`` from the FIRST subject ('pelvis' dropped),
    ``subjects`` / ``skeletons`` / ``skeletons_contact_bodies``, and ``windows`` = every ``(subject, trial, start)``
    with ``start < max(trial_length - window_size - 1, 0)`` whose strided taps ``start, start + stride, ... <
    start + window_size`` all carry measured ground-reaction forces.  ``windows`` is an int64 ``[N, 3]`` array here (the
    reference keeps a list of tuples; rows unpack the same way).

    ``__getitem__`` returns the reference tuple ``(inputs, labels, subject_index, trial)`` (:161-285): 10 input tensors
    ``[F, c]`` from the FIRST processing pass of ``F = window_size // stride`` frames read with the frame stride;
    7 label tensors over all ``F`` frames (``'all_frames'``) or the last one: tau / residual wrench / COM acceleration
    from the LAST pass, ground-contact wrench / CoP / torque / force from the FIRST pass re-ordered to the data set's
    ``contact_bodies`` (zeros for a body the subject lacks), wrench / torque / force divided by the subject's mass
    (:248-261).
    """

    def __init__(self,
                 data_path: str,
                 window_size: int,
                 geometry_folder: Optional[str],
                 device: torch.device = torch.device('cpu'),
                 dtype: torch.dtype = torch.float32,
                 testing_with_short_dataset: bool = False,
                 stride: int = 1,
                 output_data_format: str = 'last_frame',
                 skip_loading_skeletons: bool = False):
        nimble = _nimble()
        self.stride = stride
        self.output_data_format = output_data_format
        self.subject_paths: List[str] = []
        self.subjects: list = []
        self.window_size = window_size
        self.geometry_folder = geometry_folder
        self.device = device
        self.dtype = dtype
        self.contact_bodies: List[str] = []
        self.skeletons: list = []
        self.skeletons_contact_bodies: list = []

        if os.path.isdir(data_path):
            for root, _dirs, files in os.walk(data_path):
                self.subject_paths += [os.path.join(root, f) for f in files
                                       if f.endswith(".b3d") and "vander" not in f.lower()]
        else:
            assert data_path.endswith(".b3d")
            self.subject_paths.append(data_path)
        if testing_with_short_dataset:
            self.subject_paths = self.subject_paths[11:12]
        self.subject_indices: Dict[str, int] = {p: i for i, p in enumerate(self.subject_paths)}

        SubjectOnDisk = nimble.biomechanics.SubjectOnDisk
        measured = nimble.biomechanics.MissingGRFReason.notMissingGRF
        if self.subject_paths:
            head = SubjectOnDisk(self.subject_paths[0])
            self.num_dofs = head.getNumDofs()
            for body in head.getGroundForceBodies():
                if body != 'pelvis' and body not in self.contact_bodies:
                    self.contact_bodies.append(body)
        print(f"CONTACT BODIES: {self.contact_bodies}")
        self.num_contact_bodies = len(self.contact_bodies)

        taps = range(0, window_size, stride)
        index: List[np.ndarray] = []
        for i, path in enumerate(self.subject_paths):
            subject = SubjectOnDisk(path)
            if not skip_loading_skeletons:
                print(f'Loading skeleton {i + 1}/{len(self.subject_paths)} for subject {path}')
                skeleton = subject.readSkel(subject.getNumProcessingPasses() - 1, geometry_folder)
                self.skeletons.append(skeleton)
                self.skeletons_contact_bodies.append([skeleton.getBodyNode(b) for b in self.contact_bodies])
            self.subjects.append(subject)
            for trial in range(subject.getNumTrials()):
                length = subject.getTrialLength(trial)
                starts = max(length - window_size - 1, 0)
                if starts == 0:
                    continue
                missing = np.fromiter((r != measured for r in subject.getMissingGRF(trial)), dtype=bool, count=-1)
                if missing.shape[0] < length:               # a shorter flag list ends the taps early, like the slice does
                    missing = np.concatenate([missing, np.zeros(length - missing.shape[0], bool)])
                bad = np.zeros(starts, dtype=bool)
                for k in taps:                               # start + k < length for every start (start + window < length)
                    bad |= missing[k:k + starts]
                good = np.flatnonzero(~bad)
                index.append(np.stack([np.full_like(good, i), np.full_like(good, trial), good], axis=1))
        self.windows = np.concatenate(index).astype(np.int64) if index else np.zeros((0, 3), np.int64)
        self._contact_cache: Dict[int, List[int]] = {}

    def inspect_dof_indices(self):
        """the reference's consistency check over the loaded skeletons (AddBiomechanicsDataset.py:141-156): every skeleton
        has the same 23 degrees of freedom, and index j names the same joint coordinate in all of them.  Same prints, same
        assertion messages; the names are gathered per skeleton first, so a failure names the index it is about."""
        num_skeletons = len(self.skeletons)
        names = []
        for i, skeleton in enumerate(self.skeletons):
            print(f'Skeleton {i + 1}/{num_skeletons} joints:')
            num_dofs = skeleton.getNumDofs()
            row = [skeleton.getDofByIndex(j).getName() for j in range(num_dofs)]
            for j, dof_name in enumerate(row):
                print(f'  - Dof Index {j}/{num_dofs - 1}: {dof_name}')
            names.append(row)
        print('-' * 80)
        width = max((len(row) for row in names), default=0)
        assert width == 23, f'{width} unique dof indices found, expected 23'
        for j in range(width):
            at_j = [row[j] for row in names if j < len(row)]
            assert len(at_j) == num_skeletons, f'{len(at_j)} entries found at dof index {j}, expected {num_skeletons}'
            print(f' - Set of names at dof index {j}: {set(at_j)}')
            assert len(set(at_j)) == 1, f'{len(set(at_j))} distinct dof names found at dof index {j}, expected 1'

    def __len__(self) -> int:
        return int(self.windows.shape[0])

    # ---- one window --------------------------------------------------------------------------------------------
    def _np_dtype(self):
        return torch.empty(0, dtype=self.dtype).numpy().dtype

    def _read(self, index: int):
        subject_index, trial, start = (int(v) for v in self.windows[index])
        subject = self.subjects[subject_index]
        n = self.window_size // self.stride
        frames = subject.readFrames(trial, start, n, stride=self.stride, includeSensorData=False,
                                    includeProcessingPasses=True)
        assert len(frames) == n
        first = [f.processingPasses[0] for f in frames]
        last = [f.processingPasses[-1] for f in frames]
        if self.output_data_format != 'all_frames':         # labels: the window's last frame only (:216)
            first_out, last = first[-1:], last[-1:]
        else:
            first_out = first
        return subject, subject_index, trial, first, first_out, last

    def _contact_indices(self, subject_index: int, subject) -> List[int]:
        """position of every data-set contact body in THIS subject's ground-force body list, -1 if it has none (:239-241)"""
        hit = self._contact_cache.get(subject_index)
        if hit is None:
            own = list(subject.getGroundForceBodies())
            hit = [own.index(b) if b in own else -1 for b in self.contact_bodies]
            self._contact_cache[subject_index] = hit
        return hit

    @staticmethod
    def _stack(passes, field: str) -> np.ndarray:
        return np.stack([np.asarray(getattr(p, field), dtype=np.float64) for p in passes])

    def __getitem__(self, index: int) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], int, int]:
        subject, subject_index, trial, first, first_out, last = self._read(index)
        npd = self._np_dtype()
        tens = lambda passes, field: torch.from_numpy(self._stack(passes, field).astype(npd))
        inputs = {
            InputDataKeys.POS: tens(first, 'pos'),
            InputDataKeys.VEL: tens(first, 'vel'),
            InputDataKeys.ACC: tens(first, 'acc'),
            InputDataKeys.JOINT_CENTERS_IN_ROOT_FRAME: tens(first, 'jointCentersInRootFrame'),
            InputDataKeys.ROOT_LINEAR_VEL_IN_ROOT_FRAME: tens(first, 'rootLinearVelInRootFrame'),
            InputDataKeys.ROOT_LINEAR_ACC_IN_ROOT_FRAME: tens(first, 'rootLinearAccInRootFrame'),
            InputDataKeys.ROOT_ANGULAR_VEL_IN_ROOT_FRAME: tens(first, 'rootAngularVelInRootFrame'),
            InputDataKeys.ROOT_ANGULAR_ACC_IN_ROOT_FRAME: tens(first, 'rootAngularAccInRootFrame'),
            InputDataKeys.ROOT_POS_HISTORY_IN_ROOT_FRAME: tens(first, 'rootPosHistoryInRootFrame'),
            InputDataKeys.ROOT_EULER_HISTORY_IN_ROOT_FRAME: tens(first, 'rootEulerHistoryInRootFrame'),
        }
        mass = subject.getMassKg()
        labels = {
            OutputDataKeys.TAU: tens(last, 'tau'),
            OutputDataKeys.RESIDUAL_WRENCH_IN_ROOT_FRAME: tens(last, 'residualWrenchInRootFrame'),
            OutputDataKeys.COM_ACC_IN_ROOT_FRAME: tens(last, 'comAccInRootFrame'),
        }
        rows, ncb = len(first_out), self.num_contact_bodies
        contact = self._contact_indices(subject_index, subject)
        for key, field, width, per_mass in (
                (OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME, 'groundContactWrenchesInRootFrame', 6, True),
                (OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME, 'groundContactCenterOfPressureInRootFrame', 3, False),
                (OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME, 'groundContactTorqueInRootFrame', 3, True),
                (OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME, 'groundContactForceInRootFrame', 3, True)):
            src = tens(first_out, field)
            out = torch.zeros((rows, width * ncb), dtype=self.dtype)
            for i, ci in enumerate(contact):
                if ci >= 0:
                    piece = src[:, width * ci:width * ci + width]
                    out[:, width * i:width * i + width] = piece / mass if per_mass else piece
            labels[key] = out
        return inputs, labels, subject_index, trial

    # ---- one window straight into its packed row (data/WindowCache.py layout) ------------------------------------
    def row_geometry(self) -> Tuple[int, int, List[int]]:
        """(frames, label frames, input widths in the model's concat order) -- needs at least one window"""
        if len(self) == 0:
            raise ValueError("AddBiomechanicsDataset: no windows")
        frames = self.window_size // self.stride
        s, t, w0 = (int(v) for v in self.windows[0])
        probe = self.subjects[s].readFrames(t, w0, 1, stride=self.stride, includeSensorData=False,
                                            includeProcessingPasses=True)[0].processingPasses[0]
        widths = [int(np.asarray(getattr(probe, f)).shape[0]) for f in _INPUT_FIELDS]
        return frames, frames if self.output_data_format == 'all_frames' else 1, widths

    def window_row(self, index: int, out: np.ndarray, widths: Optional[List[int]] = None) -> Tuple[int, int]:
        """Fill `out` (one float32 row, `packed_row_layout`) with window `index`: inputs frame-major in the model's key
        order (FeedForwardRegressionBaseline.py:97-107), then cop | force | torque | wrench key-major, every block
        zero-padded to 4 values -- the same bits `PackedWindows.row_of(*self[index][:2])` produces, without the 17
        intermediate tensors.  Returns (subject_index, trial)."""
        if self.num_contact_bodies != 2:
            raise ValueError("packed rows hold two contact bodies (label widths 6 / 6 / 6 / 12)")
        if out.dtype != np.float32 or out.ndim != 1:
            raise ValueError("window_row: `out` must be a 1-D float32 row")
        subject, subject_index, trial, first, first_out, _last = self._read(index)
        F, Fo = len(first), len(first_out)
        if widths is None:
            widths = [int(np.asarray(getattr(first[0], f)).shape[0]) for f in _INPUT_FIELDS]
        per_frame = sum(widths)
        x_elems, x_pad, label_elems, label_pad, row_elems = packed_row_layout(F, Fo, widths)
        if out.shape[0] != row_elems:
            raise ValueError(f"window_row: row of {out.shape[0]} values, window needs {row_elems}")
        x = out[:x_elems].reshape(F, per_frame)
        off = 0
        for f, w in zip(_INPUT_FIELDS, widths):
            x[:, off:off + w] = self._stack(first, f)        # float64 -> float32, round to nearest even like torch.tensor
            off += w
        out[x_elems:x_pad] = 0.0
        mass = np.float32(subject.getMassKg())               # torch divides the float32 tensor by the scalar IN float32
        contact = self._contact_indices(subject_index, subject)
        off = x_pad
        for k, (field, width, per_mass) in enumerate((('groundContactCenterOfPressureInRootFrame', 3, False),
                                                      ('groundContactForceInRootFrame', 3, True),
                                                      ('groundContactTorqueInRootFrame', 3, True),
                                                      ('groundContactWrenchesInRootFrame', 6, True))):
            src = self._stack(first_out, field).astype(np.float32)
            seg = out[off:off + label_elems[k]].reshape(Fo, width * 2)
            out[off + label_elems[k]:off + label_pad[k]] = 0.0
            for i, ci in enumerate(contact):
                dst = seg[:, width * i:width * i + width]
                if ci < 0:
                    dst[...] = 0.0
                else:
                    piece = src[:, width * ci:width * ci + width]
                    dst[...] = piece / mass if per_mass else piece
            off += label_pad[k]
        return subject_index, trial

    # ---- DataLoader workers: SubjectOnDisk handles do not pickle; a worker re-opens them (:287-303) ----------------
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ('subjects', 'skeletons', 'skeletons_contact_bodies'):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        print('Unpickling AddBiomechanicsDataset copy in reader worker thread')
        SubjectOnDisk = _nimble().biomechanics.SubjectOnDisk
        self.subjects = [SubjectOnDisk(p) for p in self.subject_paths]
        self.skeletons, self.skeletons_contact_bodies = [], []


class SyntheticWindowDataset(Dataset):
    """Seeded synthetic motion windows with the reference's per-item layout:
    ``(inputs: Dict[str, Tensor[F,c]], labels: Dict[str, Tensor[F',c']], subject_index, trial_index)``.
    Label forces are ~10*N(0,1) so the CoP mask (> 10 N/kg) is exercised."""

    def __init__(self, num_windows: int, history_len: int = 50, stride: int = 5, num_dofs: int = 23,
                 output_data_format: str = 'all_frames', seed: int = 0, history_width: int = 0):
        self.num_windows, self.history_len, self.stride = num_windows, history_len, stride
        self.history_width = history_width or stride * 3       # groundlink: root_history_len * 3 (Groundlink.py:116-118)
        self.num_dofs, self.num_contact_bodies = num_dofs, 2
        self.output_data_format = output_data_format
        self.seed = seed
        self.frames = history_len // stride
        self.out_frames = self.frames if output_data_format == 'all_frames' else 1

    def __len__(self):
        return self.num_windows

    def __getitem__(self, index: int) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], int, int]:
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        ws = input_key_widths(self.num_dofs, self.history_width)
        inputs = {k: torch.randn(self.frames, w, generator=g) for k, w in zip(INPUT_KEY_ORDER, ws)}
        F = self.out_frames
        labels = {
            OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME: 0.3 * torch.randn(F, 6, generator=g),
            OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME: 10.0 * torch.randn(F, 6, generator=g),
            OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME: torch.randn(F, 6, generator=g),
            OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME: 2.0 * torch.randn(F, 12, generator=g),
        }
        return inputs, labels, 0, index


class MotionWindowView(Dataset):
    """[BUILD-DEFINED] a regression window as ONE motion matrix ``[F, D]`` for the diffusion denoisers: per frame the
    model-input channels in the concat order (FeedForwardRegressionBaseline.py:97-107) followed by that frame's
    cop | force | torque | wrench labels -- D = 147 + 30 at the reference defaults.  Needs ``'all_frames'`` labels."""

    def __init__(self, dataset):
        if getattr(dataset, 'output_data_format', 'all_frames') != 'all_frames':
            raise ValueError("MotionWindowView: the diffusion models need --output-data-format all_frames")
        self.dataset = dataset
        self.num_dofs = getattr(dataset, 'num_dofs', 23)
        self.num_contact_bodies = getattr(dataset, 'num_contact_bodies', 2)

    def __len__(self):
        return len(self.dataset)

    @staticmethod
    def matrix(inputs: Dict[str, torch.Tensor], labels: Dict[str, torch.Tensor]) -> torch.Tensor:
        return torch.cat([inputs[k].to(torch.float32) for k in INPUT_KEY_ORDER]
                         + [labels[k].to(torch.float32) for k in LOSS_KEY_ORDER], dim=-1)

    @property
    def feat(self) -> int:
        return int(self[0].shape[-1])

    def __getitem__(self, index: int) -> torch.Tensor:
        item = self.dataset[index]
        return self.matrix(item[0], item[1])


class SyntheticMotionWindows(Dataset):
    """Seeded ``[T, D]`` motion windows for the diffusion denoisers (x0 ~ N(0,1))."""

    def __init__(self, num_windows: int, window: int = 50, feat: int = 300, seed: int = 0):
        self.num_windows, self.window, self.feat, self.seed = num_windows, window, feat, seed

    def __len__(self):
        return self.num_windows

    def __getitem__(self, index: int) -> torch.Tensor:
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        return torch.randn(self.window, self.feat, generator=g)
