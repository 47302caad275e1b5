"""Tensor-dict contract of the reference data layer + a nimble-free synthetic window source.

``InputDataKeys`` / ``OutputDataKeys`` are the string constants every model / loss / CLI of the
reference agrees on (src/data/AddBiomechanicsDataset.py:9-42); they are reproduced verbatim because
they ARE the drop-in boundary.  The reference's ``AddBiomechanicsDataset`` reads ``.b3d`` files through
``nimblephysics`` (a third-party C++ wheel that is neither vendored nor installed here); that loader is
a "next" row of SURVEY.md §8f.  ``SyntheticWindowDataset`` yields windows with exactly the shapes and
tuple layout of ``AddBiomechanicsDataset.__getitem__`` (src/data/AddBiomechanicsDataset.py:161-285) so
the train / analyze commands and the benchmarks run without ``.b3d`` data.
"""
from typing import Dict, List, Tuple

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


def input_key_widths(num_dofs: int, history_width: int) -> List[int]:
    """channel widths per input key; history_width = stride*3 for the feedforward model
    (FeedForwardRegressionBaseline.py:92-94), root_history_len*3 for groundlink (Groundlink.py:116-118)."""
    return [num_dofs, num_dofs, num_dofs, 3, 3, 3, 3, 36, history_width, history_width]


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


class SyntheticMotionWindows(Dataset):
    """Seeded ``[T, D]`` motion windows for the diffusion denoisers (x0 ~ N(0,1))."""

    def __init__(self, num_windows: int, window: int = 50, feat: int = 300, seed: int = 0):
        self.num_windows, self.window, self.feat, self.seed = num_windows, window, feat, seed

    def __len__(self):
        return self.num_windows

    def __getitem__(self, index: int) -> torch.Tensor:
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        return torch.randn(self.window, self.feat, generator=g)
