"""Groundlink on hand-written gfx950 kernels (registry name ``groundlink``).

Same constructor, parameter names (``cnn.{1,4,7,10}.{weight,bias}``, ``fc.{2,5}.{weight,bias}``, ``fc.8.weight``),
initialisation, tensor-dict call convention and output slices as the reference class (src/models/Groundlink.py:19-156);
the arithmetic is one HIP launch plan (``plans.GroundlinkPlan``): the ten input keys are gathered into one channels-last
matrix, each Conv1d(k=7, padding_mode="replicate") + ELU is an im2col gather + the fused Linear+bias+ELU MFMA GEMM, the
fully connected part is three more GEMMs with a counter-based dropout in train mode.  The reference's registry call
(src/cli/abstract_command.py:74-79) passes the feedforward argument list and raises TypeError; ``get_model`` here calls
this constructor with its own signature."""
from typing import Dict

import torch
import torch.nn as nn

from .. import hip
from ..data.AddBiomechanicsDataset import INPUT_KEY_ORDER, InputDataKeys, OutputDataKeys
from ..module import HipModule
from ..plans import GroundlinkPlan


class Groundlink(HipModule):
    def __init__(self, num_dofs: int, num_joints: int, root_history_len: int, output_data_format: str = "all_frames",
                 cnn_kernel=7, cnn_dropout=0.0, fc_depth=3, fc_dropout=0.2, device='cpu',
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__(compute_dtype)
        if fc_depth != 3 or cnn_kernel % 2 != 1:
            raise NotImplementedError("groundlink: fc_depth 3 and an odd cnn_kernel (the reference defaults) are built")
        if cnn_dropout != 0.0:
            raise NotImplementedError("groundlink: cnn_dropout is 0.0 in the reference (Groundlink.py:20)")
        self.num_dofs, self.num_joints, self.root_history_len = num_dofs, num_joints, root_history_len
        self.output_data_format, self.cnn_kernel, self.fc_dropout = output_data_format, cnn_kernel, fc_dropout
        self.channels = num_dofs * 3 + 12 + num_joints * 3 + root_history_len * 6          # Groundlink.py:26
        feats = [self.channels, 128, 128, 256, 256]
        # construction order = the reference's (every layer default-initialised first, then xavier-normal with the relu
        # gain on the layers an ELU follows, biases zeroed: Groundlink.py:79-103), so a seeded build matches it draw for draw
        self.cnn = nn.ModuleDict()
        for idx, (ci, co) in zip(GroundlinkPlan.CONV, zip(feats[:-1], feats[1:])):
            self.cnn[str(idx)] = nn.Conv1d(ci, co, cnn_kernel, padding=cnn_kernel // 2, padding_mode="replicate",
                                           device=device)
        self.fc = nn.ModuleDict()
        for idx in GroundlinkPlan.FC:
            self.fc[str(idx)] = nn.Linear(feats[-1], feats[-1], device=device)
        self.fc["8"] = nn.Linear(feats[-1], 30, bias=False, device=device)
        gain = torch.nn.init.calculate_gain("relu")
        for m in list(self.cnn.values()) + [self.fc[str(i)] for i in GroundlinkPlan.FC]:
            torch.nn.init.xavier_normal_(m.weight, gain)
            torch.nn.init.zeros_(m.bias)
        self._plan = None
        self._fwd_count = 0

    # ---- sizes the fused trainer asks for (they depend on the window length, which the reference ctor does not know)
    def input_size_for(self, frames: int) -> int:
        return frames * self.channels

    def output_frames_for(self, frames: int) -> int:
        return frames if self.output_data_format == 'all_frames' else 1

    def _get_plan(self, device) -> GroundlinkPlan:
        if self._plan is None or self._plan.buf.device != device or self._plan.dtype != self.compute_dtype:
            self._plan = GroundlinkPlan(self.output_data_format, self.compute_dtype, device, self.fc_dropout,
                                        self.cnn_kernel)
        return self._plan

    def pack_inputs(self, input: Dict[str, torch.Tensor], device) -> torch.Tensor:
        # shape checks of the reference (Groundlink.py:107-118)
        assert len(input[InputDataKeys.POS].shape) == 3
        assert input[InputDataKeys.POS].shape[-1] == self.num_dofs
        assert input[InputDataKeys.VEL].shape[-1] == self.num_dofs
        assert input[InputDataKeys.ACC].shape[-1] == self.num_dofs
        assert input[InputDataKeys.JOINT_CENTERS_IN_ROOT_FRAME].shape[-1] == self.num_joints * 3
        assert input[InputDataKeys.ROOT_POS_HISTORY_IN_ROOT_FRAME].shape[-1] == self.root_history_len * 3
        assert input[InputDataKeys.ROOT_EULER_HISTORY_IN_ROOT_FRAME].shape[-1] == self.root_history_len * 3
        B, F = input[InputDataKeys.POS].shape[:2]
        ts = [input[k].to(device=device, dtype=torch.float32, non_blocking=True).contiguous() for k in INPUT_KEY_ORDER]
        x = self._get_plan(device).buf.get("ff.x", (B, F * self.channels), self.compute_dtype)
        hip.concat_keys(ts, x)
        return x

    def _plan_forward(self, x: torch.Tensor) -> torch.Tensor:
        F = x.shape[1] // self.channels
        out = torch.empty((x.shape[0], self.output_frames_for(F), 30), dtype=self.compute_dtype, device=x.device)
        self._fwd_count += 1
        return self._get_plan(x.device).forward(x, self.param_source(), out=out, training=self.training,
                                                step=self._fwd_count)

    def _plan_backward(self, dout, P, accumulate):
        self._plan.backward(dout, P, accumulate)
        return None

    @staticmethod
    def split_output(x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """slices of the last dimension: Groundlink.py:151-156"""
        return {
            OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME: x[:, :, 0:6],
            OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME: x[:, :, 6:12],
            OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME: x[:, :, 12:18],
            OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME: x[:, :, 18:30],
        }

    def forward(self, input: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        self.ensure_packed()
        x = self.pack_inputs(input, self._flat.device)
        return self.split_output(self.run_plan(x))
