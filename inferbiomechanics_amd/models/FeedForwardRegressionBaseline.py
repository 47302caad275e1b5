"""FeedForwardBaseline on hand-written gfx950 kernels (registry name ``feedforward``).

Same constructor, parameter names (``net.{2i}.weight/bias``), tensor-dict call convention and output
views as the reference class (src/models/FeedForwardRegressionBaseline.py:14-121); the arithmetic is
one HIP launch plan: a fused gather/cast of the 10 input keys (replacing ``torch.concat`` + ``.to(device)``,
:97-108) and one fused Linear+bias+activation MFMA GEMM per layer (replacing ``nn.Linear`` + activation,
:68-77,113), with an explicit backward (dgrad with the activation derivative fused, split-M wgrad).
"""
import logging
from typing import Dict, List

import torch
import torch.nn as nn

from .. import hip
from ..data.AddBiomechanicsDataset import INPUT_KEY_ORDER, InputDataKeys, OutputDataKeys
from ..module import HipModule
from ..plans import DenseStackPlan

ACTIVATION_FUNCS = ("relu", "tanh", "sigmoid", "silu")   # reference offers the first three (:7-11)


class FeedForwardBaseline(HipModule):
    num_dofs: int
    num_contact_bodies: int
    history_len: int
    root_history_len: int

    def __init__(self, num_dofs: int, num_contact_bodies: int, history_len: int, output_data_format: str,
                 activation: str, stride: int, root_history_len: int, hidden_dims: List[int] = [512, 512],
                 batchnorm: bool = False, dropout: bool = False, dropout_prob: float = 0.0, device: str = 'cpu',
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__(compute_dtype)
        if activation not in ACTIVATION_FUNCS:
            raise KeyError(activation)
        if dropout and not (0.0 <= dropout_prob < 1.0):
            raise ValueError(f"dropout probability has to be in [0, 1), but got {dropout_prob}")
        self.stride, self.activation, self.output_data_format = stride, activation, output_data_format
        self.num_dofs, self.num_contact_bodies = num_dofs, num_contact_bodies
        self.history_len, self.root_history_len = history_len, root_history_len
        self.device = device
        # sizes: FeedForwardRegressionBaseline.py:52,61-62
        self.input_size = (3 * num_dofs + 4 * 3 + 2 * stride * 3 + 12 * 3) * (history_len // stride)
        self.num_output_frames = (history_len // stride) if output_data_format == 'all_frames' else 1
        self.output_size = num_contact_bodies * (3 * 3 + 6) * self.num_output_frames
        dims = [self.input_size] + list(hidden_dims) + [self.output_size]
        logging.info(f"MODEL DIMENSIONS: input size = {self.input_size}, hidden dims = {hidden_dims}, "
                     f"output size = {self.output_size}")
        # same module indices as the reference Sequential (:67-77): per layer [Dropout] [BatchNorm1d(h0)] Linear, then
        # the activation on all but the last -- so `net.{j}` state-dict keys (Linear weight / bias, BatchNorm weight /
        # bias / running_mean / running_var / num_batches_tracked) are the reference's with every flag combination
        self.net = nn.ModuleDict()
        self.dropout_p = float(dropout_prob) if dropout else 0.0
        names, bns = [], []
        j = 0
        for i, (h0, h1) in enumerate(zip(dims[:-1], dims[1:])):
            if dropout:
                j += 1                                                     # nn.Dropout: no state
            if batchnorm:
                self.net[str(j)] = nn.BatchNorm1d(h0, device=device)        # parameter / buffer container (torch defaults)
                bns.append((f"net.{j}.weight", f"net.{j}.bias"))
                j += 1
            else:
                bns.append(None)
            self.net[str(j)] = nn.Linear(h0, h1, dtype=torch.float32, device=device)   # reference init (kaiming-uniform)
            names.append((f"net.{j}.weight", f"net.{j}.bias"))
            j += 1
            if i < len(dims) - 2:
                j += 1                                                     # the activation module
        self._names, self._bn = names, bns
        self.train_mode_matters = bool(dropout or batchnorm)               # HipTrainer passes training / the step counter
        self._fwd_calls = 0
        self._plan = None

    def _get_plan(self, device) -> DenseStackPlan:
        if self._plan is None or self._plan.buf.device != device or self._plan.dtype != self.compute_dtype:
            self._plan = DenseStackPlan(self._names, self.activation, self.compute_dtype, device, bn_names=self._bn,
                                        dropout_p=self.dropout_p)
        for bn in self._bn:        # (re)bind the BatchNorm buffers: load_state_dict copies in place, .to() replaces them
            if bn is not None:
                m = self.net[bn[0].split(".")[1]]
                self._plan.bn_buffers[bn[0]] = (m.running_mean, m.running_var, m.num_batches_tracked)
        return self._plan

    def pack_inputs(self, input: Dict[str, torch.Tensor], device) -> torch.Tensor:
        """The 10 keys -> one [B, F*147] matrix in HBM, frame-major (gather + cast in ONE kernel)."""
        # shape checks of the reference (:83-94)
        assert len(input[InputDataKeys.POS].shape) == 3
        assert input[InputDataKeys.POS].shape[-1] == self.num_dofs
        assert input[InputDataKeys.VEL].shape[-1] == self.num_dofs
        assert input[InputDataKeys.ACC].shape[-1] == self.num_dofs
        assert len(input[InputDataKeys.JOINT_CENTERS_IN_ROOT_FRAME].shape) == 3
        assert input[InputDataKeys.JOINT_CENTERS_IN_ROOT_FRAME].shape[-1] == 12 * 3
        assert len(input[InputDataKeys.ROOT_POS_HISTORY_IN_ROOT_FRAME].shape) == 3
        assert input[InputDataKeys.ROOT_POS_HISTORY_IN_ROOT_FRAME].shape[-1] == self.stride * 3
        assert len(input[InputDataKeys.ROOT_EULER_HISTORY_IN_ROOT_FRAME].shape) == 3
        assert input[InputDataKeys.ROOT_EULER_HISTORY_IN_ROOT_FRAME].shape[-1] == self.stride * 3
        B = input[InputDataKeys.POS].shape[0]
        ts = [input[k].to(device=device, dtype=torch.float32, non_blocking=True).contiguous() for k in INPUT_KEY_ORDER]
        x = self._get_plan(device).buf.get("ff.x", (B, self.input_size), self.compute_dtype)
        hip.concat_keys(ts, x)
        return x

    def _plan_forward(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty((x.shape[0], self.output_size), dtype=self.compute_dtype, device=x.device)
        if self.training:
            self._fwd_calls += 1           # dropout masks are keyed on (seed, step, element): a fresh draw per call
        return self._get_plan(x.device).forward(x, self.param_source(), out=out, training=self.training,
                                                step=self._fwd_calls)

    def _plan_backward(self, dout, P, accumulate):
        self._plan.backward(dout, P, accumulate)
        return None

    def split_output(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """views of the flat output: FeedForwardRegressionBaseline.py:116-121"""
        B, F = x.shape[0], self.num_output_frames
        return {
            OutputDataKeys.GROUND_CONTACT_COPS_IN_ROOT_FRAME: x[:, 0 * F:6 * F].reshape((B, F, 6)),
            OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME: x[:, 6 * F:12 * F].reshape((B, F, 6)),
            OutputDataKeys.GROUND_CONTACT_TORQUES_IN_ROOT_FRAME: x[:, 12 * F:18 * F].reshape((B, F, 6)),
            OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME: x[:, 18 * F:30 * F].reshape((B, F, 12)),
        }

    def forward(self, input: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        self.ensure_packed()
        dev = self._flat.device
        x = self.pack_inputs(input, dev)
        return self.split_output(self.run_plan(x))
