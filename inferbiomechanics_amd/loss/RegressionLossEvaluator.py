"""RegressionLossEvaluator on one fused gfx950 kernel pair (drop-in for src/loss/RegressionLossEvaluator.py).

Same constructor, ``__call__`` signature, running-list attributes, static helpers, wandb key scheme and
``print_report`` as the reference (:34-426).  What changes is HOW ``__call__`` computes: the reference
issues ~40 small kernels and 7 ``.item()`` device->host syncs per call (:184-263); here ONE launch pair
produces the 4 per-component MSE vectors, the CoP mask, the scalar loss, the six last-frame metrics and
d loss/d outputs, all into device buffers.  Metrics stay on the device and are read back only when a
report is printed / logged (deferred), so a training step has no host sync.

The four static helpers are kept as API (the reference's 24 unit tests pin them); they are plain tensor
expressions used for tests / ad-hoc analysis and are NOT on the hot path (``__call__`` never calls them).
"""
import argparse
import logging
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import hip
from ..data.AddBiomechanicsDataset import LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, InputDataKeys, OutputDataKeys

components = {0: "left-x", 1: "left-y", 2: "left-z", 3: "right-x", 4: "right-y", 5: "right-z"}
wrench_components = {
    0: "left-moment-x", 1: "left-moment-y", 2: "left-moment-z", 3: "left-force-x", 4: "left-force-y",
    5: "left-force-z", 6: "right-moment-x", 7: "right-moment-y", 8: "right-moment-z", 9: "right-force-x",
    10: "right-force-y", 11: "right-force-z"}


def component_weights(args: argparse.Namespace) -> List[float]:
    """0/1 (multiplicity) selection vector in kernel order force[6], cop[6], moment[6], wrench[12]
    (the index lists of RegressionLossEvaluator.py:217-220; a repeated index counts twice, as
    ``torch.sum(vec[idx])`` would)."""
    w = [0.0] * 30
    for i in args.predict_grf_components:
        w[0 + i] += 1.0
    for i in args.predict_cop_components:
        w[6 + i] += 1.0
    for i in args.predict_moment_components:
        w[12 + i] += 1.0
    for i in args.predict_wrench_components:
        w[18 + i] += 1.0
    return w


class _RegressionLossFn(torch.autograd.Function):
    """(cop, force, torque, wrench outputs) -> (loss, result[64]); grads are produced by the same launch."""

    @staticmethod
    def forward(ctx, o_cop, o_force, o_torque, o_wrench, labels, comp_w, threshold):
        outs = (o_cop, o_force, o_torque, o_wrench)
        B, F, _ = o_cop.shape
        dev, dt = o_cop.device, o_cop.dtype
        need_grad = any(o.requires_grad for o in outs)
        result = torch.zeros(64, dtype=torch.float32, device=dev)
        ws = torch.empty(hip.regression_loss_workspace_bytes(B, F), dtype=torch.uint8, device=dev)
        grads = None
        if need_grad:
            G = torch.empty((B, 30 * F), dtype=dt, device=dev)
            grads = (G[:, 0:6 * F].view(B, F, 6), G[:, 6 * F:12 * F].view(B, F, 6),
                     G[:, 12 * F:18 * F].view(B, F, 6), G[:, 18 * F:30 * F].view(B, F, 12))
        hip.regression_loss(outs, labels, comp_w, result, ws, grads=grads, threshold=threshold)
        ctx.grads = grads
        ctx.G = G if need_grad else None
        ctx.mark_non_differentiable(result)
        return result[0], result

    @staticmethod
    def backward(ctx, dloss, _dresult):
        if ctx.grads is None:
            return (None,) * 7
        # the four gradients are views of ONE [B, 30 F] buffer the forward launch filled: scale it in place by the
        # upstream gradient (a device scalar) with one HIP launch -- no ATen arithmetic on the backward
        # The buffer is consumed by that: a second backward over the same graph (retain_graph=True) would scale it again,
        # so it is refused, as torch refuses a second pass over freed saved tensors.
        from .DiffusionLossEvaluator import _as_f32_scalar
        if getattr(ctx, "consumed", False):
            raise RuntimeError("regression loss: backward was already run on this graph; its gradient buffer is scaled "
                               "in place and cannot be reused (call the evaluator again for a second backward)")
        ctx.consumed = True
        hip.scale_by_device_scalar(ctx.G, _as_f32_scalar(dloss))
        return tuple(ctx.grads) + (None, None, None)


class _SqDiffMean(torch.autograd.Function):
    """mean((o - l)^2, dim=(0, 1)) of device tensors through the library, differentiable w.r.t. the output and (the same
    gradient negated) the label; the result has the inputs' dtype, as the torch expression of the host path has"""

    @staticmethod
    def forward(ctx, o, l):
        ctx.save_for_backward(o, l)
        return hip.sqdiff_mean(o, l).to(o.dtype)

    @staticmethod
    def backward(ctx, dout):
        o, l = ctx.saved_tensors
        g = hip.sqdiff_mean_bwd(o, l, dout.detach().to(torch.float32).contiguous())
        return g, (-g if ctx.needs_input_grad[1] else None)


class RegressionLossEvaluator:
    def __init__(self, dataset, split: str, device='cpu'):
        self.dataset = dataset
        self.split = split
        self.losses: List[torch.Tensor] = []
        self.force_losses: List[torch.Tensor] = []
        self.moment_losses: List[torch.Tensor] = []
        self.wrench_losses: List[torch.Tensor] = []
        self.cop_losses: List[torch.Tensor] = []
        # per-call metric records; device scalars until a report needs them (deferred readback)
        self.force_reported_metrics: List = []
        self.moment_reported_metrics: List = []
        self.cop_reported_metrics: List = []
        self.wrench_reported_metrics: List = []
        self.wrench_moment_reported_metrics: List = []
        self.tau_reported_metrics: List = []
        self.com_acc_reported_metrics: List = []
        self.device = device
        self._comp_w_cache = {}
        self._warned_wandb = False

    # ---- static helpers (src/loss/RegressionLossEvaluator.py:73-158) ------------------------------------------------
    # Same checks and messages as the reference.  Tensors in HBM go through the library (ib_sqdiff_mean / ib_mask_by_threes /
    # ib_mean_norm_error; `get_squared_diff_mean_vector` stays differentiable through ib_sqdiff_mean_bwd).  HOST tensors --
    # what the reference's own unit tests pass -- take the explicit host path below: a few torch expressions on the CPU,
    # an API shim outside the GPU hot path (the training step never calls these; it uses the fused ib_regression_loss).
    @staticmethod
    def _metric_no_grad(*tensors):
        """the reported metrics (mean norm errors) are read, never differentiated: the library computes them without a
        graph.  A caller that wants gradients through them on the device is told so instead of silently getting a constant
        (the host path, like the reference statics, is differentiable)."""
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            raise hip.HipError("get_mean_norm_error / get_com_acc_error on device tensors are metrics computed without an "
                               "autograd graph: detach the inputs (or call under torch.no_grad()); the differentiable loss "
                               "terms are get_squared_diff_mean_vector and the evaluator's __call__")

    @staticmethod
    def _on_device(*tensors) -> bool:
        return all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors) and \
            all(t.dtype in (torch.float32, torch.bfloat16) for t in tensors) and len({t.dtype for t in tensors}) == 1

    @staticmethod
    def get_squared_diff_mean_vector(output_tensor: torch.Tensor, label_tensor: torch.Tensor) -> torch.Tensor:
        if output_tensor.shape != label_tensor.shape:
            raise ValueError('Output and label tensors must have the same shape')
        if len(output_tensor.shape) != 3:
            raise ValueError('Output and label tensors must be 3-dimensional')
        if output_tensor.numel() == 0:
            raise ValueError('Output and label tensors must not be empty')
        if RegressionLossEvaluator._on_device(output_tensor, label_tensor):
            return _SqDiffMean.apply(output_tensor.contiguous(), label_tensor.contiguous())
        return torch.mean((output_tensor - label_tensor) ** 2, dim=(0, 1))          # host tensors

    @staticmethod
    def get_mask_by_threes(tensor: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
        with torch.no_grad():
            if len(tensor.shape) != 3:
                raise ValueError('Mask tensor must be 3-dimensional')
            if tensor.numel() == 0:
                raise ValueError('Mask tensor must not be empty')
            if tensor.shape[-1] % 3 != 0:
                raise ValueError('Mask tensor must have a final dimension divisible by 3')
            if RegressionLossEvaluator._on_device(tensor):
                return hip.mask_by_threes(tensor.contiguous(), threshold)
            norms = torch.norm(tensor.reshape(tensor.shape[0], tensor.shape[1], -1, 3), dim=-1)       # host tensors
            mask = (norms > threshold).to(torch.float32)
            return mask.unsqueeze(3).expand(-1, -1, -1, 3).reshape(tensor.shape)

    @staticmethod
    def get_mean_norm_error(output_tensor: torch.Tensor, label_tensor: torch.Tensor, vec_size: int = 3) -> torch.Tensor:
        if output_tensor.shape != label_tensor.shape:
            raise ValueError('Output and label tensors must have the same shape')
        if len(output_tensor.shape) != 3:
            raise ValueError('Output and label tensors must be 3-dimensional')
        if output_tensor.numel() == 0:
            raise ValueError('Output and label tensors must not be empty')
        if output_tensor.shape[-1] % vec_size != 0:
            raise ValueError('Tensors must have a final dimension divisible by vec_size=' + str(vec_size))
        if RegressionLossEvaluator._on_device(output_tensor, label_tensor):
            RegressionLossEvaluator._metric_no_grad(output_tensor, label_tensor)
            with torch.no_grad():
                return hip.mean_norm_error(output_tensor.contiguous(), label_tensor.contiguous(), vec_size)
        diffs = output_tensor - label_tensor                                                          # host tensors
        last = diffs.reshape(diffs.shape[0], diffs.shape[1], -1, vec_size)[:, -1:, :, :]   # last frame only (:136)
        return torch.mean(torch.norm(last, dim=3))

    @staticmethod
    def get_com_acc_error(output_force_tensor: torch.Tensor, label_force_tensor: torch.Tensor) -> torch.Tensor:
        if output_force_tensor.shape != label_force_tensor.shape:
            raise ValueError('Output and label tensors must have the same shape')
        if len(output_force_tensor.shape) != 3:
            raise ValueError('Output and label tensors must be 3-dimensional')
        if output_force_tensor.numel() == 0:
            raise ValueError('Output and label tensors must not be empty')
        if output_force_tensor.shape[-1] != 6:
            raise ValueError('Output and label tensors must have a 6 dimensional final dimension')
        if RegressionLossEvaluator._on_device(output_force_tensor, label_force_tensor):
            RegressionLossEvaluator._metric_no_grad(output_force_tensor, label_force_tensor)
            with torch.no_grad():
                return hip.mean_norm_error(output_force_tensor.contiguous(), label_force_tensor.contiguous(), 3,
                                           fold_halves=True)
        o = output_force_tensor[:, :, :3] + output_force_tensor[:, :, 3:]                             # host tensors
        l = label_force_tensor[:, :, :3] + label_force_tensor[:, :, 3:]
        return RegressionLossEvaluator.get_mean_norm_error(o, l, vec_size=3)

    # ---- the hot path -----------------------------------------------------------------------------
    def _device(self) -> torch.device:
        d = self.device
        if isinstance(d, int):
            return torch.device('cuda', d)
        d = torch.device('cuda' if d == 'gpu' else d)
        if d.type != 'cuda' and not hip._dry_run:
            raise hip.HipError("RegressionLossEvaluator.__call__ runs on the GPU only (device=%r): construct it "
                               "with the model's device. There is no CPU fallback." % (self.device,))
        return d

    def _comp_w(self, args, dev) -> torch.Tensor:
        key = (tuple(args.predict_grf_components), tuple(args.predict_cop_components),
               tuple(args.predict_moment_components), tuple(args.predict_wrench_components), str(dev))
        t = self._comp_w_cache.get(key)
        if t is None:
            t = torch.tensor(component_weights(args), dtype=torch.float32, device=dev)
            self._comp_w_cache[key] = t
        return t

    def __call__(self, inputs: Dict[str, torch.Tensor], outputs: Dict[str, torch.Tensor],
                 labels: Dict[str, torch.Tensor], batch_subject_indices: List[int], batch_trial_indices: List[int],
                 args: argparse.Namespace, compute_report: bool = False, log_reports_to_wandb: bool = False,
                 analyze: bool = False, plot_path_root: str = 'outputs/plots') -> torch.Tensor:
        dev = self._device()
        # the reference moves (and REPLACES, in the caller's dicts) labels and outputs (:177-181)
        for key in list(labels.keys()):
            labels[key] = labels[key].to(dev, non_blocking=True)
        for key in list(outputs.keys()):
            outputs[key] = outputs[key].to(dev)
        outs = tuple(outputs[k] for k in LOSS_KEY_ORDER)
        labs = tuple(labels[k].to(torch.float32).contiguous() for k in LOSS_KEY_ORDER)
        comp_w = self._comp_w(args, dev)
        loss, result = _RegressionLossFn.apply(*outs, labs, comp_w, 10.0)

        self.force_losses.append(result[1:7])
        self.cop_losses.append(result[7:13])
        self.moment_losses.append(result[13:19])
        self.wrench_losses.append(result[19:31])
        self.losses.append(loss)
        self.force_reported_metrics.append(result[31])
        self.moment_reported_metrics.append(result[32])
        self.cop_reported_metrics.append(result[33])
        self.wrench_reported_metrics.append(result[34])
        self.wrench_moment_reported_metrics.append(result[35])
        self.com_acc_reported_metrics.append(result[36])

        tau_reported_metric: Optional[float] = None
        if compute_report:
            tau_reported_metric = self._inverse_dynamics_report(inputs, outputs, labels, batch_subject_indices)
            self.tau_reported_metrics.append(tau_reported_metric)

        if log_reports_to_wandb:
            r = result.detach().cpu()
            self.log_to_wandb(args, r[1:7], r[7:13], r[13:19], r[19:31], r[0], float(r[31]), float(r[33]),
                              float(r[32]), float(r[36]), float(r[34]), tau_reported_metric)

        if analyze:
            self._plot_force_error(outputs, labels, args, batch_subject_indices, batch_trial_indices, plot_path_root)
        return loss

    def record_result(self, result: torch.Tensor):
        """append one step's kernel result vector [64] (the fused trainer computes the loss itself and
        hands over the device buffer; cloned, not read back)"""
        r = result.detach().clone()
        self.losses.append(r[0])
        self.force_losses.append(r[1:7]); self.cop_losses.append(r[7:13])
        self.moment_losses.append(r[13:19]); self.wrench_losses.append(r[19:31])
        self.force_reported_metrics.append(r[31]); self.moment_reported_metrics.append(r[32])
        self.cop_reported_metrics.append(r[33]); self.wrench_reported_metrics.append(r[34])
        self.wrench_moment_reported_metrics.append(r[35]); self.com_acc_reported_metrics.append(r[36])

    def _inverse_dynamics_report(self, inputs, outputs, labels, batch_subject_indices) -> float:
        """per-element nimble inverse dynamics (RegressionLossEvaluator.py:265-286) -- needs the
        nimblephysics skeletons of a real AddBiomechanicsDataset; out of the hot path."""
        if self.dataset is None or not hasattr(self.dataset, 'skeletons'):
            raise NotImplementedError("compute_report=True needs nimblephysics skeletons (dataset.skeletons); "
                                      "not available for synthetic windows")
        num = outputs[OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME].shape[0]
        total = 0.0
        for b in range(num):
            skel = self.dataset.skeletons[batch_subject_indices[b]]
            skel.setPositions(inputs[InputDataKeys.POS][b, -1, :].cpu().numpy())
            skel.setVelocities(inputs[InputDataKeys.VEL][b, -1, :].cpu().numpy())
            acc = inputs[InputDataKeys.ACC][b, -1, :].cpu().numpy()
            bodies = self.dataset.skeletons_contact_bodies[batch_subject_indices[b]]
            guess = outputs[OutputDataKeys.GROUND_CONTACT_WRENCHES_IN_ROOT_FRAME][b, -1, :].detach().float().cpu().numpy() \
                * skel.getMass()
            tau = skel.getInverseDynamicsFromPredictions(acc, bodies, [guess[i * 6:i * 6 + 6] for i in range(len(bodies))],
                                                         np.zeros(6))
            err = tau - labels[OutputDataKeys.TAU][b, -1, :].cpu().numpy()
            total += np.mean(np.abs(err[6:])) / skel.getMass()
        return total / num

    def _plot_force_error(self, outputs, labels, args, subj, trial, root):
        import matplotlib.pyplot as plt
        k = OutputDataKeys.GROUND_CONTACT_FORCES_IN_ROOT_FRAME
        self.plot_ferror = ((outputs[k].float() - labels[k]) ** 2)[:, -1, :].reshape(-1, 6).detach().cpu().numpy()
        for i in args.predict_grf_components:
            plt.clf()
            plt.plot(self.plot_ferror[:, i])
            name = f"window{subj[0] if len(subj) else 0}_{trial[0] if len(trial) else 0}_grferror{components[i]}.png"
            os.makedirs(root, exist_ok=True)
            plt.savefig(os.path.join(root, name))

    # ---- reporting (key scheme of RegressionLossEvaluator.py:342-366) ------------------------------
    def log_to_wandb(self, args, force_loss, cop_loss, moment_loss, wrench_loss, loss, force_reported_metric,
                     cop_reported_metric, moment_reported_metric, com_acc_reported_metric, wrench_reported_metric,
                     tau_reported_metric):
        report = self.build_report(args, force_loss, cop_loss, moment_loss, wrench_loss, loss, force_reported_metric,
                                   cop_reported_metric, moment_reported_metric, com_acc_reported_metric,
                                   wrench_reported_metric, tau_reported_metric)
        try:
            import wandb
        except ImportError:
            if not self._warned_wandb:
                logging.warning("wandb is not installed; reports are not uploaded")
                self._warned_wandb = True
            return report
        wandb.log(report)
        return report

    def build_report(self, args, force_loss, cop_loss, moment_loss, wrench_loss, loss, force_reported_metric,
                     cop_reported_metric, moment_reported_metric, com_acc_reported_metric, wrench_reported_metric,
                     tau_reported_metric) -> Dict[str, float]:
        f = lambda v: float(v)
        report: Dict[str, float] = {
            **{f'{self.split}/force_rmse/{components[i]}': f(force_loss[i]) ** 0.5 for i in args.predict_grf_components},
            **{f'{self.split}/cop_rmse/{components[i]}': f(cop_loss[i]) ** 0.5 for i in args.predict_cop_components},
            **{f'{self.split}/moment_rmse/{components[i]}': f(moment_loss[i]) ** 0.5
               for i in args.predict_moment_components},
            **{f'{self.split}/wrench_loss/{wrench_components[i]}': f(wrench_loss[i]) ** 0.5
               for i in args.predict_wrench_components},
            f'{self.split}/loss': f(loss)}
        if force_reported_metric is not None:
            report[f'{self.split}/reports/Force Avg Err (N per kg)'] = force_reported_metric
        if cop_reported_metric is not None:
            report[f'{self.split}/reports/CoP Avg Err (m)'] = cop_reported_metric
        if moment_reported_metric is not None:
            report[f'{self.split}/reports/Moment Avg Err (Nm per kg)'] = moment_reported_metric
        if com_acc_reported_metric is not None:
            report[f'{self.split}/reports/COM Acc Avg Err (m per s^2)'] = com_acc_reported_metric
        if wrench_reported_metric is not None:
            report[f'{self.split}/reports/Wrench Avg Err (N+Nm per kg)'] = wrench_reported_metric
        if tau_reported_metric is not None:
            report[f'{self.split}/reports/Non-root Joint Torques (Inverse Dynamics) Avg Err (Nm per kg)'] = \
                tau_reported_metric
        return report

    @staticmethod
    def _mean(lst) -> Optional[float]:
        if len(lst) == 0:
            return None
        if isinstance(lst[0], torch.Tensor):
            return float(torch.stack([t.detach().reshape(()) for t in lst]).mean().cpu())   # ONE readback
        return float(np.mean(lst))

    def metric_means(self) -> Dict[str, Optional[float]]:
        return {"force": self._mean(self.force_reported_metrics), "moment": self._mean(self.moment_reported_metrics),
                "cop": self._mean(self.cop_reported_metrics), "wrench": self._mean(self.wrench_reported_metrics),
                "wrench_moment": self._mean(self.wrench_moment_reported_metrics),
                "tau": self._mean(self.tau_reported_metrics), "com_acc": self._mean(self.com_acc_reported_metrics)}

    def print_report(self, args: Optional[argparse.Namespace] = None, reset: bool = True, log_to_wandb: bool = False):
        m = self.metric_means()
        if log_to_wandb and len(self.force_losses) > 0:
            assert (args is not None)
            agg = lambda lst: torch.mean(torch.vstack([t.detach() for t in lst]), dim=0).cpu()
            self.log_to_wandb(args, agg(self.force_losses), agg(self.cop_losses), agg(self.moment_losses),
                              agg(self.wrench_losses), torch.mean(torch.hstack([l.detach() for l in self.losses])).cpu(),
                              m["force"], m["cop"], m["moment"], m["com_acc"], m["wrench"], m["tau"])
        if m["force"] is not None:
            print(f'\tForce Avg Err: {m["force"]} N / kg')
            print(f'\tCOM Acc Avg Err: {m["com_acc"]} m / s^2')
            print(f'\tCoP Avg Err: {m["cop"]} m')
            print(f'\tMoment Avg Err: {m["moment"]} Nm / kg')
            print(f'\tWrench Avg Err: {m["wrench"]} N+Nm / kg')
            print(f'\tWrench Moment Avg Err: {m["wrench_moment"]} Nm / kg')
            print(f'\tNon-root Joint Torques (Inverse Dynamics) Avg Err: {m["tau"]} Nm / kg')
        if reset:
            self.losses, self.force_losses, self.moment_losses = [], [], []
            self.wrench_losses, self.cop_losses = [], []
            self.force_reported_metrics, self.moment_reported_metrics, self.cop_reported_metrics = [], [], []
            self.wrench_reported_metrics, self.wrench_moment_reported_metrics = [], []   # reference forgets this one
            self.tau_reported_metrics, self.com_acc_reported_metrics = [], []
        return m
