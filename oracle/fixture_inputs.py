"""Closed-form inputs / parameters shared by oracle/make_golden.py and tests/ (TEST INFRASTRUCTURE ONLY).

The golden fixtures store only expected outputs; inputs and weights are rebuilt from these
closed forms on both sides.
"""
import torch

from oracle.ref_cpu import (INPUT_KEY_ORDER, K_COP, K_FORCE, K_TORQUE, K_WRENCH, det_fill, input_widths)


def ff_inputs(B, F, num_dofs, stride, dtype=torch.float32):
    ws = input_widths(num_dofs, stride)
    return {k: det_fill((B, F, w), 10 + i, 1.0, dtype) for i, (k, w) in enumerate(zip(INPUT_KEY_ORDER, ws))}


def gl_inputs(B, F, num_dofs=23, root_history_len=10, dtype=torch.float32):
    """Groundlink inputs: history keys are root_history_len*3 wide (Groundlink.py:116-118), joint centers 12*3"""
    ws = [num_dofs, num_dofs, num_dofs, 3, 3, 3, 3, 36, root_history_len * 3, root_history_len * 3]
    return {k: det_fill((B, F, w), 70 + i, 1.0, dtype) for i, (k, w) in enumerate(zip(INPUT_KEY_ORDER, ws))}


GL_CASES = [("all_frames_F10", "all_frames", 10), ("last_frame_F10", "last_frame", 10), ("all_frames_F5", "all_frames", 5)]


def ff_labels(B, F, dtype=torch.float32):
    lab = {
        K_COP: det_fill((B, F, 6), 31, 0.3, dtype),
        K_FORCE: det_fill((B, F, 6), 32, 9.0, dtype),     # norms straddle the 10.0 CoP-mask threshold
        K_TORQUE: det_fill((B, F, 6), 33, 1.0, dtype),
        K_WRENCH: det_fill((B, F, 12), 34, 2.0, dtype),
    }
    # exact-threshold edge: ||(6,8,0)|| == 10.0 -> masked OUT (strict '>'), RegressionLossEvaluator.py:205-209
    lab[K_FORCE][0, 0, 0:3] = torch.tensor([6.0, 8.0, 0.0], dtype=dtype)
    lab[K_FORCE][0, 0, 3:6] = torch.tensor([6.0, 8.0, 0.01], dtype=dtype)
    return lab


def loss_case_outputs(B=5, F=7, dtype=torch.float32):
    return {K_COP: det_fill((B, F, 6), 41, 0.4, dtype), K_FORCE: det_fill((B, F, 6), 42, 8.0, dtype),
            K_TORQUE: det_fill((B, F, 6), 43, 1.2, dtype), K_WRENCH: det_fill((B, F, 12), 44, 2.2, dtype)}


def det_state(shapes, seed0=1.0):
    """{name: shape} (state_dict order) -> deterministic float64 state (cast by the caller)."""
    new = {}
    for i, (k, shp) in enumerate(shapes.items()):
        shp = tuple(shp)
        if len(shp) >= 2:
            new[k] = det_fill(shp, seed0 + i, 1.0 / (shp[-1] ** 0.5))
        elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight"):
            new[k] = 1.0 + det_fill(shp, seed0 + i, 0.1)
        else:
            new[k] = det_fill(shp, seed0 + i, 0.05)
    return new


FF_CASES = [("h50s5_sigmoid", 50, 5, "sigmoid"), ("h50s5_relu", 50, 5, "relu"),
            ("h50s5_tanh", 50, 5, "tanh"), ("h50s1_sigmoid", 50, 1, "sigmoid")]
TL_CASES = [("d512_T50", 512, 8, 2048, 2, 50, torch.float32),
            ("d512_T200", 512, 8, 2048, 1, 200, torch.float32),
            ("d108_T10_f64", 108, 3, 60, 2, 10, torch.float64),
            ("d128_T37", 128, 4, 256, 3, 37, torch.float32)]
LOSS_SUBSETS = {"train_default": (list(range(6)), list(range(6)), list(range(6)), list(range(12))),
                "analyze_default": ([1], [], [], []),
                "mixed": ([0, 2, 5], [1, 4], [3], [0, 6, 11])}


# ---- feedforward model with the optional layers (--batchnorm / --dropout, src/cli/train.py:43-47)
FF_OPT_HIDDEN = [64, 48]
FF_OPT_CASES = [("bn_train", True, False, True), ("bn_eval", True, False, False), ("bn_drop_eval", True, True, False),
                ("drop_eval", False, True, False)]          # (name, batchnorm, dropout, train mode)
FF_OPT_B, FF_OPT_P = 6, 0.3


def ff_opt_state(shapes, seed0=21.0):
    """deterministic state for a feedforward model with BatchNorm layers: Linear weights / biases as det_state;
    BatchNorm gamma around 1, running_var positive, num_batches_tracked = 3"""
    new = {}
    for i, (k, shp) in enumerate(shapes.items()):
        shp = tuple(shp)
        if k.endswith("num_batches_tracked"):
            new[k] = torch.tensor(3, dtype=torch.int64)
        elif k.endswith("running_var"):
            new[k] = 1.0 + 0.3 * det_fill(shp, seed0 + i, 1.0)
        elif k.endswith("running_mean"):
            new[k] = det_fill(shp, seed0 + i, 0.2)
        elif len(shp) >= 2:
            new[k] = det_fill(shp, seed0 + i, 1.0 / (shp[-1] ** 0.5))
        elif k.endswith(".weight"):                     # 1-D weight = BatchNorm gamma
            new[k] = 1.0 + det_fill(shp, seed0 + i, 0.1)
        else:
            new[k] = det_fill(shp, seed0 + i, 0.05)
    return new


# window-loader fixture (tests/golden/loader_windows.npz): (name, window_size, stride, output_data_format, dtype)
LOADER_CASES = [("w50s5_all", 50, 5, "all_frames", "float32"), ("w50s5_last", 50, 5, "last_frame", "float32"),
                ("w10s1_all", 10, 1, "all_frames", "float32"), ("w20s3_all_f64", 20, 3, "all_frames", "float64")]


def loader_sample(n):
    """indices of the windows stored whole: a spread over the index (all three subjects, first and last window)"""
    return sorted({int(round(j * (n - 1) / 5)) for j in range(6)})
