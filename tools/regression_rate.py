"""Replayed step time of the fused regression training step (feedforward / groundlink) against the sum of its kernels.
Usage (GPU box): python tools/regression_rate.py [model] [B] [F]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from inferbiomechanics_amd.data.AddBiomechanicsDataset import (INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS,  # noqa: E402
                                                               input_key_widths)
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "feedforward"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda", 0)
    targs = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                               predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    torch.manual_seed(0)
    if kind == "groundlink":
        from inferbiomechanics_amd.models.Groundlink import Groundlink
        m = Groundlink(23, 12, 10, "all_frames", device=dev, compute_dtype=torch.bfloat16)
        hw = 30
    else:
        from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
        m = FeedForwardBaseline(23, 2, 5 * F, "all_frames", "sigmoid", 5, 10, device=dev, compute_dtype=torch.bfloat16)
        hw = 15
    m.train()
    inputs = {k: torch.randn(B, F, w, device=dev) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, hw))}
    labels = {k: torch.randn(B, F, c, device=dev) for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}
    tr = HipTrainer(m, "regression", "rmsprop", 1e-4, args=targs, use_graph=True)
    for _ in range(20):
        tr.step((inputs, labels))
    torch.cuda.synchronize()
    n = 2000
    for own in (False, True):
        import contextlib
        ctx = torch.cuda.stream(tr.stream) if own and tr.stream is not None else contextlib.nullcontext()
        torch.cuda.synchronize()
        with ctx:
            t0 = time.perf_counter()
            for _ in range(n):
                tr.step((inputs, labels))
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(f"{kind} B={B} F={F} ({'on the trainer stream' if own else 'from the default stream'}): {el / n * 1e6:.1f} us per step "
              f"({B * n / el:.0f} windows/s), loss {tr.loss_value():.4f}")


if __name__ == "__main__":
    main()
