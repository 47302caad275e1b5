"""Per-launch breakdown of one fused regression training step (feedforward / groundlink): every distinct C-ABI call of
an eager step is re-timed back-to-back inside a hipGraph.  Usage (GPU box): python tools/regression_prof.py [model] [B] [F] [f32|bf16]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from inferbiomechanics_amd import hip  # noqa: E402
from inferbiomechanics_amd.data.AddBiomechanicsDataset import (INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS,  # noqa: E402
                                                               input_key_widths)
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "feedforward"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    cdt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.bfloat16
    dev = torch.device("cuda", 0)
    targs = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                               predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    torch.manual_seed(0)
    if kind == "groundlink":
        from inferbiomechanics_amd.models.Groundlink import Groundlink
        m = Groundlink(23, 12, 10, "all_frames", device=dev, compute_dtype=cdt)
        hw = 30
    else:
        from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
        m = FeedForwardBaseline(23, 2, 5 * F, "all_frames", "sigmoid", 5, 10, device=dev, compute_dtype=cdt)
        hw = 15
    m.train()
    inputs = {k: torch.randn(B, F, w, device=dev) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, hw))}
    labels = {k: torch.randn(B, F, c, device=dev) for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}
    tr = HipTrainer(m, "regression", "rmsprop", 1e-4, args=targs, use_graph=False)
    for _ in range(3):
        tr.step((inputs, labels))
    torch.cuda.synchronize()
    with hip.record_launches() as rec:
        tr.step((inputs, labels))
        torch.cuda.synchronize()
    uniq = {}
    for name, args in rec.calls:
        key = (name, bench._ints(args))
        uniq.setdefault(key, [args, 0])[1] += 1
    rows = []
    for (name, ints), (args, count) in uniq.items():
        if name in hip._RecordingLib.SKIP:
            continue
        try:
            us = hip.time_recorded_call(name, args)
        except hip.HipError:                 # entries whose arguments hold host arrays cannot be replayed
            continue
        rows.append((us * count, name, count, us, ints[-5:]))
    rows.sort(reverse=True)
    print(f"{kind} B={B} F={F}: {len(rec.calls)} launches per step, sum of kernels {sum(r[0] for r in rows):.1f} us")
    for tot, name, count, us, ints in rows:
        print(f"{tot:8.1f} us  {count:2d} x {us:7.2f}  {name[3:]:24s} {list(ints)}")


if __name__ == "__main__":
    main()
