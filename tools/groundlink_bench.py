"""Groundlink training-step rate on the HIP path: fused trainer (captured graph) over a device
window cache, per dtype / window length; prints the per-kernel launch counts of one step.
usage: python tools/groundlink_bench.py [--batch 256] [--frames 10] [--steps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, nargs="+", default=[10, 50])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dtype", choices=["both", "fp32", "bf16"], default="both")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (rocprofv3 --kernel-trace lists them)")
    a = ap.parse_args()
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.Groundlink import Groundlink
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, \
        input_key_widths
    dev = torch.device("cuda:0")
    targs = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                               predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    for F in a.frames:
        for dt in (torch.float32, torch.bfloat16):
            if (a.dtype == "fp32" and dt != torch.float32) or (a.dtype == "bf16" and dt != torch.bfloat16):
                continue
            torch.manual_seed(0)
            m = Groundlink(23, 12, 10, "all_frames", device=dev, compute_dtype=dt)
            m.train()
            tr = HipTrainer(m, "regression", "adam", 1e-4, args=targs, use_graph=not a.no_graph)
            tr.adopt_stream()             # as cli/train.py does: the loop's device work on the trainer's stream
            B = a.batch
            inputs = {k: torch.randn(B, F, w, device=dev) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, 30))}
            labels = {k: torch.randn(B, F, c, device=dev) for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}
            for _ in range(5):
                tr.step((inputs, labels))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                tr.step((inputs, labels))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            # flops: forward GEMMs x3 (fwd, dgrad, wgrad; the first conv has no dgrad)
            feats = [m.channels, 128, 128, 256, 256]
            M = B * F
            fl = sum(2 * M * ci * 7 * co for ci, co in zip(feats[:-1], feats[1:])) + 2 * (2 * M * 256 * 256) + 2 * M * 256 * 30
            print(f"groundlink B={B} F={F} {str(dt).split('.')[-1]:8s}: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} windows/s  "
                  f"loss {tr.loss_value():.4f}  ~{3 * fl / ms / 1e9:.1f} TFLOP/s", flush=True)
            # the same step fed from the on-device window cache (one gather launch instead of 14 staging copies + concat)
            import numpy as np
            from inferbiomechanics_amd.data.WindowCache import DeviceWindowCache, PackedWindows
            widths = input_key_widths(23, 30)
            nwin = 4 * B
            x_el = F * sum(widths)
            x_pad = (x_el + 3) // 4 * 4
            rows = np.random.default_rng(0).standard_normal((nwin, x_pad + F * 30), dtype=np.float32)
            cache = DeviceWindowCache(PackedWindows(rows, F, F, widths), dev)
            idx = list(cache.batches(B))
            for i in range(8):
                tr.step_windows(cache, idx[i % len(idx)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                tr.step_windows(cache, idx[i % len(idx)])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            print(f"           ... from the window cache: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} windows/s", flush=True)


if __name__ == "__main__":
    main()
