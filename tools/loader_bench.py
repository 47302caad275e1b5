"""Input-pipeline comparison for the reference regression model: windows/s of the fused trainer
fed by (a) a DataLoader over in-memory reference-layout tuples (what PickledDataset gives the reference: collate of 14
small tensors per window + H2D copies + the concat kernel) and (b) the on-device window cache (one gather launch).
Usage (GPU box): python tools/loader_bench.py [windows] [batch]"""
import argparse
import sys
import time

import torch
from torch.utils.data import DataLoader

sys.path.insert(0, ".")
from inferbiomechanics_amd.data.AddBiomechanicsDataset import SyntheticWindowDataset  # noqa: E402
from inferbiomechanics_amd.data.WindowCache import DeviceWindowCache, PackedWindows  # noqa: E402
from inferbiomechanics_amd.engine import HipTrainer  # noqa: E402
from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                              predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    # in-memory reference-layout tuples = what a loaded `pickle-data` block holds (generated in bulk: the per-item
    # synthetic generator is itself ~1 ms per window)
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import (INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS,
                                                                   input_key_widths)
    g = torch.Generator().manual_seed(1)
    ins = {k: torch.randn(n, 10, w, generator=g) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, 15))}
    labs = {k: 5.0 * torch.randn(n, 10, c, generator=g) for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}
    windows = [({k: v[i] for k, v in ins.items()}, {k: v[i] for k, v in labs.items()}, 0, i) for i in range(n)]
    print(f"{n} windows generated", flush=True)

    def trainer():
        m = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, device=dev, compute_dtype=torch.bfloat16)
        t = HipTrainer(m, "regression", "rmsprop", 1e-4, args=args)
        t.adopt_stream()                  # as cli/train.py does: the loop's device work on the trainer's stream
        return t

    for workers in (0, 4):
        tr = trainer()
        dl = DataLoader(windows, batch_size=B, shuffle=False, drop_last=True, num_workers=workers, pin_memory=True,
                        persistent_workers=workers > 0)
        for epoch in range(2):                       # epoch 0 warms up (graph capture, worker start)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for inputs, labels, _, _ in dl:
                tr.step((inputs, labels))
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(f"DataLoader(workers={workers}) over in-memory tuples: {(n // B) * B / el:10.0f} windows/s", flush=True)
    t0 = time.perf_counter()
    pack = PackedWindows.from_windows(windows)
    cache = DeviceWindowCache(pack, dev)
    torch.cuda.synchronize()
    print(f"pack + upload of {n} windows ({cache.table.numel() * 4 / 2**20:.0f} MiB): {time.perf_counter() - t0:.2f} s (once)")
    tr = trainer()
    for epoch in range(3):
        batches = list(cache.batches(B))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for idx in batches:
            tr.step_windows(cache, idx)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"on-device window cache:                          {len(batches) * B / el:10.0f} windows/s "
          f"({el / len(batches) * 1e3:.3f} ms per step of {B})")


if __name__ == "__main__":
    main()
