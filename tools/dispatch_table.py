#!/usr/bin/env python3
"""Which kernel family every GEMM-like launch of the BASELINE workloads takes (csrc dispatch: NT / TN / ring / generic /
small-M / chain ...).  Prints the table as JSON; `--write` stores it as tests/golden/dispatch_table.json, the table
tests/test_dispatch_gpu.py holds the build to (a threshold edit that moves a benchmarked shape to another kernel then fails
a test instead of silently changing a benchmark).  GPU only."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from inferbiomechanics_amd import hip  # noqa: E402

GEMM_LIKE = ("ib_linear_", "ib_mlp_chain_train", "ib_time_mlp", "ib_mlp_chain_prep", "ib_ffn_chain_fwd", "ib_ffn_chain_bwd",
             "ib_ffn_infer_fwd")      # ("ib_linear_" covers ib_linear_panel_fwd / ib_linear_ln_panel_fwd)


def _rows(rec):
    """[(entry, small-integer arguments (shapes), family)] of the dispatching launches, de-duplicated, in first-use order"""
    out, seen = [], set()
    for (name, args), path in zip(rec.calls, rec.paths):
        if not name.startswith(GEMM_LIKE) or path == 0:
            continue
        dims = [v for v in args if isinstance(v, int) and not isinstance(v, bool) and 0 < v < (1 << 24)]
        key = (name, tuple(dims), path)
        if key not in seen:
            seen.add(key)
            out.append([name, dims, hip.PATH_NAMES[path]])
    return out


def collect():
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    from inferbiomechanics_amd.engine import HipTrainer
    dev = torch.device("cuda", 0)
    table = {}
    for wl in ("mlp_denoiser_T50", "transformer_denoiser_T50"):          # BASELINE configs[1], configs[2] (= [3] per rank)
        kind, T, D, B = bench.WORKLOADS[wl]
        model = bench.build_model(kind, T, D, torch.bfloat16, dev)
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4, use_graph=False)
        batches = bench.make_batches(1, B, T, D, torch.bfloat16, dev, seed=0)
        tr.step(batches[0])
        with hip.record_launches() as rec:
            tr.step(batches[0])
            torch.cuda.synchronize()
        table[f"{wl}_B{B}_bf16_train_step"] = _rows(rec)
        del tr, model
    for B in (1, 16, 256):                                               # configs[4]: T = 200, one DDIM denoise step
        model = bench.build_model("transformer", 200, 300, torch.bfloat16, dev)
        sampler = DDIMSampler(model, 100, use_graph=False)
        xT = torch.randn(B, 200, 300, device=dev)
        sampler.sample(xT, steps=2)
        with hip.record_launches() as rec:
            sampler.sample(xT, steps=2)
            torch.cuda.synchronize()
        table[f"transformer_denoiser_T200_B{B}_bf16_ddim_step"] = _rows(rec)
        del sampler, model
    # configs[0]: the reference-shape regression step, fp32, B = 4 and 64
    import argparse
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, input_key_widths
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    targs = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                               predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    for B in (4, 64):
        m = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[512, 512], device=dev)
        m.train()
        inputs = {k: torch.randn(B, 10, w, device=dev) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, 15))}
        labels = {k: torch.randn(B, 10, c, device=dev) for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}
        tr = HipTrainer(m, "regression", "rmsprop", 1e-4, args=targs, use_graph=False)
        tr.step((inputs, labels))
        with hip.record_launches() as rec:
            tr.step((inputs, labels))
            torch.cuda.synchronize()
        table[f"feedforward_ref_shape_B{B}_fp32_train_step"] = _rows(rec)
        del tr, m
    return table


if __name__ == "__main__":
    t = collect()
    txt = json.dumps(t, indent=1)
    if "--write" in sys.argv:
        dst = os.path.join(ROOT, "gpurun_out", "dispatch_table.json")     # copy into tests/golden/ after review
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        open(dst, "w").write(txt + "\n")
    print(txt)
