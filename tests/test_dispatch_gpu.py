"""Kernel selection is ~30 thresholds inside csrc (NT from 640 rows, small-M tiles up to 64 tiles, K-split Linear+LayerNorm
below 4096 rows, ...).  This test pins, for every GEMM-like launch of every BASELINE workload (MLP / transformer training
step at B = 256, T = 50; the T = 200 DDIM step at B = 1 / 16 / 256; the reference-shape fp32 regression step), the kernel
FAMILY the build dispatches to (ib_debug_last_path) against tests/golden/dispatch_table.json (written by
tools/dispatch_table.py on the GPU box and reviewed): a threshold edit that moves a benchmarked shape to another kernel
fails here instead of silently changing a benchmark.  -m gpu."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_every_baseline_shape_takes_its_kernel_family(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    for k in ("IB_NO_NT", "IB_NO_TN", "IB_NO_RING", "IB_NO_SMALLM", "IB_NO_CHAIN", "IB_CHAIN_V1", "IB_NT_MIN_M",
              "IB_SMALLM_TILES", "IB_LINLN_MAX_M", "IB_LINLN_K512_MAX_M", "IB_TN_TARGET", "IB_NO_LINEAR_LN", "IB_NO_FFN_CHAIN"):
        assert k not in os.environ, f"{k} is set: the dispatch table describes the default build"
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools.dispatch_table import collect
    want = json.load(open(os.path.join(golden_dir, "dispatch_table.json")))
    got = collect()
    assert sorted(got) == sorted(want)
    for wl in want:
        g = {(e, tuple(d)): f for e, d, f in got[wl]}
        w = {(e, tuple(d)): f for e, d, f in want[wl]}
        assert g == w, (wl, {k: (g.get(k), w.get(k)) for k in set(g) | set(w) if g.get(k) != w.get(k)})
    # the headline shapes, spelled out: the chain kernel, the NT kernel for the transformer's M = 12800 GEMMs, TN for the
    # grouped weight gradients
    fam = lambda wl: {f for _, _, f in got[wl]}
    assert "chain_v2" in fam("mlp_denoiser_T50_B256_bf16_train_step")
    assert {"nt256x128", "tn256x128", "ffn_chain"} <= fam("transformer_denoiser_T50_B256_bf16_train_step")
    assert "smallm" in fam("transformer_denoiser_T200_B1_bf16_ddim_step")
    assert {"smallm", "wgrad_small"} <= fam("feedforward_ref_shape_B4_fp32_train_step")     # fp32: csrc/gemm_f32_small.hip
