#!/usr/bin/env python3
"""Headline benchmark of the hot path (driver contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one fused training step (q_sample -> denoiser fwd -> eps-MSE -> bwd -> [RCCL grad all-reduce] ->
fused optimizer) over one batch of synthetic motion windows already resident in HBM.  Headline workload = the
largest single-GPU training configuration of BASELINE.json: configs[2], the 4-layer d_model = 512 transformer denoiser at
T = 50, D = 300, bf16 storage / fp32 accumulate, per-GPU batch 256; at N > 1 the same model data-parallel = configs[3]
(weak scaling: global batch 256 N).  The MLP denoiser (configs[1]), the DDIM loop (configs[4]) and the reference-shape
regression step (configs[0]) ride as extra keys; a compact `summary` object comes LAST in the line, so a truncated tail
still carries every headline figure.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PEAK_XGMI_GBS = 7 * 153.0                          # per GPU, all seven links

WORKLOADS = {
    # name: (model kind, T, D, per-GPU batch)
    "mlp_denoiser_T50": ("mlp", 50, 300, 256),
    "transformer_denoiser_T50": ("transformer", 50, 300, 256),
    "transformer_denoiser_T200": ("transformer", 200, 300, 64),
}


def build_model(kind, T, D, dtype, dev):
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    torch.manual_seed(0)
    if kind == "mlp":
        return DiffusionMLP(D, [512, 512], device=dev, compute_dtype=dtype)
    return DiffusionTransformer(D, T, d_model=512, num_heads=8, dim_feedforward=2048, num_layers=4, device=dev,
                                compute_dtype=dtype)


def train_flops_per_window(kind, T, D):
    """algorithmic FLOPs (2*MAC, training = 3x forward GEMM FLOPs), SURVEY.md §8d"""
    if kind == "mlp":
        mac = D * 512 + 512 * 512 + 512 * D
        return 3 * 2 * mac * T
    d, ffn, L = 512, 2048, 4
    per_tok = L * (4 * d * d + 2 * d * ffn + 2 * T * d) + (D + 30) * d + d * D
    return 3 * 2 * per_tok * T


def make_batches(nb, B, T, D, dtype, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for _ in range(nb):
        x0 = torch.randn(B, T, D, generator=g).to(dev, dtype)
        eps = torch.randn(B, T, D, generator=g).to(dev, dtype)
        t = torch.randint(0, 1000, (B,), generator=g, dtype=torch.int64).to(dev)
        out.append((x0, t, eps))
    return out


def cpu_baseline(kind, T, D, B, dev, budget_s=12.0, ncmp=None):
    """The oracle (torch-eager CPU port of the reference arithmetic + build-defined diffusion wrapper), fp32, one GPU's
    share of the host cores, same workload shape; bounded sample.  "At matched diffusion loss" (north_star): the CPU run and
    the fused GPU trainer (fp32 parity mode and the benchmarked bf16 mode) start from the SAME initial weights and take the
    SAME batches; the loss each computes at step `ncmp` is reported side by side."""
    from inferbiomechanics_amd.engine import HipTrainer
    from oracle import ref_cpu as R
    if ncmp is None:
        # the oracle's transformer step at batch 256 takes seconds on the host: the losses are compared after 3 steps and
        # the timed sample is bounded by `budget_s` (at least 2 steps); the MLP affords 20 + up to 200
        ncmp = 20 if kind == "mlp" else 3
    # one GPU's share of the host (the GPU box gives 16 worker CPUs per GPU); all 256 logical CPUs of the box
    # on these small GEMMs is 80x SLOWER (thread oversubscription: 9.8 windows/s measured)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = torch.Generator().manual_seed(1234)
    host = [(torch.randn(B, T, D, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randn(B, T, D, generator=g))
            for _ in range(4)]
    w0, gpu_loss = None, {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        model = build_model(kind, T, D, dt, dev)                     # torch.manual_seed(0) inside: same init both times
        if w0 is None:
            w0 = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        tr = HipTrainer(model, "diffusion", "rmsprop", 1e-4, use_graph=True)
        for i in range(ncmp + 1):
            x0, t, eps = host[i % len(host)]
            tr.step((x0.to(dev, dt), t.to(dev), eps.to(dev, dt)))
        gpu_loss[name] = tr.loss_value()                             # computed at step `ncmp`, before that step's update
        del tr, model
    params = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    if kind == "mlp":
        fwd = lambda p, x, t: R.denoiser_mlp_forward(p, x, t, [512, 512])
    else:
        fwd = lambda p, x, t: R.denoiser_transformer_forward(p, x, t, 4, 8)
    state = {k: R.optim_init_state("rmsprop", v.detach()) for k, v in params.items()}
    tabs = {k: v.to(torch.float32) for k, v in R.schedule_tables().items()}

    def step(i):
        x0, t, eps = host[i % len(host)]
        for v in params.values():
            v.grad = None
        xt = R.q_sample(x0, t, eps, tabs)
        loss = R.eps_mse(fwd(params, xt, t), eps)
        loss.backward()
        with torch.no_grad():
            for k, v in params.items():
                v.copy_(R.optim_step("rmsprop", v, v.grad, state[k], 1e-4, i + 1))
        return float(loss.detach())

    cpu_loss = None
    for i in range(ncmp + 1):
        cpu_loss = step(i)
    n, t0 = 0, time.perf_counter()
    while True:
        step(ncmp + 1 + n)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 2) or n >= 200:
            break
    rel = lambda a: round(abs(a - cpu_loss) / max(abs(cpu_loss), 1e-30), 6)
    return {"value": round(B * n / el, 1), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} training steps of batch {B} ({kind} denoiser, T={T}, D={D}, fp32, RMSprop) in {el:.1f} s "
                      f"(after {ncmp + 1} untimed matched-loss steps)",
            "matched_loss": {"step": ncmp, "cpu_oracle_f32": round(cpu_loss, 6), "gpu_f32": round(gpu_loss["f32"], 6),
                             "gpu_bf16": round(gpu_loss["bf16"], 6), "rel_diff_f32": rel(gpu_loss["f32"]),
                             "rel_diff_bf16": rel(gpu_loss["bf16"]),
                             "note": "same initial weights (seed 0), same 4 host batches cycled, RMSprop 1e-4; each side's "
                                     "loss as computed at this step"}}


def cli_path_leg(windows=1 << 18, epochs=3):
    """the SAME workload through the product surface: `main.py train --model-type diffusion-mlp --compute-dtype bf16
    --window-cache hbm` (tools/cli_rate.py): windows resident in HBM, x0 gather + t / eps drawn on the device per step"""
    from tools.cli_rate import cli_rate
    r = cli_rate(windows=windows, epochs=epochs)
    return {"windows_per_s": r["windows_per_s"], "epoch_windows_per_s": r["epoch_windows_per_s"],
            "steps_per_epoch": r["steps_per_epoch"], "command": r["command"], "note": r["note"]}


def regression_ref_shape_leg(dev, B, gpu_steps=400, cpu_budget_s=4.0):
    """BASELINE.json configs[0] / SURVEY.md §8d "Config 1": the reference's own model shape -- FeedForwardBaseline([512, 512],
    sigmoid), history 50 / stride 5 => 1470 -> 512 -> 512 -> 300 (src/models/FeedForwardRegressionBaseline.py:52,63), fp32,
    RMSprop lr 1e-4 (src/cli/train.py:41,190), the reference loss with every component (RegressionLossEvaluator.__call__) --
    one training step of the loop body src/cli/train.py:240-284: HipTrainer (fused, hipGraph-replayed, fp32 kernels) next to
    the oracle's CPU step, timed in the same run on the same batch from the same initial weights; the losses of both after
    the same number of steps are reported side by side ("matched loss trajectory")."""
    import argparse as _ap
    from inferbiomechanics_amd.data.AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, input_key_widths
    from inferbiomechanics_amd.engine import HipTrainer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from oracle import ref_cpu as R
    F = 10
    targs = _ap.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                          predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    torch.manual_seed(0)
    m = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[512, 512], device=dev,
                            compute_dtype=torch.float32)
    m.train()
    w0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    inputs = {k: torch.randn(B, F, w, generator=g) for k, w in zip(INPUT_KEY_ORDER, input_key_widths(23, 15))}
    labels = {k: torch.randn(B, F, c, generator=g) * (10.0 if "Force" in k or "FORCE" in k.upper() else 1.0)
              for k, c in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS)}          # force ~ 10 N(0,1): the CoP mask (> 10.0) is exercised
    din = {k: v.to(dev) for k, v in inputs.items()}
    dlab = {k: v.to(dev) for k, v in labels.items()}
    tr = HipTrainer(m, "regression", "rmsprop", 1e-4, args=targs, use_graph=True)
    prev = tr.adopt_stream()
    ncmp = 20
    for _ in range(ncmp):
        tr.step((din, dlab))
    gpu_loss = tr.loss_value()                       # loss of step `ncmp` (computed before that step's update)
    for _ in range(30):
        tr.step((din, dlab))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(gpu_steps):
        tr.step((din, dlab))
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / gpu_steps * 1e3
    # the number above is the IN-PLACE path: the same 14 device tensors come back every step (a loader recycling its device
    # buffers), so from the second appearance the step replays a graph that reads them where they lie.  A loader that makes
    # fresh device tensors every step takes the STAGED path (14 small copies into the trainer's static buffers): timed on a
    # second trainer over 64 distinct pre-generated batches, rotating.
    m2 = FeedForwardBaseline(23, 2, 50, "all_frames", "sigmoid", 5, 10, hidden_dims=[512, 512], device=dev,
                             compute_dtype=torch.float32)
    m2.train()
    tr2 = HipTrainer(m2, "regression", "rmsprop", 1e-4, args=targs, use_graph=True)
    ring = [({k: (v + 0.01 * j).to(dev) for k, v in inputs.items()}, {k: v.clone().to(dev) for k, v in labels.items()})
            for j in range(64)]
    for j in range(80):
        tr2.step(ring[j % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(gpu_steps):
        tr2.step(ring[j % 64])
    torch.cuda.synchronize()
    gpu_ms_staged = (time.perf_counter() - t0) / gpu_steps * 1e3
    del tr2, m2, ring
    if prev is not None:
        torch.cuda.set_stream(prev)
    # the oracle's CPU step: same weights, same batch, same optimizer arithmetic
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    params = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    layers = [(params[f"net.{2 * i}.weight"], params[f"net.{2 * i}.bias"]) for i in range(3)]
    state = {k: R.optim_init_state("rmsprop", v.detach()) for k, v in params.items()}

    def cpu_step(i):
        for v in params.values():
            v.grad = None
        out = R.feedforward_forward(layers, inputs, "sigmoid", F)
        loss, _, _ = R.regression_loss(out, labels, range(6), range(6), range(6), range(12))
        loss.backward()
        with torch.no_grad():
            for k, v in params.items():
                v.copy_(R.optim_step("rmsprop", v, v.grad, state[k], 1e-4, i + 1))
        return float(loss)

    cpu_loss = None
    for i in range(ncmp):
        cpu_loss = cpu_step(i)
    n, t1 = 0, time.perf_counter()
    while True:
        cpu_step(ncmp + n)
        n += 1
        el = time.perf_counter() - t1
        if el > cpu_budget_s or n >= 2000:
            break
    cpu_ms = el / n * 1e3
    del tr, m
    return {"batch": B, "gpu_ms_per_step": round(gpu_ms, 4), "gpu_windows_per_s": round(B / gpu_ms * 1e3, 1),
            "gpu_path_timed": "in place (the same device tensors recur: pinned graph, no staging copies)",
            "gpu_ms_per_step_staged": round(gpu_ms_staged, 4),
            "gpu_staged_note": "64 distinct device batches rotating: every batch is copied into the static buffers (14 copies)",
            "speedup_staged": round(cpu_ms / gpu_ms_staged, 1),
            "cpu_ms_per_step": round(cpu_ms, 3), "cpu_windows_per_s": round(B / cpu_ms * 1e3, 1),
            "cpu_cores": torch.get_num_threads(), "cpu_steps_timed": n, "gpu_steps_timed": gpu_steps,
            "speedup": round(cpu_ms / gpu_ms, 1),
            "loss_at_step_%d" % ncmp: {"gpu": round(gpu_loss, 6), "cpu_oracle": round(cpu_loss, 6),
                                       "rel_diff": round(abs(gpu_loss - cpu_loss) / max(abs(cpu_loss), 1e-30), 8)}}


KERNEL_OF = {   # C-ABI entry -> device kernel it launches (names as rocprofv3 --kernel-trace reports them)
    "ib_mlp_chain_train": "mlp_chain2_kernel<4, 3, 10, true>", "ib_mlp_chain_prep": "time_mlp_fwd_kernel<4, 4, 1> (+ weight packing blocks)",
    "ib_linear_wgrad_slabs": "gemm_ring_kernel<false, false, EPI_WGRAD>", "ib_slab_reduce_multi": "slab_reduce_multi_kernel",
    "ib_linear_wgrad_slabs_multi": "gemm_tn_kernel<false> (gemm_ring_wgrad_multi_kernel for short reductions)",
    "ib_linear_wgrad_slabs_multi_bias": "gemm_tn256w4_kernel<true> (gemm_tn_kernel<true> for shapes that are not multiples of 256)", "ib_linear_dgrad_wt": "gemm_nt_kernel",
    "ib_optim_step_sources": "optim_kernel<true>",
    "ib_step_reduce": "step_reduce_kernel",
    "ib_colsum_segments": "colsum_segments_kernel",
    "ib_linear_fwd": "gemm_kernel<T, true, true, EPI_FWD>", "ib_linear_dgrad": "gemm_kernel<T, true, false, EPI_DGRAD>",
    "ib_linear_wgrad": "gemm_kernel<T, false, false, EPI_WGRAD> (+ slab_reduce_kernel)",
    "ib_layernorm_fwd": "layernorm_fwd_kernel", "ib_layernorm_bwd": "layernorm_bwd_kernel (+ segment_colsum_kernel)",
    "ib_attention_fwd": "attn_fwd_mfma", "ib_attention_bwd": "attn_bwd_mfma", "ib_segment_colsum": "segment_colsum_kernel",
    "ib_ffn_chain_fwd": "ffn_chain_fwd_kernel<true, true, false> (<true, false, false> for the top layer)", "ib_ffn_chain_bwd": "ffn_chain_bwd_kernel<true, true, false> (<true, false, false> for the top layer)",
    "ib_ffn_chain_fwd_attn": "ffn_chain_fwd_kernel<true, true, true> (one-window panels: the layer's token-local half + the next layer's in-projection and attention; <true, false, false> for the top layer)",
    "ib_ffn_chain_bwd_attn": "ffn_chain_bwd_kernel<true, false, true> (one-window panels: the whole layer's backward incl. attention backward and in-projection dgrad)",
    "ib_ffn_chain_pack": "ffn_pack_kernel", "ib_diffusion_draw": "diffusion_draw_kernel",
    "ib_mse_loss": "mse_partial_kernel (+ mse_final_kernel)", "ib_q_sample": "q_sample_kernel",
    "ib_gather_rows": "gather_rows_kernel", "ib_cast": "cast2d_kernel", "ib_cast2d": "cast2d_kernel"}


def _ints(args):
    return tuple(v for v in args if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 31))


def record_eager_step(trainer, batches):
    """one EAGER training step with every C-ABI call recorded.  Under data parallelism the step contains the gradient
    all-reduce, a collective: EVERY rank must run this (only rank 0 uses the recording)."""
    from inferbiomechanics_amd import hip
    saved = trainer.use_graph, trainer._rec
    trainer.use_graph, trainer._rec = False, None
    with hip.record_launches() as rec:
        trainer.step(batches[0])
        torch.cuda.synchronize()
    trainer.use_graph, trainer._rec = saved
    return rec


def roofline_leg(rec, dtype_name, gemm_family=False, workload=None):
    """Device time of every distinct launch of ONE training step, measured live with HIP events on the launch
    stream: each distinct C-ABI call of the recorded eager step is re-issued 20x inside a hipGraph (so host launch
    overhead is not in the number) and replayed 3x between two events.  The dominant entry gets the roofline object:
    achieved = algorithmic FLOPs per launch / average launch duration (GEMMs), or algorithmic bytes / duration.
    gemm_family: the Linear GEMM entry points (forward, dgrad, weight gradient -- one kernel family, gemm.hip) count as
    ONE candidate, with every shape listed (the transformer step is 29 GEMM launches of 8 shapes)."""
    from inferbiomechanics_amd import hip
    uniq = {}
    for i, (name, args) in enumerate(rec.calls):
        if name in hip._RecordingLib.SKIP:
            continue
        note = rec.notes.get(i)
        key = (name, _ints(args), None if note is None else str(note[2]))
        if key not in uniq:
            uniq[key] = [args, 0, note]
        uniq[key][1] += 1
    rows = []
    for (name, ints, _), (args, count, note) in uniq.items():
        us = hip.time_recorded_call(name, args)
        rows.append({"entry": name, "dims": list(ints), "launches_per_step": count, "avg_launch_us": round(us, 2),
                     "us_per_step": round(us * count, 2), "_note": note})
    rows.sort(key=lambda r: -r["us_per_step"])
    total = sum(r["us_per_step"] for r in rows)
    es = 2 if dtype_name == "bf16" else 4

    def work(r):
        """(algorithmic flops, algorithmic bytes) of ONE launch; dims = the call's small-integer arguments, which
        for every entry point end with (..., M, N, K, dtype) or (..., M, N, dtype)"""
        d, e = r["dims"], r["entry"]
        if r["_note"] is not None:                      # grouped launches: shapes sit in arrays, the wrapper noted them
            return r["_note"][0], r["_note"][1]
        if e == "ib_mlp_chain_train":
            # dims end with (M, T, D, H, L).  FLOPs: the forward GEMMs (D->H, (L-1) x H->H, H->D) and the dgrad chain
            # (D->H transposed head, (L-1) x H->H).  Bytes: what the step needs in HBM given that the weight gradients
            # are separate GEMMs -- x0, eps in; x_t, dpred, h_i, dz_i out; the packed weights once per workgroup wave
            # from L2 are not HBM traffic (2 x the weight bytes counted once).  u_i (written, re-read) is overhead.
            M, T_, D_, H_, L_ = d[-5], d[-4], d[-3], d[-2], d[-1]
            fw = D_ * H_ + (L_ - 1) * H_ * H_ + H_ * D_
            bw = D_ * H_ + (L_ - 1) * H_ * H_
            return 2 * M * (fw + bw), 4 * M * D_ * es + 2 * L_ * M * H_ * es + (fw + bw) * es
        if e == "ib_linear_wgrad_slabs":
            M, N, K = d[-4], d[-3], d[-2]
            return 2 * M * N * K, (M * K + M * N) * es + N * K * 4
        if e in ("ib_linear_fwd", "ib_linear_dgrad", "ib_linear_dgrad_wt", "ib_linear_wgrad", "ib_linear_ln_fwd",
                 "ib_linear_wgrad_bias"):
            M, N, K = d[-4], d[-3], d[-2]
            return 2 * M * N * K, (M * K + N * K) * es + M * N * (4 if "wgrad" in e else es)
        if e in ("ib_attention_fwd", "ib_attention_bwd"):
            B_, T_, H_, dh = d[-5], d[-4], d[-3], d[-2]
            io = B_ * T_ * H_ * dh * es
            return (4 if e.endswith("fwd") else 10) * T_ * T_ * dh * B_ * H_, (4 if e.endswith("fwd") else 8) * io
        if e in ("ib_ffn_chain_fwd", "ib_ffn_chain_bwd"):
            # the token-local half of a layer: out-projection (d x d) + both feed-forward GEMMs (2 x d x ffn) per token row,
            # forward or dgrad.  Bytes: what must cross HBM -- attn / x (dy / s2 / s1) in, x1 / s1 / s2 / y (ds2 / ds1 / dattn)
            # and the [M, ffn] weight-gradient operand out, the packed weights once
            M, d_, ff = d[-3], d[-2], d[-1]
            io = (6 if e.endswith("fwd") else 6) * M * d_ * es + M * ff * es
            return 2 * M * (d_ * d_ + 2 * d_ * ff), io + (d_ * d_ + 2 * d_ * ff) * es
        if e in ("ib_layernorm_fwd", "ib_layernorm_bwd"):
            M, N = d[-3], d[-2]
            return 0, (2 if e.endswith("fwd") else 3) * M * N * es
        return 0, 0

    # the dominant KERNEL = the entry point (or, with gemm_family, the GEMM kernel family) with the largest share of the
    # step, all its shapes together
    fam = {}
    for r in rows:
        fl, by = work(r)
        r["tflops"] = round(fl / (r["avg_launch_us"] * 1e-6) / 1e12, 1) if fl and r["avg_launch_us"] > 0 else None
        name = "linear GEMMs" if (gemm_family and r["entry"] in GEMM_ENTRIES) else r["entry"]
        f = fam.setdefault(name, {"us": 0.0, "launches": 0, "flops": 0, "bytes": 0, "rows": []})
        f["us"] += r["us_per_step"]; f["launches"] += r["launches_per_step"]
        f["flops"] += fl * r["launches_per_step"]; f["bytes"] += by * r["launches_per_step"]
        f["rows"].append(r)
    top_e, top = max(fam.items(), key=lambda kv: kv[1]["us"])
    avg_us = top["us"] / top["launches"]
    if top_e == "linear GEMMs":
        kernel = ("Linear GEMM family: gemm_nt_kernel (forward, dgrad; gemm_nt.hip), gemm_tn_kernel (weight gradients; "
                  "gemm_tn.hip), gemm_ring_kernel / gemm_kernel (the remaining shapes; gemm.hip)")
        shapes = [{"entry": r["entry"], "MNK": (r["_note"][2] if r["_note"] else r["dims"][-4:-1]),
                   "launches": r["launches_per_step"], "us": r["avg_launch_us"], "tflops": r["tflops"]} for r in top["rows"]]
    else:
        kernel = KERNEL_OF.get(top_e, top_e)
        shapes = [(r["dims"][-5:] if top_e == "ib_mlp_chain_train" else r["dims"][-4:-1]) for r in top["rows"]]
        if top["rows"][0]["_note"] is not None:          # launch forms described by the wrapper (fused layer launches)
            shapes = [dict(r["_note"][2], launches=r["launches_per_step"], us=r["avg_launch_us"], tflops=r["tflops"])
                      if isinstance(r["_note"][2], dict) else r["_note"][2] for r in top["rows"]]
    out = {"kernel": kernel, "entry": top_e, "launches_per_step": top["launches"],
           "avg_launch_us": round(avg_us, 2), "share_of_step_device_time": round(top["us"] / total, 3),
           "shapes": shapes, "traffic": None}
    if top_e == "ib_mlp_chain_train":
        out["shape_fields"] = ["tokens M = B*T", "T", "D", "H", "blocks L"]
    if top["flops"]:
        ach = top["flops"] / (top["us"] * 1e-6) / 1e12
        out.update({"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype_name], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[dtype_name], 4),
                    "algorithmic_flops_per_launch": top["flops"] // top["launches"],
                    "algorithmic_bytes_per_launch": top["bytes"] // top["launches"]})
    elif top["bytes"]:
        ach = top["bytes"] / (top["us"] * 1e-6) / 1e9
        out.update({"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": top["bytes"] // top["launches"]})
    else:
        out.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None})
    # HBM traffic per launch and MFMA utilisation: STORED figures from the committed rocprofv3 --pmc passes (counters cannot
    # be collected inside this run), keyed by WORKLOAD and entry (tools/summarize_profile.py); FETCH_SIZE x2 on gfx950 +
    # WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES / elapsed SIMD cycles.  The line says that they are stored, which build they
    # were collected on, and whether the kernel sources have changed since (`traffic_stale`).  null if no profile of this
    # workload's dominant kernel has been summarised.
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and workload is not None:
        try:
            from tools.csrc_hash import csrc_hash
            tj = json.load(open(tpath))
            t = tj.get(workload, {}).get(top_e)
            meta = tj.get("_meta", {}).get(workload, {})
            if t:                                    # `traffic`: HBM bytes per launch (a number, like `achieved`)
                out["traffic"] = t["hbm_bytes_per_launch"]
                if "mfma_util" in t:
                    out["mfma_util"] = t["mfma_util"]
                here = csrc_hash()
                out["traffic_measured"] = "stored"
                out["traffic_stale"] = meta.get("csrc_hash") != here
                out["traffic_source"] = dict({k: v for k, v in t.items() if k not in ("hbm_bytes_per_launch", "mfma_util")},
                                             profiled_csrc_hash=meta.get("csrc_hash"), this_csrc_hash=here,
                                             profiled_git=meta.get("git_head_when_summarised"), tag=meta.get("tag"))
        except Exception:
            pass
    for r in rows:
        r.pop("_note", None)
    return out, rows[:16], total


def ddim_leg(dev, dtype, B=16, T=200, D=300, steps=100):
    """BASELINE config 5: T = 200 transformer denoiser, 100-step DDIM, one captured denoise step replayed."""
    from inferbiomechanics_amd.diffusion.sampler import DDIMSampler
    model = build_model("transformer", T, D, dtype, dev)
    sampler = DDIMSampler(model, steps, use_graph=True)
    xT = torch.randn(B, T, D, device=dev)
    # one untimed loop of the same length (its buffers, graph and per-loop tables exist afterwards: a first 100-step call
    # after a 3-step warm-up was sometimes 2x slower -- allocations inside the loop), then three timed loops: the median
    sampler.sample(xT)
    torch.cuda.synchronize()
    loops = []
    for _ in range(3):
        t0 = time.perf_counter()
        sampler.sample(xT)
        torch.cuda.synchronize()
        loops.append(time.perf_counter() - t0)
    el = sorted(loops)[1]
    # forward FLOPs of one denoise step (the frame-embedding half of the input projection and
    # the time-MLP are computed once per loop, not per step)
    d, ffn, L = 512, 2048, 4
    flops = 2 * (L * (4 * d * d + 2 * d * ffn + 2 * T * d) + D * d + d * D) * T * B
    ach = flops * steps / el / 1e12
    dn = "bf16" if dtype == torch.bfloat16 else "f32"
    return {"workload": f"transformer_denoiser_T{T} B={B} {steps}-step DDIM (hipGraph-replayed step)",
            "steps_per_sec": round(steps / el, 1), "window_steps_per_sec": round(B * steps / el, 1),
            "ms_per_sample_batch": round(el * 1e3, 2), "timed_loops_ms": [round(v * 1e3, 2) for v in loops],
            "roofline": {"bound": "mfma" if B >= 16 else "launch latency (one window: every GEMM is a single wave of "
                         "workgroups; the figure is reported against the MFMA peak all the same)",
                         "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dn], "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_TFLOPS[dn], 4), "algorithmic_flops_per_step": flops,
                         "scope": "whole denoise step (forward plan + DDIM update), wall clock over the loop"}}


GEMM_ENTRIES = ("ib_linear_fwd", "ib_linear_dgrad", "ib_linear_dgrad_wt", "ib_linear_wgrad", "ib_linear_wgrad_slabs",
                "ib_linear_wgrad_slabs_multi", "ib_linear_wgrad_slabs_multi_bias", "ib_linear_wgrad_bias", "ib_linear_ln_fwd")


def algorithmic_bytes_per_step(kind, T, D, B, nparams):
    """compulsory HBM traffic of one training step, SURVEY.md §8d 'Algorithmic bytes': per window x0 + eps read and the
    saved activations written once / read once; per step the weights read twice (bf16), gradients written, optimizer
    state read / written (RMSprop: w, g, v = 5 fp32 passes) ~ 7 x 4 bytes per parameter"""
    if kind == "mlp":
        per_window = 2 * T * D * 2 + 2 * T * 512 * 2 * 2
    else:
        per_window = 2 * T * D * 2 + 2 * 4 * T * (512 * 4 + 1536 + 2048) * 2
    return B * per_window + 7 * 4 * nparams


def train_leg(workload, a, dev, world, rank, steps, warmup, sync, with_roofline=True):
    """one training workload: W warm-up steps, EXACTLY K timed steps between barriers + synchronisations, max over
    ranks; then the single-step distribution and (rank 0) the per-launch roofline table."""
    from inferbiomechanics_amd.engine import HipTrainer
    kind, T, D, B = WORKLOADS[workload]
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = build_model(kind, T, D, dtype, dev)
    trainer = HipTrainer(model, "diffusion", a.opt_type, 1e-4, use_graph=not a.no_graph, bucket_mb=a.bucket_mb,
                         overlap_comm={"auto": None, "on": True, "off": False}[a.overlap_comm])
    # >= 64 pre-generated batches.  64 x 15.4 MB (x0 + eps, bf16) = 983 MB, four times the 256-MiB Infinity
    # Cache, so the timed loop reads its inputs from HBM (16 batches = 246 MB could have been cache hits)
    batches = make_batches(a.batches, B, T, D, dtype, dev, seed=rank)
    # the whole leg runs ON the trainer's stream, as cli/train.py's loop does: a step() called from another stream hands
    # over through two cross-queue events per step (25 us of the 0.218 ms MLP step)
    prev_stream = trainer.adopt_stream()
    # priming (not part of W): eager warm-up, the generic graph, and the graph of EVERY batch of the ring -- no capture
    # may fall into the timed region whatever --steps / --warmup are (round 1: four ~1 ms captures inside 20 timed steps)
    trainer.pin_batches(batches)
    for i in range(len(batches)):
        trainer.step(batches[i])
    for i in range(warmup):
        trainer.step(batches[i % len(batches)])
    cap0 = trainer.captures
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.step(batches[i % len(batches)])
    sync()
    el = time.perf_counter() - t0
    if trainer.captures != cap0:
        raise SystemExit(f"bench: {trainer.captures - cap0} graph capture(s) inside the timed region")
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.cpu())
    loss = trainer.loss_value()

    # distribution of single-step times (median, p10 / p90): one event per step on the caller's stream,
    # read back after the run -- no host synchronisation inside it
    rehearsal = os.environ.get("IB_BENCH_REHEARSAL") == "1"      # one GPU, gloo: every all-reduce goes through the host
    nq = 6 if rehearsal else max(20, min(steps, 400))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nq + 1)]
    evs[0].record()
    for i in range(nq):
        trainer.step(batches[i % len(batches)])
        evs[i + 1].record()
    sync()
    dts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nq))
    pct = lambda q: round(dts[min(nq - 1, int(q * nq))], 4)

    xg = None
    if world > 1 and trainer.ddp:
        # the gradient all-reduce alone (same buffer, same communicator): bus bandwidth against the xGMI peak
        g = trainer.grad
        for _ in range(3):
            dist.all_reduce(g)
        sync()
        t1 = time.perf_counter()
        nar = 3 if rehearsal else 20
        for _ in range(nar):
            dist.all_reduce(g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nar
        busbw = 2 * (world - 1) / world * g.numel() * 4 / dt / 1e9
        xg = {"allreduce_bytes": g.numel() * 4, "avg_us": round(dt * 1e6, 1), "busbw_GBs": round(busbw, 1),
              "peak_GBs": PEAK_XGMI_GBS, "frac": round(busbw / PEAK_XGMI_GBS, 4),
              "note": "peak = 7 xGMI links x 153 GB/s per GPU; busbw = 2(N-1)/N x bytes / time"}
        g.zero_()
    rec = record_eager_step(trainer, batches) if with_roofline else None     # all ranks: the step holds the all-reduce
    if prev_stream is not None:
        torch.cuda.synchronize()
        torch.cuda.set_stream(prev_stream)
    out = None
    if rank == 0:
        value = world * B * steps / el
        nparams = sum(p.numel() for p in model.parameters())
        tfl = value * train_flops_per_window(kind, T, D) / 1e12
        hbm = algorithmic_bytes_per_step(kind, T, D, B, nparams) * world / (el / steps) / 1e9
        out = {"workload": f"{workload}_D{D}_B{B}_{a.dtype}", "value": round(value, 1), "unit": "windows/s",
               "ms_per_step": round(el / steps * 1e3, 4), "steps": steps, "warmup": warmup,
               "step_ms": {"p10": pct(0.10), "median": pct(0.50), "p90": pct(0.90), "samples": nq},
               "final_loss": round(loss, 6), "train_tflops": round(tfl, 2),
               "captures_in_timed_region": trainer.captures - cap0,
               "config": {"per_gpu_batch": B, "global_batch": B * world, "window": T, "feat": D, "optimizer": a.opt_type,
                          "hipgraph": not a.no_graph, "parallelism": f"dp{world}", "pregenerated_batches": len(batches),
                          "grad_buckets": len(trainer.buckets.ranges) if trainer.ddp else 0,
                          "bucket_bytes": [4 * (hi - lo) for lo, hi in trainer.buckets.ranges] if trainer.ddp else [],
                          "overlap_comm": bool(trainer.overlap_comm),
                          "loop_stream": "trainer (HipTrainer.adopt_stream, as cli/train.py)" if prev_stream is not None
                          else "caller"},
               # whole-step fractions of the three rooflines SURVEY.md §8d names (per GPU)
               "step_fractions": {"mfma": round(tfl / world / PEAK_TFLOPS[a.dtype], 4),
                                  "hbm": round(hbm / world / PEAK_HBM_GBS, 4),
                                  "xgmi": None if xg is None else xg["frac"]}}
        if xg is not None:
            out["xgmi"] = xg
        if rec is not None:
            rl, breakdown, dev_us = roofline_leg(rec, a.dtype, gemm_family=False, workload=workload)
            out["roofline"] = rl
            out["step_sum_of_kernel_us"] = round(dev_us, 1)
            out["step_breakdown"] = breakdown
    del trainer, model, batches
    torch.cuda.empty_cache()
    return out


def self_launch(n):
    """Run this very command line as n ranks of one node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` as a child process (the reference
    expects an env:// launcher too: src/cli/train.py:99-102, src/.gitignore:10-11).  Returns the child's return code
    after printing rank 0's JSON line (the last stdout line that parses as the result object)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        sys.stderr.write("bench: the launched ranks printed no result line\n")
        return 1
    return p.returncode


def graph_collectives_child(n, workload, form, steps=60, warmup=10, timeout_s=240):
    """Data-parallel runs only: the SAME main workload once more in the OTHER form of the collectives than the headline's
    (`form` = "1": all-reduces captured inside the step's hipGraph -- one graph per step, no graph cut and no host action per
    collective; "0": host actions between graph segments) -- in FRESH child processes (a new torch.distributed.run launch
    of n ranks; nothing that has touched a GPU is ever re-exec'd), bounded by a timeout, and fenced: whatever happens there
    (non-zero exit, timeout, no JSON) becomes an `error` string in the extra key and never costs the record of the headline
    run, whose own form was chosen by the start-up probe (inferbiomechanics_amd/ddp_probe.py)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
              "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)
    env.update(IB_GRAPH_COLLECTIVES=form, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    if n == 1:
        env["IB_DDP_SELFTEST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n), "--steps", str(steps), "--warmup",
           str(warmup), "--workload", workload, "--no-mlp", "--no-ddim", "--no-cpu-baseline", "--no-variant-child", "--no-roofline"]
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(p.pid, signal.SIGKILL)          # the launcher's own process group: exactly the children started here
            p.communicate()
            return {"error": f"timed out after {timeout_s} s", "seconds": round(time.perf_counter() - t0, 1)}
        for ln in out.decode(errors="replace").splitlines():
            ln = ln.strip()
            if ln.startswith("{") and '"metric"' in ln:
                try:
                    d = json.loads(ln)
                    return {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": steps,
                            "warmup": warmup, "final_loss": d.get("final_loss"), "rccl_world": d.get("rccl_world"),
                            "seconds": round(time.perf_counter() - t0, 1)}
                except (ValueError, KeyError):
                    pass
        return {"error": f"child exited with code {p.returncode} and printed no result line",
                "seconds": round(time.perf_counter() - t0, 1)}
    except Exception as exc:                          # nothing in here may cost the record
        return {"error": repr(exc)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="transformer_denoiser_T50", choices=list(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--opt-type", default="rmsprop")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batches", type=int, default=64, help="pre-generated synthetic batches resident in HBM (SURVEY §8d: >= 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ddim", action="store_true")
    ap.add_argument("--no-cli-path", action="store_true", help="skip the `main.py train` rate of the same workload")
    ap.add_argument("--variant-child", action="store_true",
                    help="[N > 1] also run the collectives' OTHER form (captured in the graph / host actions) in child processes "
                         "(doubles the processes on the node for its duration: opt-in)")
    ap.add_argument("--no-variant-child", action="store_true", help="(accepted for older command lines; the variant is opt-in)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch roofline leg (child runs)")
    ap.add_argument("--no-mlp", action="store_true", help="skip the configs[1] MLP denoiser leg (extra key `mlp_T50`)")
    ap.add_argument("--bucket-mb", type=float, default=13.0,
                    help="gradient bucket size when all-reduces overlap the backward: each bucket boundary cuts the captured "
                         "graph (about 15 us); 13 MiB = one transformer layer of the T=50 denoiser")
    ap.add_argument("--overlap-comm", default="auto", choices=["auto", "on", "off"],
                    help="data-parallel policy: all-reduce buckets during the backward (on) or once after it (off)")
    a = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL prints its version there
    # at communicator creation) are sent to stderr for the whole run; the result is written to the saved descriptor
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)

    # the host driver of this pool only supports dmabuf IPC; without this RCCL's cross-process buffer registration fails
    # (hipIpcGetMemHandle: invalid argument).  Must be in the environment before the first HIP call.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # what capturing the all-reduces inside the step graph needs from the process group (read when it is created): c10d's
    # completion-event cache off, its flight recorder on (inferbiomechanics_amd/ddp_probe.py, engine._drain_c10d_watchdog)
    from inferbiomechanics_amd import ddp_probe as _probe_env
    _probe_env.prepare_env()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    forced_form = os.environ.get("IB_GRAPH_COLLECTIVES")      # set by a caller (the variant child below; tools): no probe
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has made no GPU / HIP
        # call yet (and never will): the ranks are FRESH child processes under torch.distributed.run (subprocess, never
        # os.exec*); rank 0's single JSON line and the launcher's return code are relayed.
        os.dup2(out_fd, 1)
        raise SystemExit(self_launch(a.gpus))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # IB_BENCH_REHEARSAL=1: the multi-rank control flow on a ONE-GPU box -- every rank on device 0, gloo instead of
    # RCCL (which refuses two ranks on one device).  Only for checking that no rank waits for a collective another rank
    # never enters; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("IB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    selftest = os.environ.get("IB_DDP_SELFTEST") == "1"      # 1-rank run of the whole RCCL / bucket / segment path
    backend = None
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()
    # a host-side group for waits that must not occupy the GPUs (the variant child below runs its own RCCL kernels)
    host_group = dist.new_group(backend="gloo") if (world > 1 and not rehearsal) else None

    from inferbiomechanics_amd import hip
    hip.lib()

    def sync():
        torch.cuda.synchronize()
        if world > 1 or selftest:
            dist.barrier()
            torch.cuda.synchronize()

    kind, T, D, B = WORKLOADS[a.workload]
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    main_leg = train_leg(a.workload, a, dev, world, rank, a.steps, a.warmup, sync, with_roofline=not a.no_roofline)
    # BASELINE configs[1]: the token-wise MLP denoiser at T = 50, its own step count (0.13 ms steps); every rank takes part
    # (under data parallelism the step holds the all-reduces)
    mlp_leg = None
    if kind != "mlp" and not a.no_mlp:
        mlp_leg = train_leg("mlp_denoiser_T50", a, dev, world, rank, 8 if rehearsal else 2000, 2 if rehearsal else 200, sync,
                            with_roofline=not a.no_roofline)
    variant = None
    # (IB_BENCH_FORCE_VARIANT=1 with IB_DDP_SELFTEST=1: the same mechanism on a one-GPU box, a 1-rank child)
    want_variant = (world > 1 and a.variant_child) or (selftest and os.environ.get("IB_BENCH_FORCE_VARIANT") == "1")
    from inferbiomechanics_amd import ddp_probe
    coll = ddp_probe.verdict() if (world > 1 or selftest) else None
    if want_variant and not rehearsal and not a.no_variant_child and forced_form is None and coll is not None:
        sync()
        if rank == 0:
            if coll["source"] == "probe" and not coll["captured"]:
                variant = {"skipped": "the start-up probe of the captured form failed (" + str(coll["why"])[:200] + ")"}
            else:
                variant = graph_collectives_child(world, a.workload, "0" if coll["captured"] else "1")
                variant["form"] = "host actions between graph segments" if coll["captured"] else "captured in the step graph"
        if host_group is not None:
            dist.barrier(group=host_group)             # the other ranks wait on the HOST (gloo), their GPUs idle
    if rank == 0:
        cfg = dict(main_leg["config"])
        which = {"mlp": "configs[1]", "transformer": "configs[2]" if world == 1 else
                 "configs[3]: configs[2] data-parallel, 256 windows per GPU"}[kind] if T == 50 else "configs[4] model, training"
        cfg["workload"] = main_leg["workload"] + f" (BASELINE.json {which})"
        line = {
            "metric": "motion-windows/sec training (+ DDIM steps/sec)", "value": main_leg["value"], "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_leg["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": cfg,
            # proof that the collective library saw N ranks (a SCALE record must show rccl_world == n_gpus)
            "rccl_world": dist.get_world_size() if backend else 1, "backend": backend,
        }
        if coll is not None:
            # how the step's all-reduces are issued, and who decided (ddp_probe.py: a probe in fresh child processes)
            line["collectives"] = dict(coll, form="captured in the step graph" if coll["captured"] else
                                       "host actions between graph segments")
        for k in ("step_ms", "final_loss", "train_tflops", "captures_in_timed_region", "step_fractions", "xgmi", "roofline",
                  "step_sum_of_kernel_us"):
            if k in main_leg:
                line[k] = main_leg[k]
        summary = {"headline": {"workload": cfg["workload"], "windows_per_s": main_leg["value"],
                                "ms_per_step": main_leg["ms_per_step"], "n_gpus": world,
                                "mfma_frac_step": main_leg["step_fractions"]["mfma"]}}
        if "roofline" in main_leg:
            rl = main_leg["roofline"]
            summary["roofline"] = {k: rl.get(k) for k in ("entry", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                          "traffic", "launches_per_step")}
        if coll is not None:
            summary["collectives"] = {"captured_in_step_graph": coll["captured"], "decided_by": coll["source"]}
        if variant is not None:
            variant["what"] = ("same workload with the collectives in the OTHER form than the headline run, in fresh child "
                               "processes")
            line["graph_collectives_variant"] = variant
            summary["other_collectives_form_ms"] = variant.get("ms_per_step", variant.get("error", variant.get("skipped")))
        if not a.no_cpu_baseline and world == 1:            # rank 0 at N=1 only (the other ranks wait at the barrier)
            line["cpu_baseline"] = cpu_baseline(kind, T, D, B, dev)
            ml = line["cpu_baseline"]["matched_loss"]
            summary["cpu_baseline"] = {"windows_per_s": line["cpu_baseline"]["value"], "cores": line["cpu_baseline"]["cores"],
                                       "speedup": round(main_leg["value"] / max(line["cpu_baseline"]["value"], 1e-9), 1),
                                       "matched_loss_rel_diff_f32": ml["rel_diff_f32"],
                                       "matched_loss_rel_diff_bf16": ml["rel_diff_bf16"]}
        if mlp_leg is not None:
            mlp_leg["workload"] += " (BASELINE.json configs[1])"
            summary["mlp_T50"] = {"windows_per_s": mlp_leg["value"], "ms_per_step": mlp_leg["ms_per_step"],
                                  "mfma_frac_step": mlp_leg["step_fractions"]["mfma"]}
            if "roofline" in mlp_leg:
                summary["mlp_T50"]["chain_kernel_frac"] = mlp_leg["roofline"].get("frac")
            if not a.no_cpu_baseline and world == 1:
                mlp_leg["cpu_baseline"] = cpu_baseline("mlp", 50, 300, 256, dev)
                summary["mlp_T50"]["cpu_windows_per_s"] = mlp_leg["cpu_baseline"]["value"]
                summary["mlp_T50"]["matched_loss_rel_diff_bf16"] = mlp_leg["cpu_baseline"]["matched_loss"]["rel_diff_bf16"]
                if not a.no_cli_path:
                    cp = cli_path_leg()
                    cp["fraction_of_bench_value"] = round(cp["windows_per_s"] / mlp_leg["value"], 3)
                    mlp_leg["cli_path"] = cp
                    summary["mlp_T50"]["cli_path_windows_per_s"] = cp["windows_per_s"]
            mlp_leg.pop("step_breakdown", None)             # the per-launch tables live in profiles/ (keeps the line short)
            line["mlp_T50"] = mlp_leg
        if not a.no_ddim:
            legs = [ddim_leg(dev, dtype, B=b) for b in (1, 16, 256)]          # B in {1, 16, 256}
            line["ddim"] = legs[1]                                            # B = 16: the quoted figure
            line["ddim_batches"] = [legs[0], legs[2]]
            summary["ddim_T200_steps_per_s"] = {f"B{b}": lg["steps_per_sec"] for b, lg in zip((1, 16, 256), legs)}
            summary["ddim_T200_mfma_frac"] = {f"B{b}": lg["roofline"]["frac"] for b, lg in zip((1, 16, 256), legs)}
        if not a.no_cpu_baseline and world == 1:
            # BASELINE.json configs[0]: the reference's own CPU-runnable case (plumbing; launch-latency bound on the GPU)
            line["regression_ref_shape"] = {
                "workload": "FeedForwardBaseline([512,512], sigmoid) 1470->300, fp32, RMSprop 1e-4, reference loss "
                            "(BASELINE.json configs[0]; SURVEY.md 8d Config 1; src/cli/train.py:240-284)",
                "bound": "launch latency (11 dependent launches of a few us each; 7 MFLOP per window)",
                "legs": [regression_ref_shape_leg(dev, b) for b in (4, 64)]}
            summary["regression_ref_shape"] = {f"B{lg['batch']}": {"gpu_ms": lg["gpu_ms_per_step"], "cpu_ms": lg["cpu_ms_per_step"],
                                                                    "loss_rel_diff": lg["loss_at_step_20"]["rel_diff"]}
                                               for lg in line["regression_ref_shape"]["legs"]}
        if "step_breakdown" in main_leg:
            line["step_breakdown"] = main_leg["step_breakdown"][:10]
        line["summary"] = summary                           # LAST: an 8-KB tail of the line still holds every headline figure
        sys.stdout.flush()
        os.write(out_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or selftest:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
