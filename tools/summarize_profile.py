#!/usr/bin/env python3
"""Turn a tools/profile_round.sh collection (gpurun_out/prof_<tag>/) into the committed evidence under profiles/:

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_per_kernel.md        per (kernel, grid) average duration joined with the HBM counters
  profiles/traffic.json               {workload: {C-ABI entry: {"hbm_bytes_per_launch": ..., ...}}, "_meta": {workload: {...}}}
                                      read by bench.py's roofline leg.  Keyed by WORKLOAD: the same entry point (and the
                                      "linear GEMMs" family) runs different kernels / shapes in the MLP and the transformer
                                      step.  _meta carries the hash of the kernel sources the counters were collected on
                                      (tools/csrc_hash.py, written on the GPU box by tools/profile_round.sh) and the git commit.

usage: tools/summarize_profile.py <tag> <workload>     e.g. r03_mlp mlp_denoiser_T50 / r03_tr transformer_denoiser_T50

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE / WRITE_SIZE are in KiB and come from SEPARATE
--pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md §HBM).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRY_OF = [# round 5: the one-window-panel launches with the attention inside (and the top layer's launch of the same entry point)
            ("ffn_chain_bwd_kernel<true, false, true", "", "ib_ffn_chain_bwd_attn"),
            ("ffn_chain_fwd_kernel<true, true, true, false", "", "ib_ffn_chain_fwd_attn"),
            ("ffn_chain_fwd_kernel<true, false, false, false", "", "ib_ffn_chain_fwd_attn"),
            ("ffn_chain_fwd_kernel<true, true, false, true", "", "ib_ffn_chain_fwd_infer"),
            ("ffn_chain_fwd_kernel<true, false, false, true", "", "ib_ffn_chain_fwd_infer"),
            ("ffn_chain_fwd_kernel", "", "ib_ffn_chain_fwd"), ("ffn_chain_bwd_kernel", "", "ib_ffn_chain_bwd"),
            ("ffn_pack_kernel", "", "ib_ffn_chain_pack"), ("diffusion_draw_kernel", "", "ib_diffusion_draw"),
            ("gemm_tn256w4_kernel", "", "ib_linear_wgrad_slabs_multi"), ("gemm_tn256_kernel", "", "ib_linear_wgrad_slabs_multi"), ("gemm_tn_kernel", "", "ib_linear_wgrad_slabs_multi"), ("gemm_nt_kernel<0, 0, false", "", "ib_linear_fwd"),
            ("gemm_nt_kernel<1, 0", "", "ib_linear_fwd"), ("gemm_nt_kernel<0, 1", "", "ib_linear_dgrad"),
            ("gemm_nt_kernel<0, 0, true", "", "ib_linear_dgrad"), ("gemm_nt_kernel", "", "ib_linear_fwd"),
            ("mlp_chain_kernel", "", "ib_mlp_chain_train"), ("mlp_chain2_kernel", "", "ib_mlp_chain_train"),
            ("time_mlp_bwd_kernel", "", "ib_time_mlp_bwd"), ("layernorm_bwd512_kernel", "", "ib_layernorm_bwd"),
            ("layernorm_fwd512_kernel", "", "ib_layernorm_fwd"), ("gemm_ring_wgrad_multi_kernel", "", "ib_linear_wgrad_slabs_multi"),
            ("optim_kernel<true>", "", "ib_optim_step_sources"), ("optim_kernelILb1", "", "ib_optim_step_sources"),
            ("step_reduce_kernel", "", "ib_step_reduce"), ("time_mlp_fwd_kernel", "", "ib_mlp_chain_prep"),
            ("gemm_ring_kernel", "Lb0ELb0ELi2", "ib_linear_wgrad_slabs"), ("gemm_ring_kernel", "Lb1ELb1ELi0", "ib_linear_fwd"),
            ("gemm_ring_kernel", "Lb1ELb0ELi1", "ib_linear_dgrad"), ("slab_reduce_multi_kernel", "", "ib_slab_reduce_multi"),
            ("colsum_segments_kernel", "", "ib_colsum_segments"),
            ("gemm_kernel", "Lb1ELb1ELi0", "ib_linear_fwd"), ("gemm_kernel", "Lb1ELb0ELi1", "ib_linear_dgrad"),
            ("gemm_kernel", "Lb0ELb0ELi2", "ib_linear_wgrad"), ("layernorm_fwd_kernel", "", "ib_layernorm_fwd"),
            ("layernorm_bwd_kernel", "", "ib_layernorm_bwd"), ("attn_fwd_mfma", "", "ib_attention_fwd"),
            ("attn_bwd_mfma", "", "ib_attention_bwd")]


def entry_of(name):
    for a, b, e in ENTRY_OF:
        if a in name and b in name:
            return e
    # demangled form of the template kernels
    if "gemm_ring_kernel" in name:
        if "false, false" in name:
            return "ib_linear_wgrad_slabs"
        return "ib_linear_fwd" if "true, true" in name else "ib_linear_dgrad"
    if "gemm_kernel" in name:
        if "true, true" in name or ", true, 0>" in name:
            return "ib_linear_fwd"
        if "false, 1>" in name:
            return "ib_linear_dgrad"
    return None


def first(pattern):
    """the NEWEST match: gpurun merges every call's files into gpurun_out/, so earlier collections of the same tag linger"""
    g = glob.glob(pattern)
    return max(g, key=os.path.getmtime) if g else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    workload = sys.argv[2] if len(sys.argv) > 2 else ("transformer_denoiser_T50" if "_tr" in tag else "mlp_denoiser_T50")
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = first(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    trace = list(csv.DictReader(open(first(os.path.join(src, "trace", "*", "*kernel_trace.csv")))))

    def counters(sub):
        f = first(os.path.join(src, sub, "*", "*counter_collection.csv"))
        a = collections.defaultdict(list)
        if f:
            for r in csv.DictReader(open(f)):
                a[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        return a
    F, W = counters("pmc_fetch"), counters("pmc_write")

    def counters_by_name(sub):
        f = first(os.path.join(src, sub, "*", "*counter_collection.csv"))
        a = collections.defaultdict(lambda: collections.defaultdict(list))
        if f:
            for r in csv.DictReader(open(f)):
                a[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return a
    Q = counters_by_name("pmc_mfma")

    def mfma(k):
        """MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (elapsed cycles x 1024),
        elapsed cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs) -- the gfx94x `MfmaUtil`
        expression of rocprofv3's own counter file; executed MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512"""
        q = Q.get(k)
        if not q or not q.get("SQ_VALU_MFMA_BUSY_CYCLES") or not q.get("GRBM_GUI_ACTIVE"):
            return None, None
        avg = lambda n: sum(q[n]) / len(q[n]) if q.get(n) else 0.0
        busy, gui = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE")
        return (busy / (gui / 8.0 * 1024.0) if gui else None), avg("SQ_INSTS_VALU_MFMA_MOPS_BF16") * 512.0
    T = collections.defaultdict(list)
    n = len(trace)
    for r in trace[n // 4:]:                         # skip process start-up / warm-up dispatches
        gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        T[(r["Kernel_Name"], str(gs))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, v in T.items():
        f, w = F.get(k), W.get(k)
        fk = sum(f) / len(f) if f else None
        wk = sum(w) / len(w) if w else None
        mu, mfl = mfma(k)
        rows.append({"kernel": k[0], "grid_threads": int(k[1]), "dispatches": len(v), "avg_us": sum(v) / len(v) / 1e3,
                     "total_ms": sum(v) / 1e6, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                     "hbm_bytes": None if fk is None or wk is None else (2 * fk + wk) * 1024,
                     "mfma_util": mu, "mfma_flops": mfl})
    rows.sort(key=lambda r: -r["total_ms"])
    with open(os.path.join(dst, f"{tag}_per_kernel.md"), "w") as f:
        f.write(f"# {tag}: per-kernel durations (rocprofv3 --kernel-trace, un-graphed bench.py) joined with HBM counters\n\n"
                "`hbm MB/launch` = (2 x FETCH_SIZE + WRITE_SIZE) KiB from separate `--pmc` passes (gfx950 FETCH_SIZE correction).\n"
                "Durations of sub-10 us kernels are inflated in the un-graphed run (the GPU idles between host launches);\n"
                "tools/kbench.py / bench.py's roofline leg time launches back-to-back inside a hipGraph.\n\n"
                "`MFMA util` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) from a third `--pmc` pass (busy matrix-pipe\n"
                "cycles over elapsed SIMD cycles: 100 % = the dense bf16 peak); `MFMA GFLOP` = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = the\n"
                "FLOPs the matrix pipes EXECUTED per launch (padding rows / columns included).\n\n"
                "| kernel | grid (threads) | dispatches | avg us | total ms | FETCH KiB | WRITE KiB | hbm MB/launch | MFMA util | MFMA GFLOP |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            name = r["kernel"].replace("void ", "").replace("(anonymous namespace)::", "")
            name = name.split("(")[0][-70:] or r["kernel"][:70]
            f.write(f"| `{name}` | {r['grid_threads']} | {r['dispatches']} | {r['avg_us']:.1f} | {r['total_ms']:.2f} | "
                    f"{'' if r['FETCH_SIZE_KiB'] is None else round(r['FETCH_SIZE_KiB'])} | "
                    f"{'' if r['WRITE_SIZE_KiB'] is None else round(r['WRITE_SIZE_KiB'])} | "
                    f"{'' if r['hbm_bytes'] is None else round(r['hbm_bytes'] / 1e6, 1)} | "
                    f"{'' if r['mfma_util'] is None else format(100 * r['mfma_util'], '.1f') + ' %'} | "
                    f"{'' if not r['mfma_flops'] else round(r['mfma_flops'] / 1e9, 2)} |\n")
    tpath = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if any(isinstance(v, dict) and "hbm_bytes_per_launch" in v for v in traffic.values()):
        traffic = {}                                 # round-2 layout (keyed by entry only): discard, it mixed workloads
    per = {}
    fam = collections.defaultdict(lambda: {"bytes": 0.0, "n": 0, "us": 0.0, "busy": 0.0, "flops": 0.0})
    GEMM = ("ib_linear_fwd", "ib_linear_dgrad", "ib_linear_wgrad", "ib_linear_wgrad_slabs", "ib_linear_wgrad_slabs_multi")
    for r in rows:
        e = entry_of(r["kernel"])
        if e and r["hbm_bytes"] is not None:
            for name in ([e, "linear GEMMs"] if e in GEMM else [e]):
                f_ = fam[name]
                f_["bytes"] += r["hbm_bytes"] * r["dispatches"]
                f_["n"] += r["dispatches"]
                f_["us"] += r["avg_us"] * r["dispatches"]
                if r["mfma_util"] is not None:       # time-weighted utilisation of the family
                    f_["busy"] += r["mfma_util"] * r["avg_us"] * r["dispatches"]
                    f_["flops"] += (r["mfma_flops"] or 0.0) * r["dispatches"]
    for e, v in fam.items():
        per[e] = {"hbm_bytes_per_launch": round(v["bytes"] / v["n"]), "avg_launch_us_in_profile": round(v["us"] / v["n"], 2),
                      "dispatches": v["n"], "mfma_util": round(v["busy"] / v["us"], 4) if v["busy"] else None,
                      "mfma_flops_executed_per_launch": round(v["flops"] / v["n"]) if v["flops"] else None,
                      "source": f"profiles/{tag}_per_kernel.md (rocprofv3 --pmc: FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in "
                      "separate passes; bytes = (2*FETCH + WRITE) KiB; mfma_util = busy / (GRBM_GUI_ACTIVE / 8 x 1024))"}
    traffic[workload] = per
    hfile = os.path.join(src, "csrc_hash.txt")
    ghash = os.popen(f"git -C {ROOT} rev-parse --short HEAD 2>/dev/null").read().strip()
    traffic.setdefault("_meta", {})[workload] = {
        "tag": tag, "csrc_hash": open(hfile).read().strip() if os.path.exists(hfile) else None,
        "git_head_when_summarised": ghash,
        "collected": "rocprofv3 --pmc passes of tools/profile_round.sh on an MI355X (stored, not measured by bench.py)"}
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(open(os.path.join(dst, f"{tag}_per_kernel.md")).read()[:3000])
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
