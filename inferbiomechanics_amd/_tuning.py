"""Every dispatch threshold and kill-switch of the host side in ONE object.

The shipping path reads no environment for tuning: `tuning.<name>` is the default below unless a test / tool has set the
attribute on the object (`monkeypatch.setattr(tuning, "no_nt", True)`).  Environment overrides (the `IB_*` variables of the
A/B tooling under tools/) are honoured ONLY while the measurement build of the library (`-DIB_AB`, selected through
IB_HIP_LIB) is the one loaded -- the rule the C side follows since round 4: a stray variable in a production environment
cannot move a benchmarked shape to another kernel.

Booleans: any non-empty value of the variable means True.  Integers: `int(value)`.  Strings as they are.
Launcher / rank variables (WORLD_SIZE, RANK, IB_HIP_LIB, IB_DDP_SELFTEST, IB_BENCH_REHEARSAL, IB_GRAPH_COLLECTIVES) are not
tuning and stay where they are read.
"""
import os

# attribute: (environment variable honoured by measurement builds, default)
_SPEC = {
    # ---- streams / branches
    "no_branch": ("IB_NO_BRANCH", ""),                 # comma-separated branch names to run inline
    "no_layer_branch": ("IB_NO_LAYER_BRANCH", False),
    "layer_branch": ("IB_LAYER_BRANCH", False),
    "no_outproj_branch": ("IB_NO_OUTPROJ_BRANCH", False),
    "pos_own_branch": ("IB_POS_OWN_BRANCH", False),   # frame-embedding gradients on a fifth branch of the step's tail (rounds 2-4)
    # ---- GEMM families
    "no_nt": ("IB_NO_NT", False),
    "no_wgrad_bias": ("IB_NO_WGRAD_BIAS", False),
    "no_layer_group": ("IB_NO_LAYER_GROUP", False),
    "no_tail_split": ("IB_NO_TAIL_SPLIT", False),
    "no_skinny": ("IB_NO_SKINNY", False),
    # ---- transformer layer launches
    "no_ffn_chain": ("IB_NO_FFN_CHAIN", False),
    "no_qkv_fuse": ("IB_NO_QKV_FUSE", False),
    "no_attn_fuse": ("IB_NO_ATTN_FUSE", False),        # round 5: attention inside the panel launches
    "no_lag_group": ("IB_NO_LAG_GROUP", False),
    # ---- sampler (frozen-weight forward)
    "no_qkv_panel": ("IB_NO_QKV_PANEL", False),
    "qkv_panel_max_m": ("IB_QKV_PANEL_MAX_M", 1600),
    "no_linear_ln": ("IB_NO_LINEAR_LN", False),
    "linln_k512_max_m": ("IB_LINLN_K512_MAX_M", 4095),
    "linln_max_m": ("IB_LINLN_MAX_M", 4096),
    "no_linln_panel": ("IB_NO_LINLN_PANEL", False),
    "linln_panel_max_m": ("IB_LINLN_PANEL_MAX_M", 8192),
    "no_ffn_infer": ("IB_NO_FFN_INFER", False),
    "ffn_infer_max_m": ("IB_FFN_INFER_MAX_M", 8192),
    "no_time_table": ("IB_NO_TIME_TABLE", False),
    "no_infer_chain": ("IB_NO_INFER_CHAIN", False),    # round 5: large-batch sampler on the training-shape launches
    "infer_chain_min_m": ("IB_INFER_CHAIN_MIN_M", 8193),
    "no_infer_split": ("IB_NO_INFER_SPLIT", False),    # round 5: the short last round of panels through the row-panel kernels
    "infer_split_max_rem": ("IB_INFER_SPLIT_MAX_REM", 96),   # swept: 88 panels +4.7 %, 119 panels -3.4 %
    # ---- padding of the D-wide projections
    "no_pad": ("IB_NO_PAD", False),
    "no_train_pad": ("IB_NO_TRAIN_PAD", False),
    "pad_min_m": ("IB_PAD_MIN_M", 2560),
    # ---- time MLP / MLP denoiser chain
    "no_time_fuse": ("IB_NO_TIME_FUSE", False),
    "no_time_bwd_fuse": ("IB_NO_TIME_BWD_FUSE", False),
    "no_tb_rider": ("IB_NO_TB_RIDER", False),
    "skip_time_bwd": ("IB_SKIP_TIME_BWD", False),      # timing only: drops work, never set outside tools/
    "no_chain": ("IB_NO_CHAIN", False),
    "chain_v1": ("IB_CHAIN_V1", False),
    "no_defer": ("IB_NO_DEFER", False),
    # ---- trainer
    "no_opt_fuse": ("IB_NO_OPT_FUSE", False),
    "no_early_opt": ("IB_NO_EARLY_OPT", False),
    "no_zero_copy": ("IB_NO_ZERO_COPY", False),
    "no_pinned_graphs": ("IB_NO_PINNED_GRAPHS", False),
    "async_inline": ("IB_ASYNC_INLINE", False),
    "no_bucket_opt": ("IB_NO_BUCKET_OPT", False),      # round 5: per-bucket optimizer launches under data parallelism
}


class Tuning:
    def __getattr__(self, name):                       # only reached when no override was set on the object
        try:
            env, default = _SPEC[name]
        except KeyError:
            raise AttributeError(f"tuning has no switch {name!r}") from None
        from . import hip
        if hip.measurement_build():
            v = os.environ.get(env)
            if v:
                if isinstance(default, bool):
                    return True
                return int(v) if isinstance(default, int) else v
        return default

    @staticmethod
    def spec():
        """{attribute: (environment variable, default)} -- documentation and tests"""
        return dict(_SPEC)


tuning = Tuning()
