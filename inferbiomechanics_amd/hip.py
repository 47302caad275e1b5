"""ctypes binding of ``libib_hip.so`` (the C-ABI declared in ``include/ib_hip.h``).

PyTorch is used here only for device memory and streams: every wrapper takes torch tensors, checks
shape / dtype / contiguity on the host (a faulting kernel can reset the whole node), and passes raw
device pointers + sizes + the current HIP stream to the library.  There is NO CPU or eager-PyTorch
fallback: if the library is missing or a call fails, a ``HipError`` is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IB_HIP_LIB: another build of the same library (A/B runs of kernel variants inside one gpurun call); never a fallback
LIB_PATH = os.environ.get("IB_HIP_LIB") or os.path.join(_HERE, "lib", "libib_hip.so")
# the measurement build of the same sources (-DIB_AB: environment A/B switches + in-kernel stamp hooks); selected ONLY through
# IB_HIP_LIB by tools/ and by the tests that compare kernel families -- the product never loads it
AB_LIB_PATH = os.path.join(_HERE, "lib", "ab", "libib_hip_ab.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ib_hip.h")

F32, BF16 = 0, 1
ACT = {"none": 0, "identity": 0, None: 0, "relu": 1, "tanh": 2, "sigmoid": 3, "silu": 4, "elu": 5}
OPT = {"sgd": 0, "adam": 1, "rmsprop": 2, "adagrad": 3, "adadelta": 4, "adamax": 5}
OPT_NUM_STATES = {"sgd": 0, "adam": 2, "rmsprop": 1, "adagrad": 1, "adadelta": 2, "adamax": 2}


class HipError(RuntimeError):
    pass


_lib = None

_c = ctypes
_vp, _i64, _i32, _f32, _sz = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_float, _c.c_size_t
_SIGS = {
    "ib_version": (_c.c_int, []),
    "ib_error_string": (_c.c_char_p, [_c.c_int]),
    "ib_linear_fwd": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _c.c_int, _vp, _i64, _vp,
                                 _i64, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_linear_dgrad": (_c.c_int, [_vp, _i64, _vp, _i64, _c.c_int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64,
                                   _c.c_int, _vp]),
    "ib_linear_dgrad_wt": (_c.c_int, [_vp, _i64, _vp, _i64, _c.c_int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64,
                                      _c.c_int, _vp]),
    "ib_transpose_multi": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "ib_linear_dgrad_skinny": (_c.c_int, [_vp, _i64, _vp, _i64, _c.c_int, _vp, _i64, _vp, _i64, _vp, _c.c_int, _i64, _i64,
                                          _i64, _c.c_int, _vp]),
    "ib_linear_wgrad_bias": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _c.c_int, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_linear_wgrad_workspace": (_sz, [_i64, _i64, _i64]),
    "ib_linear_wgrad": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _c.c_int, _vp, _sz, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_linear_ln_fwd_workspace": (_sz, [_i64, _i64, _i64]),
    "ib_linear_ln_fwd": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz,
                                    _i64, _i64, _i64, _f32, _c.c_int, _vp]),
    "ib_linear_ln_panel_workgroups": (_c.c_int, [_i64, _i64, _i64, _vp]),
    "ib_linear_ln_panel_fwd": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ib_linear_panel_workgroups": (_c.c_int, [_i64, _i64, _i64]),
    "ib_linear_panel_fwd": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ib_ffn_infer_workspace": (_sz, [_i64, _i64, _i64]),
    "ib_ffn_infer_workgroups": (_c.c_int, [_i64, _i64, _i64, _vp, _vp]),
    "ib_ffn_infer_fwd": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _i64, _i64, _f32, _vp]),
    "ib_linear_wgrad_slabs_workspace": (_sz, [_i64, _i64, _i64]),
    "ib_linear_wgrad_slabs": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _sz, _vp, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_linear_wgrad_slabs_multi": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "ib_linear_wgrad_slabs_multi_bias": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int,
                                                    _vp]),
    "ib_slab_reduce_multi": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "ib_step_reduce": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c.c_int, _vp, _vp, _vp, _vp, _vp,
                                  _c.c_int, _vp]),
    "ib_step_reduce_parts": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _c.c_int, _vp]),
    "ib_segment_colsum": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _c.c_int, _c.c_int, _c.c_int, _vp]),
    "ib_layernorm_bwd_reduce": (_c.c_int, [_vp, _sz, _vp, _vp, _c.c_int, _i64, _i64, _vp]),
    "ib_layernorm_fwd": (_c.c_int, [_vp, _vp, _c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32,
                                    _c.c_int, _vp]),
    "ib_layernorm_bwd_workspace": (_sz, [_i64, _i64]),
    "ib_layernorm_bwd": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp, _sz,
                                    _vp, _i64, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_attention_fwd": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_attention_bwd": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_tiny_matmul": (_c.c_int, [_vp, _c.c_int, _i64, _i64, _vp, _c.c_int, _i64, _i64, _vp, _c.c_int, _i64, _c.c_int, _i64, _i64,
                                  _i64, _vp]),
    "ib_attention_fwd_drop": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _c.c_float, _c.c_uint32, _c.c_int32, _vp,
                                         _c.c_int, _vp]),
    "ib_attention_bwd_drop": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _c.c_float, _c.c_uint32, _c.c_int32,
                                         _vp, _c.c_int, _vp]),
    "ib_attention_drop_mask": (_c.c_int, [_vp, _i64, _i64, _i64, _c.c_float, _c.c_uint32, _c.c_int32, _vp, _vp]),
    "ib_concat_keys": (_c.c_int, [_vp, _vp, _i32, _vp, _i64, _c.c_int, _vp]),
    "ib_cast": (_c.c_int, [_vp, _c.c_int, _vp, _c.c_int, _i64, _vp]),
    "ib_cast2d": (_c.c_int, [_vp, _i64, _c.c_int, _vp, _i64, _c.c_int, _i64, _i64, _vp]),
    "ib_regression_loss_workspace": (_sz, [_i64, _i64]),
    "ib_regression_loss": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _sz, _i64, _i64, _c.c_int, _vp]),
    "ib_regression_loss_strided": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _vp, _sz, _i64, _i64, _c.c_int, _vp]),
    "ib_mse_loss_workspace": (_sz, [_i64]),
    "ib_mse_loss": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _i64, _c.c_int, _vp]),
    "ib_mse_loss_partial": (_c.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _sz, _i64, _i64, _c.c_int, _vp]),
    "ib_mse_loss_finalize": (_c.c_int, [_vp, _sz, _vp, _i64, _vp]),
    "ib_optim_step": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "ib_optim_step_sources": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _i32, _vp, _vp, _vp, _c.c_int, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "ib_im2col_replicate": (_c.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _c.c_int, _c.c_int, _vp]),
    "ib_col2im_replicate": (_c.c_int, [_vp, _i64, _vp, _c.c_int, _vp, _i64, _i64, _i64, _c.c_int, _c.c_int, _vp]),
    "ib_dropout": (_c.c_int, [_vp, _vp, _i64, _f32, _c.c_uint32, _i32, _vp, _c.c_int, _vp]),
    "ib_debug_set_ffn_prof": (_c.c_int, [_vp]),
    "ib_ffn_chain_supported": (_c.c_int, [_i64, _i64]),
    "ib_ffn_chain_packed_elems": (_sz, [_i64, _i64]),
    "ib_ffn_chain_workgroups": (_c.c_int, [_i64, _i64, _i64, _vp]),
    "ib_ffn_chain_mask_bytes": (_sz, [_i64, _i64, _i64]),
    "ib_ffn_chain_pack": (_c.c_int, [_vp] * 9 + [_c.c_int, _i64, _i64, _vp]),
    "ib_ffn_chain_fwd": (_c.c_int, [_vp] * 23 + [_i64, _i64, _i64, _f32, _vp]),
    "ib_ffn_chain_bwd": (_c.c_int, [_vp] * 20 + [_i64, _i64, _i64, _vp]),
    "ib_ffn_chain_fwd_infer": (_c.c_int, [_vp] * 14 + [_i64, _i64, _i64, _f32, _vp]),
    "ib_ffn_chain_attn_workgroups": (_c.c_int, [_i64, _i64, _i64, _i64]),
    "ib_ffn_chain_attn_mask_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "ib_ffn_chain_fwd_attn": (_c.c_int, [_vp] * 25 + [_i64, _i64, _i64, _i64, _f32, _vp]),
    "ib_ffn_chain_bwd_attn": (_c.c_int, [_vp] * 19 + [_i64, _i64, _i64, _i64, _vp]),
    "ib_sqdiff_mean": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _c.c_int, _vp]),
    "ib_sqdiff_mean_bwd": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _c.c_int, _vp]),
    "ib_mask_by_threes": (_c.c_int, [_vp, _vp, _i64, _f32, _c.c_int, _vp]),
    "ib_mean_norm_error": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _c.c_int, _c.c_int, _c.c_int, _vp]),
    "ib_gather_rows_bwd": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "ib_diffusion_draw": (_c.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _c.c_uint64, _i32, _vp,
                                     _c.c_uint32, _c.c_int, _vp]),
    "ib_philox_words": (_c.c_int, [_vp, _i64, _c.c_uint64, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "ib_gather_windows": (_c.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _c.c_int, _vp, _vp, _vp]),
    "ib_gather_rows": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_mlp_chain_supported": (_c.c_int, [_i64, _i64, _c.c_int]),
    "ib_mlp_chain_packed_elems": (_sz, [_i64, _i64, _c.c_int]),
    "ib_mlp_chain_workgroups": (_c.c_int, [_i64, _vp]),
    "ib_mlp_chain_pack": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _c.c_int, _vp]),
    "ib_mlp_chain_partial_width": (_i64, [_i64, _i64, _c.c_int]),
    "ib_mlp_chain_train": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                      _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _c.c_int,
                                      _f32, _vp]),
    "ib_set_ptrs": (_c.c_int, [_vp, _c.c_int, _vp, _vp]),
    "ib_colsum_segments": (_c.c_int, [_vp, _i64, _i64, _c.c_int, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "ib_debug_set_chain_prof": (_c.c_int, [_vp]),
    "ib_debug_set_gemm_prof": (_c.c_int, [_vp]),
    "ib_debug_set_nt_prof": (_c.c_int, [_vp]),
    "ib_debug_stamp": (_c.c_int, [_vp, _vp]),
    "ib_time_mlp_fwd_supported": (_c.c_int, [_i64, _i64, _i64]),
    "ib_mlp_chain_prep": (_c.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                     _i64, _i64, _vp, _vp, _vp, _i64, _i64, _c.c_int, _vp, _vp]),
    "ib_time_mlp_fwd": (_c.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                   _i64, _i64, _vp]),
    "ib_linear_wgrad_slabs_multi_tb": (_c.c_int, [_c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int,
                                                  _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ib_time_mlp_bwd_supported": (_c.c_int, [_i64, _i64, _i64]),
    "ib_time_mlp_bwd_slab_count": (_c.c_int, [_i64]),
    "ib_optim_ticket_words": (_c.c_int, []),
    "ib_time_mlp_bwd": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ib_sum_partials": (_c.c_int, [_vp, _i64, _f32, _vp, _vp]),
    "ib_q_sample": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _c.c_int, _vp]),
    "ib_ddim_step": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i64, _c.c_int, _vp]),
    "ib_counter_add": (_c.c_int, [_vp, _i32, _vp]),
    "ib_batchnorm_fwd": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f32, _f32, _c.c_int,
                                    _c.c_int, _vp]),
    "ib_batchnorm_bwd": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _c.c_int, _c.c_int, _vp, _i64,
                                    _i64, _i64, _c.c_int, _c.c_int, _vp]),
    "ib_scale_by_device_scalar": (_c.c_int, [_vp, _vp, _i64, _c.c_int, _vp]),
    "ib_fill_i64": (_c.c_int, [_vp, _i64, _i64, _vp]),
    "ib_graph_begin": (_c.c_int, [_vp]),
    "ib_graph_end": (_c.c_int, [_vp, _c.POINTER(_vp)]),
    "ib_graph_launch": (_c.c_int, [_vp, _vp]),
    "ib_graph_destroy": (_c.c_int, [_vp]),
    "ib_stream_create": (_c.c_int, [_c.POINTER(_vp)]),
    "ib_stream_destroy": (_c.c_int, [_vp]),
    "ib_event_create": (_c.c_int, [_c.POINTER(_vp)]),
    "ib_event_record": (_c.c_int, [_vp, _vp]),
    "ib_event_elapsed_ms": (_c.c_int, [_vp, _vp, _c.POINTER(_f32)]),
    "ib_event_destroy": (_c.c_int, [_vp]),
    "ib_selftest_tr16": (_c.c_int, [_vp, _vp, _vp]),
    "ib_debug_set_ablate": (_c.c_int, [_c.c_int]),
    "ib_debug_last_path": (_c.c_int, []),
}


def declared_symbols() -> List[str]:
    """Every function the public header declares (used by the CPU test that checks the exports)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ib_[a-z0-9_]+)\s*\(", src)))


_HOST_ONLY = ("_workspace", "_supported", "_workgroups", "_packed_elems", "_partial_width", "_last_path", "_slab_count", "_ticket_words", "_mask_bytes")   # pure host queries: no launch, no stream


class _DryRunLib:
    """TEST-ONLY stand-in (tests/test_plumbing_cpu.py): marshals every argument through the real ctypes
    signature (so arity / type errors surface) and returns IB_OK without launching anything.  It computes
    nothing -- outputs are left untouched -- and is never selected by product code."""

    def __init__(self, real):
        self._real = real
        self.calls = []

    def __getattr__(self, name):
        res, args = _SIGS[name]
        real = getattr(self._real, name)
        if name.endswith(_HOST_ONLY) or name in ("ib_version", "ib_error_string"):
            return real

        def call(*a):
            if len(a) != len(args):
                raise TypeError(f"{name}: expected {len(args)} arguments, got {len(a)}")
            for v, t in zip(a, args):
                if v is not None or t not in (_vp,):
                    t.from_param(v) if hasattr(t, "from_param") else None
            self.calls.append(name)
            return 0
        return call


class _TimingLib:
    """Brackets every C-ABI call with HIP events recorded on the launch stream (bench.py's roofline leg:
    per-entry-point device time, measured live on the stream the kernels are launched on)."""

    def __init__(self, real):
        self._real = real
        self.records = []      # (name, int_args, ev0, ev1)

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if not name.startswith("ib_") or name.startswith(("ib_event", "ib_graph")) or name.endswith(_HOST_ONLY) \
                or name in ("ib_version", "ib_error_string"):
            return real

        def call(*a):
            e0, e1 = Event(), Event()
            e0.record()
            rc = real(*a)
            e1.record()
            self.records.append((name, tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 31)), e0, e1))
            return rc
        return call

    def summary(self):
        """{(name, int_args): [ms, ...]} (synchronises)"""
        out = {}
        for name, ints, e0, e1 in self.records:
            out.setdefault((name, ints), []).append(e0.elapsed_ms(e1))
        return out


class _RecordingLib:
    """Records every C-ABI call (name + raw arguments) of one eager step so bench.py can re-issue each distinct
    call back-to-back inside a hipGraph and time it with HIP events (no host launch overhead in the number)."""

    SKIP = ("ib_optim_step", "ib_counter_add")       # mutate state when repeated

    def __init__(self, real):
        self._real = real
        self.calls = []      # (name, args)
        self.paths = []      # per call: the kernel family the entry point dispatched to (IB_PATH_*, 0 = no dispatch)
        self.notes = {}      # index into calls -> (algorithmic flops, algorithmic bytes) for calls whose shapes sit in arrays

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if not name.startswith("ib_") or name.startswith(("ib_event", "ib_graph")) or name.endswith(_HOST_ONLY) \
                or name in ("ib_version", "ib_error_string"):
            return real

        def call(*a):
            global _work_note
            if _work_note is not None:
                self.notes[len(self.calls)] = _work_note
                _work_note = None
            self.calls.append((name, a))
            self._real.ib_debug_last_path()            # clear
            rc = real(*a)
            self.paths.append(int(self._real.ib_debug_last_path()))
            return rc
        return call


PATH_NAMES = {0: "-", 1: "nt256x128", 2: "tn256x128", 3: "ring128", 4: "generic", 5: "smallm", 6: "skinny", 7: "wgrad_small",
              8: "ring_multi", 9: "linear_ln", 10: "chain_v2", 11: "chain_v1", 12: "tn256x256", 13: "nt_splitk", 14: "ffn_chain", 15: "linear_ln_panel", 16: "ffn_infer", 17: "linear_panel"}
_work_note = None     # set by a wrapper right before a grouped launch: (flops, bytes) of that launch, for bench.py


class _StampLib:
    """TIMING-ONLY (tools/timeline.py): brackets every C-ABI launch with two one-thread kernels that write the
    device wall clock, on the launch's own stream -- under hipGraph capture they become graph nodes, so a replay
    yields the device-side timeline of the step (which launches overlap, where the gaps are)."""

    def __init__(self, real, slots: int = 4096, only=None):
        self._real = real
        self.buf = torch.zeros(slots, dtype=torch.int64, device="cuda")
        self.calls = []      # (name, int args)
        self.only = only     # None = every launch; else the entry-point names to bracket (less perturbation)

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if not name.startswith("ib_") or name.startswith(("ib_event", "ib_graph", "ib_debug")) \
                or name.endswith(_HOST_ONLY) or name in ("ib_version", "ib_error_string") \
                or (self.only is not None and name not in self.only):
            return real

        def call(*a):
            k = len(self.calls)
            base = self.buf.data_ptr() + 16 * k
            self._real.ib_debug_stamp(ctypes.c_void_p(base), a[-1])
            rc = real(*a)
            self._real.ib_debug_stamp(ctypes.c_void_p(base + 8), a[-1])
            self.calls.append((name, tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 31)),
                               a[-1].value if hasattr(a[-1], "value") else a[-1]))
            return rc
        return call

    def timeline(self):
        """[(name, ints, stream, start_us, end_us)] relative to the first stamp (synchronises)"""
        torch.cuda.synchronize()
        t = self.buf.cpu().tolist()
        n = len(self.calls)
        t0 = min(t[2 * k] for k in range(n))
        return [(self.calls[k][0], self.calls[k][1], self.calls[k][2], (t[2 * k] - t0) * 0.01, (t[2 * k + 1] - t0) * 0.01)
                for k in range(n)]


class stamp_launches:
    def __init__(self, only=None):
        self.only = only

    def __enter__(self):
        global _lib
        self._saved = lib()
        self.lib = _StampLib(self._saved, only=self.only)
        _lib = self.lib
        return self.lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


class record_launches:
    def __enter__(self):
        global _lib
        self._saved = lib()
        self.rec = _RecordingLib(self._saved)
        _lib = self.rec
        return self.rec

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


_free_streams: dict = {}          # device index -> handles whose torch wrapper has died: handed out again, never destroyed


class _OwnedStream:
    """returns an ib_stream_create handle to the free list when its torch wrapper dies.  Not destroyed: a loop that adopted a
    trainer's stream (HipTrainer.adopt_stream) may have left it as the thread's CURRENT torch stream beyond the trainer's
    life -- a destroyed handle there is a segfault at the next launch, a recycled one is just a stream."""

    def __init__(self, ptr, index):
        self.ptr, self.index = ptr, index

    def __del__(self):
        try:
            _free_streams.setdefault(self.index, []).append(self.ptr)
        except Exception:
            pass


def new_stream(device=None) -> "torch.cuda.Stream":
    """A stream of the library's own (ib_stream_create), seen by torch as an ExternalStream.  NOT torch.cuda.Stream(): torch
    hands its streams out round-robin from a pool of 32 per device and c10d's communication stream comes from the same pool,
    so in a process that has made 32 streams (a few trainers: each has a dozen side branches) a new side branch IS an older
    stream -- possibly c10d's, which a capture then drags in (the watchdog's event query fails: process abort), or a live
    plan's branch (silently serialised)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    free = _free_streams.get(dev.index)
    if free:
        ptr = free.pop()
    else:
        out = ctypes.c_void_p(0)
        with torch.cuda.device(dev):
            _check(lib().ib_stream_create(ctypes.byref(out)), "ib_stream_create")
        ptr = out.value
    s = torch.cuda.ExternalStream(ptr, device=dev)
    s._ib_owner = _OwnedStream(ptr, dev.index)
    return s


def time_recorded_call(name: str, args, reps: int = 20, rounds: int = 3) -> float:
    """average device time (us) of one launch of a recorded call: `reps` launches captured in a graph on a side
    stream, replayed `rounds` times between two HIP events recorded on that stream"""
    fn = getattr(lib(), name)
    s = new_stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    a = list(args)
    a[-1] = sp                                   # every entry point takes the stream last
    with torch.cuda.stream(s):
        for _ in range(2):
            _check(fn(*a), name)
        g = Graph()
        g.begin()
        for _ in range(reps):
            _check(fn(*a), name)
        g.end()
        g.launch()
        e0, e1 = Event(), Event()
        e0.record()
        for _ in range(rounds):
            g.launch()
        e1.record()
        ms = e0.elapsed_ms(e1)
    torch.cuda.synchronize()
    return ms * 1e3 / (reps * rounds)


class time_launches:
    """context manager: `with hip.time_launches() as tl: ...; tl.summary()`"""

    def __enter__(self):
        global _lib
        self._saved = lib()
        self.tl = _TimingLib(self._saved)
        _lib = self.tl
        return self.tl

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


_dry_run = False


def set_dry_run(on: bool):
    """TEST-ONLY: see _DryRunLib."""
    global _dry_run, _lib
    _dry_run = bool(on)
    _lib = None


def lib():
    """Load the library (once).  Raises HipError if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C inferbiomechanics_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = _DryRunLib(l) if _dry_run else l
    return _lib


def measurement_build() -> bool:
    """True while the -DIB_AB build of the library is the one loaded (tools/ and A/B tests select it through IB_HIP_LIB)"""
    return os.path.basename(LIB_PATH) != "libib_hip.so"


def _check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed: {lib().ib_error_string(rc).decode()} (code {rc})")


def stream_ptr() -> int:
    if _dry_run:
        return 0
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise HipError(f"unsupported storage dtype {t}; the HIP path computes in float32 or bfloat16")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype: Optional[torch.dtype] = None, dim: Optional[int] = None):
    if not isinstance(t, torch.Tensor) or not (t.is_cuda or _dry_run):
        raise HipError(f"{name}: expected a tensor in device (HBM) memory")
    if dtype is not None and t.dtype != dtype:
        raise HipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if dim is not None and t.dim() != dim:
        raise HipError(f"{name}: expected {dim}-D, got shape {tuple(t.shape)}")


def _mat(t: torch.Tensor, name: str, dtype: torch.dtype):
    """2-D row-major view with unit inner stride -> (rows, cols, ld)."""
    _req(t, name, dtype, 2)
    if t.stride(1) != 1 and t.shape[1] != 1:
        raise HipError(f"{name}: inner stride must be 1, got strides {t.stride()}")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    if ld < t.shape[1]:
        raise HipError(f"{name}: leading dimension {ld} < columns {t.shape[1]}")
    return t.shape[0], t.shape[1], ld


# --------------------------------------------------------------------------------------------
# Linear family
# --------------------------------------------------------------------------------------------
def linear_fwd(x, w, bias, y, act="none", z=None, add_div=None, add_mod=None, seg=0):
    dt = x.dtype
    M, K, ldx = _mat(x, "x", dt)
    N, Kw, ldw = _mat(w, "w", dt)
    My, Ny, ldy = _mat(y, "y", dt)
    if Kw != K or My != M or Ny != N:
        raise HipError(f"linear_fwd shape mismatch: x{tuple(x.shape)} w{tuple(w.shape)} y{tuple(y.shape)}")
    if bias is not None:
        _req(bias, "bias", torch.float32, 1)
        if bias.numel() != N or not bias.is_contiguous():
            raise HipError("bias must be a contiguous fp32 vector of length N")
    ldz = 0
    if z is not None:
        Mz, Nz, ldz = _mat(z, "z", dt)
        if (Mz, Nz) != (M, N):
            raise HipError("z shape mismatch")
    ld_ad = ld_am = 0
    if add_div is not None:
        r, c, ld_ad = _mat(add_div, "add_div", dt)
        if seg <= 0 or c != N or r * seg < M:
            raise HipError(f"add_div must be [ceil(M/seg), N]; got {tuple(add_div.shape)} with seg={seg}, M={M}")
    if add_mod is not None:
        r, c, ld_am = _mat(add_mod, "add_mod", dt)
        if seg <= 0 or c != N or r < min(seg, M):
            raise HipError(f"add_mod must be [seg, N]; got {tuple(add_mod.shape)} with seg={seg}")
    _check(lib().ib_linear_fwd(_ptr(x), ldx, _ptr(w), ldw, _ptr(bias), _ptr(add_div), ld_ad, _ptr(add_mod), ld_am,
                               int(seg), ACT[act], _ptr(y), ldy, _ptr(z), ldz, M, N, K, dtype_code(dt), stream_ptr()),
           "ib_linear_fwd")
    return y


def linear_dgrad(dz, w, dx, act_below="none", aux=None, addend=None):
    dt = dz.dtype
    M, N, lddz = _mat(dz, "dz", dt)
    Nw, K, ldw = _mat(w, "w", dt)
    Mx, Kx, lddx = _mat(dx, "dx", dt)
    if Nw != N or Mx != M or Kx != K:
        raise HipError(f"linear_dgrad shape mismatch: dz{tuple(dz.shape)} w{tuple(w.shape)} dx{tuple(dx.shape)}")
    ldaux = 0
    if ACT[act_below] != 0:
        if aux is None:
            raise HipError("linear_dgrad: activation below needs aux")
        Ma, Ka, ldaux = _mat(aux, "aux", dt)
        if (Ma, Ka) != (M, K):
            raise HipError("aux shape mismatch")
    ldadd = 0
    if addend is not None:
        Ma, Ka, ldadd = _mat(addend, "addend", dt)
        if (Ma, Ka) != (M, K):
            raise HipError("addend shape mismatch")
    _check(lib().ib_linear_dgrad(_ptr(dz), lddz, _ptr(w), ldw, ACT[act_below], _ptr(aux), ldaux, _ptr(addend), ldadd,
                                 _ptr(dx), lddx, M, N, K, dtype_code(dt), stream_ptr()), "ib_linear_dgrad")
    return dx


def linear_dgrad_wt(dz, wt, dx, act_below="none", aux=None, addend=None) -> bool:
    """dx = (dz . w) * act'(aux) + addend with the weight given TRANSPOSED (wt = w^T [K, N], see transpose_multi): the
    large-M bf16 path (256 x 128 LDS-DMA kernel).  False = the problem does not qualify, nothing was launched."""
    dt = dz.dtype
    if dt != torch.bfloat16:
        return False
    M, N, lddz = _mat(dz, "dz", dt)
    K, Nw, ldwt = _mat(wt, "wt", dt)
    Mx, Kx, lddx = _mat(dx, "dx", dt)
    if Nw != N or Mx != M or Kx != K:
        raise HipError(f"linear_dgrad_wt: shape mismatch dz{tuple(dz.shape)} wt{tuple(wt.shape)} dx{tuple(dx.shape)}")
    ldaux = ldadd = 0
    if ACT[act_below] != 0:
        if aux is None:
            raise HipError("linear_dgrad_wt: act_below needs aux")
        Ma, Ka, ldaux = _mat(aux, "aux", dt)
        if (Ma, Ka) != (M, K):
            raise HipError("linear_dgrad_wt: aux shape mismatch")
    if addend is not None:
        Ma, Ka, ldadd = _mat(addend, "addend", dt)
        if (Ma, Ka) != (M, K):
            raise HipError("linear_dgrad_wt: addend shape mismatch")
    rc = lib().ib_linear_dgrad_wt(_ptr(dz), lddz, _ptr(wt), ldwt, ACT[act_below], _ptr(aux) if ACT[act_below] != 0 else None,
                                  ldaux, _ptr(addend), ldadd, _ptr(dx), lddx, M, N, K, dtype_code(dt), stream_ptr())
    if rc == -5:          # IB_E_UNSUPPORTED
        return False
    _check(rc, "ib_linear_dgrad_wt")
    return True


def transpose_multi(pairs):
    """pairs: [(src [R, C], dst [C, R])] bf16 -- every dst = src^T in ONE launch (at most 32 matrices per launch)"""
    for a in range(0, len(pairs), 32):
        chunk = pairs[a:a + 32]
        n = len(chunk)
        geo = []
        for src, dst in chunk:
            R, C, lds = _mat(src, "src", torch.bfloat16)
            Cd, Rd, ldd = _mat(dst, "dst", torch.bfloat16)
            if (Cd, Rd) != (C, R):
                raise HipError(f"transpose_multi: dst must be {C} x {R}, got {tuple(dst.shape)}")
            geo.append((R, C, lds, ldd))
        cv = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        S = (ctypes.c_void_p * n)(*[p_[0].data_ptr() for p_ in chunk])
        D = (ctypes.c_void_p * n)(*[p_[1].data_ptr() for p_ in chunk])
        LS = (ctypes.c_int64 * n)(*[g_[2] for g_ in geo])
        LD = (ctypes.c_int64 * n)(*[g_[3] for g_ in geo])
        Rs = (ctypes.c_int64 * n)(*[g_[0] for g_ in geo])
        Cs = (ctypes.c_int64 * n)(*[g_[1] for g_ in geo])
        _check(lib().ib_transpose_multi(n, cv(S), cv(LS), cv(D), cv(LD), cv(Rs), cv(Cs), BF16, stream_ptr()),
               "ib_transpose_multi")


def linear_dgrad_skinny(dz, w, dx, act_below="none", aux=None, dbias=None, accumulate=False) -> bool:
    """few-row dgrad with the bias gradient of the layer below fused (dbias (+)= column sums of dx); False when the shape
    does not qualify (the caller then uses linear_dgrad + a column-sum launch)"""
    dt = dz.dtype
    M, N, lddz = _mat(dz, "dz", dt)
    Nw, K, ldw = _mat(w, "w", dt)
    Mx, Kx, lddx = _mat(dx, "dx", dt)
    if Nw != N or Mx != M or Kx != K:
        raise HipError(f"linear_dgrad_skinny shape mismatch: dz{tuple(dz.shape)} w{tuple(w.shape)} dx{tuple(dx.shape)}")
    ldaux = 0
    if ACT[act_below] != 0:
        if aux is None:
            raise HipError("linear_dgrad_skinny: activation below needs aux")
        Ma, Ka, ldaux = _mat(aux, "aux", dt)
        if (Ma, Ka) != (M, K):
            raise HipError("aux shape mismatch")
    if dbias is not None:
        _req(dbias, "dbias", torch.float32)
        if dbias.numel() != K or not dbias.is_contiguous():
            raise HipError("dbias must be a contiguous fp32 vector of length K")
    rc = lib().ib_linear_dgrad_skinny(_ptr(dz), lddz, _ptr(w), ldw, ACT[act_below], _ptr(aux), ldaux, _ptr(dx), lddx,
                                      _ptr(dbias), int(accumulate), M, N, K, dtype_code(dt), stream_ptr())
    if rc == -5:                       # IB_E_UNSUPPORTED
        return False
    _check(rc, "ib_linear_dgrad_skinny")
    return True


def linear_wgrad_bias(dz, x, dw, dbias, accumulate=False) -> bool:
    """dw (+)= dz^T x and dbias (+)= column sums of dz from ONE launch (short reductions); False when the shape does not
    qualify (the caller then issues linear_wgrad + a column-sum launch)"""
    dt = dz.dtype
    M, N, lddz = _mat(dz, "dz", dt)
    Mx, K, ldx = _mat(x, "x", dt)
    if Mx != M:
        raise HipError("linear_wgrad_bias: dz / x row counts differ")
    Nw, Kw, lddw = _mat(dw, "dw", torch.float32)
    if (Nw, Kw) != (N, K):
        raise HipError(f"linear_wgrad_bias: dw must be [{N},{K}], got {tuple(dw.shape)}")
    _req(dbias, "dbias", torch.float32)
    if dbias.numel() != N or not dbias.is_contiguous():
        raise HipError("dbias must be a contiguous fp32 vector of length N")
    rc = lib().ib_linear_wgrad_bias(_ptr(dz), lddz, _ptr(x), ldx, _ptr(dw), lddw, _ptr(dbias), int(accumulate), M, N, K,
                                    dtype_code(dt), stream_ptr())
    if rc == -5:
        return False
    _check(rc, "ib_linear_wgrad_bias")
    return True


def linear_wgrad_workspace_bytes(M, N, K) -> int:
    return int(lib().ib_linear_wgrad_workspace(M, N, K))


def linear_wgrad(dz, x, dw, workspace, accumulate=False):
    dt = dz.dtype
    M, N, lddz = _mat(dz, "dz", dt)
    Mx, K, ldx = _mat(x, "x", dt)
    Nw, Kw, lddw = _mat(dw, "dw", torch.float32)
    if Mx != M or Nw != N or Kw != K:
        raise HipError(f"linear_wgrad shape mismatch: dz{tuple(dz.shape)} x{tuple(x.shape)} dw{tuple(dw.shape)}")
    need = linear_wgrad_workspace_bytes(M, N, K)
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    if need > wsb:
        raise HipError(f"linear_wgrad: workspace of {need} bytes required, got {wsb}")
    _check(lib().ib_linear_wgrad(_ptr(dz), lddz, _ptr(x), ldx, _ptr(dw), lddw, int(accumulate), _ptr(workspace), wsb,
                                 M, N, K, dtype_code(dt), stream_ptr()), "ib_linear_wgrad")
    return dw


def linear_ln_fwd(x, w, bias, res, gamma, beta, y, workspace, eps=1e-5) -> bool:
    """y = LayerNorm(res + x w^T + bias) as a K-split GEMM + one fused reduction / LayerNorm launch.  Returns False when
    the shape does not qualify (the caller then issues linear_fwd + layernorm_fwd)."""
    dt = x.dtype
    M, K, ldx = _mat(x, "x", dt)
    N, Kw, ldw = _mat(w, "w", dt)
    if Kw != K:
        raise HipError("linear_ln_fwd: x / w reduction lengths differ")
    My, Ny, ldy = _mat(y, "y", dt)
    if (My, Ny) != (M, N):
        raise HipError("linear_ln_fwd: y must be [M, N]")
    ldres = 0
    if res is not None:
        Mr, Nr, ldres = _mat(res, "res", dt)
        if (Mr, Nr) != (M, N):
            raise HipError("linear_ln_fwd: res must be [M, N]")
    for t, n in ((gamma, "gamma"), (beta, "beta")) + (((bias, "bias"),) if bias is not None else ()):
        _req(t, n, torch.float32, 1)
        if t.numel() != N:
            raise HipError(f"linear_ln_fwd: {n} must be fp32 [N]")
    need = int(lib().ib_linear_ln_fwd_workspace(M, N, K))
    wsb = workspace.numel() * workspace.element_size()
    if wsb < need:
        raise HipError(f"linear_ln_fwd: workspace of {need} bytes required, got {wsb}")
    rc = lib().ib_linear_ln_fwd(_ptr(x), ldx, _ptr(w), ldw, _ptr(bias), _ptr(res), ldres, _ptr(gamma), _ptr(beta), _ptr(y),
                                ldy, None, None, None, _ptr(workspace), wsb, M, N, K, float(eps), dtype_code(dt),
                                stream_ptr())
    if rc == -5:
        return False
    _check(rc, "ib_linear_ln_fwd")
    return True


def linear_ln_panel_ok(M: int, N: int, K: int) -> bool:
    return bool(lib().ib_linear_ln_panel_workgroups(int(M), int(N), int(K), None))


def linear_ln_panel_fwd(x, w_packed, bias, res, gamma, beta, y, eps=1e-5) -> bool:
    """y = LayerNorm(res + x W^T + bias) for a [512, 512] weight given as its packed image (ffn_chain_pack), one launch over
    row panels (csrc/linln_panel.hip).  Returns False when the shape does not qualify."""
    dt = torch.bfloat16
    M, K, ldx = _mat(x, "x", dt)
    My, N, ldy = _mat(y, "y", dt)
    _req(w_packed, "w_packed", dt, 1)
    if My != M or w_packed.numel() < N * K or not w_packed.is_contiguous():
        raise HipError("linear_ln_panel_fwd: y must be [M, N], w_packed the [N x K] image")
    ldres = 0
    if res is not None:
        Mr, Nr, ldres = _mat(res, "res", dt)
        if (Mr, Nr) != (M, N):
            raise HipError("linear_ln_panel_fwd: res must be [M, N]")
    for t, n in ((gamma, "gamma"), (beta, "beta")) + (((bias, "bias"),) if bias is not None else ()):
        _req(t, n, torch.float32, 1)
        if t.numel() != N:
            raise HipError(f"linear_ln_panel_fwd: {n} must be fp32 [N]")
    rc = lib().ib_linear_ln_panel_fwd(_ptr(x), ldx, _ptr(w_packed), _ptr(bias), _ptr(res), ldres, _ptr(gamma), _ptr(beta),
                                      _ptr(y), ldy, M, N, K, float(eps), stream_ptr())
    if rc == -5:
        return False
    _check(rc, "ib_linear_ln_panel_fwd")
    return True


def linear_panel_ok(M: int, N: int, K: int) -> bool:
    return bool(lib().ib_linear_panel_workgroups(int(M), int(N), int(K)))


def linear_panel_fwd(x, w_packed, bias, y) -> bool:
    """y = x W^T + bias, K = 512, N a multiple of 512, W given as N / 512 packed [512 x 512] images (csrc/linln_panel.hip)"""
    dt = torch.bfloat16
    M, K, ldx = _mat(x, "x", dt)
    My, N, ldy = _mat(y, "y", dt)
    _req(w_packed, "w_packed", dt, 1)
    if My != M or w_packed.numel() < N * K or not w_packed.is_contiguous():
        raise HipError("linear_panel_fwd: y must be [M, N], w_packed the N / 512 images")
    if bias is not None:
        _req(bias, "bias", torch.float32, 1)
        if bias.numel() != N:
            raise HipError("linear_panel_fwd: bias must be fp32 [N]")
    rc = lib().ib_linear_panel_fwd(_ptr(x), ldx, _ptr(w_packed), _ptr(bias), _ptr(y), ldy, M, N, K, stream_ptr())
    if rc == -5:
        return False
    _check(rc, "ib_linear_panel_fwd")
    return True


def ffn_infer_panels(M: int, d: int, ffn: int) -> int:
    """panel count of ffn_infer_fwd (0: the shape is not supported)"""
    n = ctypes.c_int32(0)
    return int(n.value) if lib().ib_ffn_infer_workgroups(int(M), int(d), int(ffn), None, ctypes.byref(n)) else 0


def ffn_infer_fwd(x1, packed, b1, b2, gamma, beta, y, workspace, eps=1e-5):
    """y = LayerNorm(x1 + W2 ReLU(W1 x1 + b1) + b2) with frozen weights from a layer's packed image: a launch over
    (panel, hidden chunk) workgroups leaving fp32 partial products + the slab-reduction LayerNorm launch
    (csrc/linln_panel.hip::ffn_coop_kernel); workspace: ib_ffn_infer_workspace bytes"""
    dt = torch.bfloat16
    M, d, ldx = _mat(x1, "x1", dt)
    My, dy, ldy = _mat(y, "y", dt)
    ffn = b1.numel()
    if (My, dy) != (M, d) or ldx != d or ldy != d:
        raise HipError("ffn_infer_fwd: x1 / y must be contiguous [M, d]")
    _req(packed, "packed", dt, 1)
    for t, n, k in ((b1, "b1", ffn), (b2, "b2", d), (gamma, "gamma", d), (beta, "beta", d)):
        _req(t, n, torch.float32, 1)
        if t.numel() != k:
            raise HipError(f"ffn_infer_fwd: {n} must be fp32 [{k}]")
    if not ffn_infer_panels(M, d, ffn):
        raise HipError(f"ffn_infer_fwd: shape [M = {M}, d = {d}, ffn = {ffn}] is not supported")
    if packed.numel() < ffn_chain_packed_elems(d, ffn):
        raise HipError("ffn_infer_fwd: packed image too small")
    wsb = workspace.numel() * workspace.element_size()
    _check(lib().ib_ffn_infer_fwd(_ptr(x1), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(workspace),
                                  wsb, M, d, ffn, float(eps), stream_ptr()), "ib_ffn_infer_fwd")


def linear_wgrad_slabs(dz, x, workspace) -> int:
    """split-M partial slabs of dw = dz^T x into `workspace`; returns the slab count (see slab_reduce_multi)"""
    dt = dz.dtype
    M, N, lddz = _mat(dz, "dz", dt)
    Mx, K, ldx = _mat(x, "x", dt)
    if Mx != M:
        raise HipError("linear_wgrad_slabs: dz / x row counts differ")
    need = int(lib().ib_linear_wgrad_slabs_workspace(M, N, K))
    wsb = workspace.numel() * workspace.element_size()
    if need > wsb:
        raise HipError(f"linear_wgrad_slabs: workspace of {need} bytes required, got {wsb}")
    n = ctypes.c_int(0)
    _check(lib().ib_linear_wgrad_slabs(_ptr(dz), lddz, _ptr(x), ldx, _ptr(workspace), wsb,
                                       ctypes.cast(ctypes.pointer(n), ctypes.c_void_p), M, N, K, dtype_code(dt),
                                       stream_ptr()), "ib_linear_wgrad_slabs")
    return n.value if not _dry_run else 1


def linear_wgrad_slabs_multi(problems, bias_parts=None, time_bwd=None):
    """problems: [(dz, x, workspace)] -- split-M slabs of every dW = dz^T x in ONE launch.  Returns the slab counts, or
    None when the shapes do not all qualify for the ring kernel (issue linear_wgrad_slabs one by one then).
    bias_parts: per problem None or an fp32 [32, N] tensor that receives the bias gradient's per-slice partial sums (rows
    0 .. slab count - 1).
    time_bwd = (de, w2, zu, s, dw1_slabs, db1_slabs): the operands of time_mlp_bwd() -- its workgroups ride in this launch
    (ib_linear_wgrad_slabs_multi_tb).  None is returned when the combined launch does not qualify: NOTHING was launched."""
    n = len(problems)
    if bias_parts is not None:
        for (dz, _, _), bp in zip(problems, bias_parts):
            if bp is not None:
                _req(bp, "bias_part", torch.float32, 2)
                if bp.shape[0] < 32 or bp.shape[1] != dz.shape[1] or not bp.is_contiguous():
                    raise HipError("linear_wgrad_slabs_multi: bias_part must be a contiguous fp32 [32, N]")
    dt = problems[0][0].dtype
    geo = []
    for dz, x, ws in problems:
        M, N, lddz = _mat(dz, "dz", dt)
        Mx, K, ldx = _mat(x, "x", dt)
        if Mx != M:
            raise HipError("linear_wgrad_slabs_multi: dz / x row counts differ")
        need = int(lib().ib_linear_wgrad_slabs_workspace(M, N, K))
        if ws.numel() * ws.element_size() < need:
            raise HipError("linear_wgrad_slabs_multi: workspace too small")
        geo.append((M, N, K, lddz, ldx))
    if _dry_run:
        lib().ib_linear_wgrad_slabs_multi(n, None, None, None, None, None, None, None, None, None, None, dtype_code(dt), None)
        return [1] * n
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    A = (ctypes.c_void_p * n)(*[p_[0].data_ptr() for p_ in problems])
    X = (ctypes.c_void_p * n)(*[p_[1].data_ptr() for p_ in problems])
    W = (ctypes.c_void_p * n)(*[p_[2].data_ptr() for p_ in problems])
    WB = (ctypes.c_size_t * n)(*[p_[2].numel() * p_[2].element_size() for p_ in problems])
    LA = (ctypes.c_int64 * n)(*[g_[3] for g_ in geo])
    LX = (ctypes.c_int64 * n)(*[g_[4] for g_ in geo])
    Ms = (ctypes.c_int64 * n)(*[g_[0] for g_ in geo])
    Ns = (ctypes.c_int64 * n)(*[g_[1] for g_ in geo])
    Ks = (ctypes.c_int64 * n)(*[g_[2] for g_ in geo])
    out = (ctypes.c_int32 * n)()
    global _work_note
    es_ = 2 if dt == torch.bfloat16 else 4
    if isinstance(_lib, _RecordingLib):
        _work_note = (sum(2 * g_[0] * g_[1] * g_[2] for g_ in geo),
                      sum((g_[0] * g_[1] + g_[0] * g_[2]) * es_ + g_[1] * g_[2] * 4 for g_ in geo),
                      [[g_[0], g_[1], g_[2]] for g_ in geo])
    if time_bwd is not None:
        if bias_parts is not None and any(b is not None for b in bias_parts):
            raise HipError("linear_wgrad_slabs_multi: time_bwd rider and bias_parts cannot be combined")
        de, w2, zu, s_, sw, sb = time_bwd
        bt = torch.bfloat16
        B_, out_, ld_de = _mat(de, "de", bt)
        o2, hid_, ldw2 = _mat(w2, "w2", bt)
        Bz, hz, ldzu = _mat(zu, "zu", bt)
        Bs, temb_, lds_ = _mat(s_, "s", bt)
        nsl = time_mlp_bwd_slab_count(B_)
        if o2 != out_ or hz != hid_ or Bz != B_ or Bs != B_ or tuple(sw.shape) != (nsl, hid_, temb_) \
                or tuple(sb.shape) != (nsl, hid_) or sw.dtype != torch.float32 or sb.dtype != torch.float32 \
                or not sw.is_contiguous() or not sb.is_contiguous():
            raise HipError("linear_wgrad_slabs_multi: time_bwd operands do not chain")
        rc = lib().ib_linear_wgrad_slabs_multi_tb(n, cv(A), cv(LA), cv(X), cv(LX), cv(W), cv(WB), cv(out), cv(Ms), cv(Ns), cv(Ks),
                                                  dtype_code(dt), _ptr(de), ld_de, _ptr(w2), ldw2, _ptr(zu), ldzu, _ptr(s_), lds_,
                                                  _ptr(sw), _ptr(sb), B_, temb_, hid_, out_, stream_ptr())
    elif bias_parts is not None and any(b is not None for b in bias_parts):
        BP = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in bias_parts])
        rc = lib().ib_linear_wgrad_slabs_multi_bias(n, cv(A), cv(LA), cv(X), cv(LX), cv(W), cv(WB), cv(BP), cv(out), cv(Ms),
                                                    cv(Ns), cv(Ks), dtype_code(dt), stream_ptr())
    else:
        rc = lib().ib_linear_wgrad_slabs_multi(n, cv(A), cv(LA), cv(X), cv(LX), cv(W), cv(WB), cv(out), cv(Ms), cv(Ns), cv(Ks),
                                               dtype_code(dt), stream_ptr())
    if rc == -5:          # IB_E_UNSUPPORTED
        return None
    _check(rc, "ib_linear_wgrad_slabs_multi")
    return list(out)


def slab_reduce_multi(items, accumulate=False):
    """items: [(workspace, nslab, dw)] with dw fp32 2-D [N, K] (row pitch % 4 == 0): dw (+)= sum of its slabs"""
    n = len(items)
    for ws, ns, dw in items:
        _req(ws, "workspace")
        _mat(dw, "dw", torch.float32)
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    slabs = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    nslab = (ctypes.c_int32 * n)(*[int(it[1]) for it in items])
    dws = (ctypes.c_void_p * n)(*[it[2].data_ptr() for it in items])
    ldd = (ctypes.c_int64 * n)(*[it[2].stride(0) for it in items])
    Ns = (ctypes.c_int32 * n)(*[it[2].shape[0] for it in items])
    Ks = (ctypes.c_int32 * n)(*[it[2].shape[1] for it in items])
    _check(lib().ib_slab_reduce_multi(n, cv(slabs), cv(nslab), cv(dws), cv(ldd), cv(Ns), cv(Ks), int(accumulate),
                                      stream_ptr()), "ib_slab_reduce_multi")


def step_reduce(items, part, rows: int, segs, accumulate=False):
    """slab_reduce_multi(items) + colsum_segments(part, rows, segs) as ONE launch"""
    n = len(items)
    for ws, ns, dw in items:
        _req(ws, "workspace")
        _mat(dw, "dw", torch.float32)
    pr, pc, ld = _mat(part, "part", torch.float32)
    if rows > pr:
        raise HipError("step_reduce: rows exceed the partial array")
    for c0, nc, d, d2, sc in segs:
        _req(d, "dst", torch.float32)
        if d.numel() < nc or not d.is_contiguous() or c0 + nc > pc:
            raise HipError("step_reduce: segment does not fit")
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    slabs = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    nslab = (ctypes.c_int32 * n)(*[int(it[1]) for it in items])
    dws = (ctypes.c_void_p * n)(*[it[2].data_ptr() for it in items])
    ldd = (ctypes.c_int64 * n)(*[it[2].stride(0) for it in items])
    Ns = (ctypes.c_int32 * n)(*[it[2].shape[0] for it in items])
    Ks = (ctypes.c_int32 * n)(*[it[2].shape[1] for it in items])
    m = len(segs)
    col0 = (ctypes.c_int32 * m)(*[s[0] for s in segs])
    ncols = (ctypes.c_int32 * m)(*[s[1] for s in segs])
    dst = (ctypes.c_void_p * m)(*[s[2].data_ptr() for s in segs])
    dst2 = (ctypes.c_void_p * m)(*[(s[3].data_ptr() if s[3] is not None else None) for s in segs])
    scale = (ctypes.c_float * m)(*[float(s[4]) for s in segs])
    _check(lib().ib_step_reduce(n, cv(slabs), cv(nslab), cv(dws), cv(ldd), cv(Ns), cv(Ks), _ptr(part), ld, rows, m,
                                cv(col0), cv(ncols), cv(dst), cv(dst2), cv(scale), int(accumulate), stream_ptr()),
           "ib_step_reduce")


def step_reduce_parts(items, segs, accumulate=False):
    """ONE launch that finishes a group of gradients: items = [(slab workspace, nslab, dw fp32 [N, K])] (split-M slabs of
    weight gradients, at most 8), segs = [(part fp32 [rows_total, ld], rows, dst fp32 [ncols])] -- dst = column sums over
    part[:rows, :ncols] (bias / LayerNorm partial sums, each matrix left by a different launch; at most 24)"""
    n = len(items)
    for ws, ns, dw in items:
        _req(ws, "workspace")
        _mat(dw, "dw", torch.float32)
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    slabs = (ctypes.c_void_p * max(n, 1))(*[it[0].data_ptr() for it in items])
    nslab = (ctypes.c_int32 * max(n, 1))(*[int(it[1]) for it in items])
    dws = (ctypes.c_void_p * max(n, 1))(*[it[2].data_ptr() for it in items])
    ldd = (ctypes.c_int64 * max(n, 1))(*[it[2].stride(0) for it in items])
    Ns = (ctypes.c_int32 * max(n, 1))(*[it[2].shape[0] for it in items])
    Ks = (ctypes.c_int32 * max(n, 1))(*[it[2].shape[1] for it in items])
    m = len(segs)
    for part, rows, dst in segs:
        pr, pc, ld = _mat(part, "part", torch.float32)
        _req(dst, "dst", torch.float32)
        if rows > pr or dst.numel() != pc or not dst.is_contiguous():
            raise HipError("step_reduce_parts: segment does not fit its partial matrix")
    parts = (ctypes.c_void_p * max(m, 1))(*[s_[0].data_ptr() for s_ in segs])
    lds = (ctypes.c_int64 * max(m, 1))(*[(s_[0].stride(0) if s_[0].shape[0] > 1 else s_[0].shape[1]) for s_ in segs])
    rows = (ctypes.c_int32 * max(m, 1))(*[int(s_[1]) for s_ in segs])
    col0 = (ctypes.c_int32 * max(m, 1))(*[0 for _ in segs])
    ncols = (ctypes.c_int32 * max(m, 1))(*[s_[0].shape[1] for s_ in segs])
    dst = (ctypes.c_void_p * max(m, 1))(*[s_[2].data_ptr() for s_ in segs])
    _check(lib().ib_step_reduce_parts(n, cv(slabs), cv(nslab), cv(dws), cv(ldd), cv(Ns), cv(Ks), m, cv(parts), cv(lds), cv(rows),
                                      cv(col0), cv(ncols), cv(dst), None, int(accumulate), stream_ptr()),
           "ib_step_reduce_parts")


def segment_colsum(x, out, seg, mode=0, accumulate=False, out_bf16=None):
    dt = x.dtype
    M, N, ldx = _mat(x, "x", dt)
    S, No, ldo = _mat(out, "out", torch.float32)
    nseg = (M + seg - 1) // seg if mode == 0 else min(seg, M)
    if No != N or S < nseg:
        raise HipError(f"segment_colsum: out must be [{nseg}, {N}], got {tuple(out.shape)}")
    ldl = 0
    if out_bf16 is not None:
        Sl, Nl, ldl = _mat(out_bf16, "out_bf16", torch.bfloat16)
        if Nl != N or Sl < nseg:
            raise HipError("segment_colsum: out_bf16 shape mismatch")
    _check(lib().ib_segment_colsum(_ptr(x), ldx, _ptr(out), ldo, _ptr(out_bf16), ldl, M, N, int(seg), int(mode),
                                   int(accumulate), dtype_code(dt), stream_ptr()), "ib_segment_colsum")
    return out


# --------------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------------
def _rows(t: torch.Tensor, name: str, dt):
    _req(t, name, dt)
    if not t.is_contiguous():
        raise HipError(f"{name} must be contiguous")
    return t.numel() // t.shape[-1], t.shape[-1]


def _add_div(add_div, seg, M, N, dt):
    if add_div is None:
        return None, 0, 0
    r, c, ld = _mat(add_div, "add_div", dt)
    if seg <= 0 or c != N or r * seg < M:
        raise HipError(f"add_div must be [ceil(M/seg), N]; got {tuple(add_div.shape)} with seg={seg}, M={M}")
    return add_div.data_ptr(), ld, int(seg)


def layernorm_fwd(x, gamma, beta, y, mean, rstd, res=None, act="none", eps=1e-5, add_div=None, seg=0):
    dt = x.dtype
    M, N = _rows(x, "x", dt)
    for t, n in ((y, "y"),) + (((res, "res"),) if res is not None else ()):
        if _rows(t, n, dt) != (M, N):
            raise HipError(f"{n} shape mismatch")
    for t, n in ((gamma, "gamma"), (beta, "beta")):
        _req(t, n, torch.float32, 1)
        if t.numel() != N or not t.is_contiguous():
            raise HipError(f"{n} must be contiguous fp32 [N]")
    for t, n in ((mean, "mean"), (rstd, "rstd")):
        _req(t, n, torch.float32)
        if t.numel() != M or not t.is_contiguous():
            raise HipError(f"{n} must be contiguous fp32 [M]")
    ap, ald, aseg = _add_div(add_div, seg, M, N, dt)
    _check(lib().ib_layernorm_fwd(_ptr(x), _ptr(res), ACT[act], _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                  _ptr(rstd), ap, ald, aseg, M, N, float(eps), dtype_code(dt), stream_ptr()),
           "ib_layernorm_fwd")
    return y


def layernorm_bwd_workspace_bytes(M, N) -> int:
    return int(lib().ib_layernorm_bwd_workspace(M, N))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, res=None, dres=None, act="none",
                  accumulate=False, add_div=None, seg=0):
    dt = x.dtype
    M, N = _rows(x, "x", dt)
    for t, n in ((dy, "dy"), (dx, "dx")) + (((res, "res"),) if res is not None else ()) + \
            (((dres, "dres"),) if dres is not None else ()):
        if _rows(t, n, dt) != (M, N):
            raise HipError(f"{n} shape mismatch")
    if (dgamma is None) != (dbeta is None):
        raise HipError("layernorm_bwd: dgamma and dbeta are given together or deferred together")
    for t, n in ((gamma, "gamma"),) + (((dgamma, "dgamma"), (dbeta, "dbeta")) if dgamma is not None else ()):
        _req(t, n, torch.float32, 1)
        if t.numel() != N or not t.is_contiguous():
            raise HipError(f"{n} must be contiguous fp32 [N]")
    for t, n in ((mean, "mean"), (rstd, "rstd")):
        _req(t, n, torch.float32)
        if t.numel() != M:
            raise HipError(f"{n} must be fp32 [M]")
    need = layernorm_bwd_workspace_bytes(M, N)
    wsb = workspace.numel() * workspace.element_size()
    if wsb < need:
        raise HipError(f"layernorm_bwd: workspace of {need} bytes required, got {wsb}")
    ap, ald, aseg = _add_div(add_div, seg, M, N, dt)
    _check(lib().ib_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(res), ACT[act], _ptr(gamma), _ptr(mean), _ptr(rstd),
                                  _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), int(accumulate), _ptr(workspace),
                                  wsb, ap, ald, aseg, M, N, dtype_code(dt), stream_ptr()), "ib_layernorm_bwd")
    return dx


def layernorm_bwd_reduce(workspace, dgamma, dbeta, M, N, accumulate=False):
    for t, n in ((dgamma, "dgamma"), (dbeta, "dbeta")):
        _req(t, n, torch.float32, 1)
        if t.numel() != N or not t.is_contiguous():
            raise HipError(f"{n} must be contiguous fp32 [N]")
    wsb = workspace.numel() * workspace.element_size()
    _check(lib().ib_layernorm_bwd_reduce(_ptr(workspace), wsb, _ptr(dgamma), _ptr(dbeta), int(accumulate), int(M), int(N),
                                         stream_ptr()), "ib_layernorm_bwd_reduce")


def tiny_matmul(A, B, C, accumulate=False):
    """C[M,N] (+)= A[M,K] . B[K,N] for problems far below a GEMM tile; A, B, C are 2-D views with ANY strides for A and B
    (transposes, column slices) and a unit column stride for C, each fp32 or bf16"""
    for t, n in ((A, "A"), (B, "B"), (C, "C")):
        _req(t, n, None, 2)
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise HipError(f"tiny_matmul: {n} must be fp32 or bf16")
    M, K = A.shape
    K2, N = B.shape
    if K2 != K or tuple(C.shape) != (M, N) or C.stride(1) != 1:
        raise HipError(f"tiny_matmul: shapes {tuple(A.shape)} x {tuple(B.shape)} -> {tuple(C.shape)} (C rows contiguous)")
    _check(lib().ib_tiny_matmul(_ptr(A), dtype_code(A.dtype), A.stride(0), A.stride(1), _ptr(B), dtype_code(B.dtype),
                                B.stride(0), B.stride(1), _ptr(C), dtype_code(C.dtype), C.stride(0), int(accumulate), M, N, K,
                                stream_ptr()), "ib_tiny_matmul")
    return C


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def _drop_args(drop):
    """drop = (p, seed, step, step_dev) of a train-mode call, or None"""
    if drop is None:
        return None
    p, seed, step, step_dev = drop
    if not 0.0 <= p < 1.0:
        raise HipError(f"dropout probability has to be in [0, 1), but got {p}")
    if step_dev is not None:
        _req(step_dev, "step_dev", torch.int32)
    return float(p), int(seed) & 0xFFFFFFFF, int(step), _ptr(step_dev)


def attention_fwd(qkv, out, lse, num_heads, drop=None):
    """drop = (p, seed, step, step_dev): dropout on the softmax probabilities (nn.MultiheadAttention(dropout=p), train mode)"""
    dt = qkv.dtype
    _req(qkv, "qkv", dt, 3)
    B, T, d3 = qkv.shape
    d = d3 // 3
    if d * 3 != d3 or d % num_heads or not qkv.is_contiguous():
        raise HipError(f"qkv must be contiguous [B,T,3*H*dh], got {tuple(qkv.shape)}")
    _req(out, "out", dt, 3)
    if tuple(out.shape) != (B, T, d) or not out.is_contiguous():
        raise HipError("out must be contiguous [B,T,H*dh]")
    _req(lse, "lse", torch.float32)
    if lse.numel() != B * num_heads * T or not lse.is_contiguous():
        raise HipError("lse must be contiguous fp32 [B,H,T]")
    da = _drop_args(drop)
    if da is not None:
        _check(lib().ib_attention_fwd_drop(_ptr(qkv), _ptr(out), _ptr(lse), B, T, num_heads, d // num_heads, *da,
                                           dtype_code(dt), stream_ptr()), "ib_attention_fwd_drop")
        return out
    _check(lib().ib_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, T, num_heads, d // num_heads, dtype_code(dt),
                                  stream_ptr()), "ib_attention_fwd")
    return out


def attention_drop_mask(B, T, num_heads, drop, device):
    """the multipliers (0 or 1 / (1 - p)) attention_fwd(..., drop=drop) applies, fp32 [B, H, T, T] (tests)"""
    mask = torch.empty(B, num_heads, T, T, dtype=torch.float32, device=device)
    _check(lib().ib_attention_drop_mask(_ptr(mask), B, T, num_heads, *_drop_args(drop), stream_ptr()),
           "ib_attention_drop_mask")
    return mask


def attention_bwd(qkv, out, dout, lse, dqkv, num_heads, drop=None):
    dt = qkv.dtype
    B, T, d3 = qkv.shape
    d = d3 // 3
    for t, n, shp in ((qkv, "qkv", (B, T, d3)), (dqkv, "dqkv", (B, T, d3)), (out, "out", (B, T, d)),
                      (dout, "dout", (B, T, d))):
        _req(t, n, dt, 3)
        if tuple(t.shape) != shp or not t.is_contiguous():
            raise HipError(f"{n} must be contiguous {shp}")
    _req(lse, "lse", torch.float32)
    if lse.numel() != B * num_heads * T:
        raise HipError("lse must be fp32 [B,H,T]")
    da = _drop_args(drop)
    if da is not None:
        _check(lib().ib_attention_bwd_drop(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T, num_heads,
                                           d // num_heads, *da, dtype_code(dt), stream_ptr()), "ib_attention_bwd_drop")
        return dqkv
    _check(lib().ib_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T, num_heads,
                                  d // num_heads, dtype_code(dt), stream_ptr()), "ib_attention_bwd")
    return dqkv


# --------------------------------------------------------------------------------------------
# packing / casts
# --------------------------------------------------------------------------------------------
def concat_keys(tensors: Sequence[torch.Tensor], out: torch.Tensor):
    rows = None
    widths = []
    for i, t in enumerate(tensors):
        _req(t, f"input[{i}]", torch.float32)
        if not t.is_contiguous():
            raise HipError(f"input[{i}] must be contiguous")
        r = t.numel() // t.shape[-1]
        if rows is None:
            rows = r
        elif r != rows:
            raise HipError("all keys must have the same [B,F] leading shape")
        widths.append(t.shape[-1])
    _req(out, "out")
    if out.numel() != rows * sum(widths) or not out.is_contiguous():
        raise HipError("concat_keys: out has the wrong size")
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    ws = (ctypes.c_int32 * n)(*widths)
    _check(lib().ib_concat_keys(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p), n, _ptr(out),
                                rows, dtype_code(out.dtype), stream_ptr()), "ib_concat_keys")
    return out


def cast(src: torch.Tensor, dst: torch.Tensor):
    _req(src, "src")
    _req(dst, "dst")
    if src.numel() != dst.numel() or not src.is_contiguous() or not dst.is_contiguous():
        raise HipError("cast: src/dst must be contiguous with equal numel")
    _check(lib().ib_cast(_ptr(src), dtype_code(src.dtype), _ptr(dst), dtype_code(dst.dtype), src.numel(),
                         stream_ptr()), "ib_cast")
    return dst


def cast2d(src: torch.Tensor, dst: torch.Tensor):
    r, c, lds = _mat(src, "src", src.dtype)
    r2, c2, ldd = _mat(dst, "dst", dst.dtype)
    if (r, c) != (r2, c2):
        raise HipError("cast2d shape mismatch")
    _check(lib().ib_cast2d(_ptr(src), lds, dtype_code(src.dtype), _ptr(dst), ldd, dtype_code(dst.dtype), r, c,
                           stream_ptr()), "ib_cast2d")
    return dst


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
def regression_loss_workspace_bytes(B, F) -> int:
    return int(lib().ib_regression_loss_workspace(B, F))


def _bfc(t: torch.Tensor, name: str, dt, C: int):
    """[B,F,C] tensor with unit component stride -> (batch stride, frame stride).  Dense keys have frame stride C;
    Groundlink's interleaved output (a slice of the last dim of [B,F,30]) has frame stride 30."""
    _req(t, name, dt, 3)
    B, F, c = t.shape
    if c != C or t.stride(2) != 1 or (F > 1 and t.stride(1) < C):
        raise HipError(f"{name}: expected [B,F,{C}] with unit component stride, got {tuple(t.shape)} strides {t.stride()}")
    fs = t.stride(1) if F > 1 else C
    return (t.stride(0) if B > 1 else F * fs), fs


def regression_loss(outs, labels, comp_w, result, workspace, grads=None, threshold=10.0):
    """outs/labels/grads: (cop, force, torque, wrench) tuples."""
    dt = outs[0].dtype
    B, F, _ = outs[0].shape
    Cs = (6, 6, 6, 12)
    og = [_bfc(t, f"out[{i}]", dt, Cs[i]) for i, t in enumerate(outs)]
    obs = (ctypes.c_int64 * 4)(*[g_[0] for g_ in og])
    ofs = (ctypes.c_int64 * 4)(*[g_[1] for g_ in og])
    for i, t in enumerate(outs):
        if tuple(t.shape) != (B, F, Cs[i]):
            raise HipError(f"out[{i}] must be [{B},{F},{Cs[i]}], got {tuple(t.shape)}")
    for i, t in enumerate(labels):
        _req(t, f"label[{i}]", torch.float32, 3)
        if tuple(t.shape) != (B, F, Cs[i]) or not t.is_contiguous():
            raise HipError(f"label[{i}] must be contiguous fp32 [{B},{F},{Cs[i]}], got {tuple(t.shape)}")
    gbs = gfs = None
    gp = [None] * 4
    if grads is not None:
        gg = [_bfc(t, f"grad[{i}]", dt, Cs[i]) for i, t in enumerate(grads)]
        gbs = (ctypes.c_int64 * 4)(*[g_[0] for g_ in gg])
        gfs = (ctypes.c_int64 * 4)(*[g_[1] for g_ in gg])
        gp = [t.data_ptr() for t in grads]
        for i, t in enumerate(grads):
            if tuple(t.shape) != (B, F, Cs[i]):
                raise HipError("grad shape mismatch")
    _req(comp_w, "comp_w", torch.float32, 1)
    _req(result, "result", torch.float32, 1)
    if comp_w.numel() != 30 or result.numel() < 64:
        raise HipError("comp_w must be [30], result [64]")
    wsb = workspace.numel() * workspace.element_size()
    if wsb < regression_loss_workspace_bytes(B, F):
        raise HipError("regression_loss: workspace too small")
    cv = lambda a: None if a is None else ctypes.cast(a, ctypes.c_void_p)
    _check(lib().ib_regression_loss_strided(_ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), cv(obs), cv(ofs),
                                            _ptr(labels[0]), _ptr(labels[1]), _ptr(labels[2]), _ptr(labels[3]),
                                            _ptr(comp_w), float(threshold), _ptr(result), gp[0], gp[1], gp[2], gp[3],
                                            cv(gbs), cv(gfs), _ptr(workspace), wsb, B, F, dtype_code(dt), stream_ptr()),
           "ib_regression_loss_strided")
    return result


def mse_loss_workspace_bytes(n) -> int:
    return int(lib().ib_mse_loss_workspace(n))


def mse_loss(pred, target, result, workspace, dpred=None):
    dt = pred.dtype
    _req(pred, "pred", dt)
    _req(target, "target", dt)
    n = pred.numel()
    if target.numel() != n or not pred.is_contiguous() or not target.is_contiguous():
        raise HipError("mse_loss: pred/target must be contiguous with equal numel")
    if dpred is not None and (dpred.dtype != dt or dpred.numel() != n or not dpred.is_contiguous()):
        raise HipError("mse_loss: dpred mismatch")
    _req(result, "result", torch.float32)
    wsb = workspace.numel() * workspace.element_size()
    if wsb < mse_loss_workspace_bytes(n):
        raise HipError("mse_loss: workspace too small")
    _check(lib().ib_mse_loss(_ptr(pred), _ptr(target), _ptr(dpred), _ptr(result), _ptr(workspace), wsb, n,
                             dtype_code(dt), stream_ptr()), "ib_mse_loss")
    return result


def mse_loss_partial(pred, target, workspace, dpred=None):
    """pred / dpred: contiguous, or row-padded 2-D [rows, cols] views; target: contiguous with rows*cols elements"""
    dt = pred.dtype
    _req(pred, "pred", dt)
    _req(target, "target", dt)
    if not target.is_contiguous():
        raise HipError("mse_loss: target must be contiguous")
    if pred.dim() == 2 and not pred.is_contiguous():
        rows, cols, ldp = _mat(pred, "pred", dt)
    else:
        if not pred.is_contiguous():
            raise HipError("mse_loss: pred must be contiguous or a 2-D row-padded view")
        rows, cols, ldp = 1, pred.numel(), pred.numel()
    if target.numel() != rows * cols:
        raise HipError("mse_loss: pred/target size mismatch")
    ldd = 0
    if dpred is not None:
        _req(dpred, "dpred", dt)
        if dpred.dim() == 2 and not dpred.is_contiguous():
            r2, c2, ldd = _mat(dpred, "dpred", dt)
            if (r2, c2) != (rows, cols):
                raise HipError("mse_loss: dpred shape mismatch")
        else:
            if not dpred.is_contiguous() or dpred.numel() != rows * cols:
                raise HipError("mse_loss: dpred mismatch")
            ldd = cols
    wsb = workspace.numel() * workspace.element_size()
    if wsb < mse_loss_workspace_bytes(rows * cols):
        raise HipError("mse_loss: workspace too small")
    _check(lib().ib_mse_loss_partial(_ptr(pred), ldp, _ptr(target), _ptr(dpred), ldd, _ptr(workspace), wsb, rows, cols,
                                     dtype_code(dt), stream_ptr()), "ib_mse_loss_partial")


def mse_loss_finalize(workspace, result, n):
    _req(result, "result", torch.float32)
    wsb = workspace.numel() * workspace.element_size()
    _check(lib().ib_mse_loss_finalize(_ptr(workspace), wsb, _ptr(result), int(n), stream_ptr()), "ib_mse_loss_finalize")


# --------------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------------
def optim_ticket_words() -> int:
    """int32 words of the optimizer's exit-ticket buffer (self-counting mode)"""
    return int(lib().ib_optim_ticket_words())


def optim_step(opt: str, p, g, s1, s2, lr, step=None, step_dev=None, grad_scale=1.0, shadow=None, ticket=None,
               sources=None):
    """step: the step number of the bias corrections; with `step_dev` the kernel uses `*step_dev + step`, so the default is
    0 there (the device counter alone) and 1 without a device counter.
    sources = (items, part, rows, segs): the gradient of some ranges of g is still partial sums -- items =
    [(slab workspace, nslab, dw view into g)], segs = [(col0, ncols, dst view into g | the loss scalar, dst2, scale)] over
    part[:rows], or 7-tuples (.., part_i, rows_i) that name their own partial matrix (part may then be None).  The optimizer sums them itself (ib_optim_step_sources) instead of a separate ib_step_reduce launch.
    An optional fifth entry lists views into g whose parameters were already updated this step (launches over a slice of
    the buffers with step_dev, no ticket and step=1: same step number as the self-counting launch that follows)."""
    if step is None:
        step = 0 if step_dev is not None else 1
    _req(p, "p", torch.float32, 1)
    _req(g, "g", torch.float32, 1)
    n = p.numel()
    if g.numel() != n or not p.is_contiguous() or not g.is_contiguous():
        raise HipError("optim_step: flat contiguous fp32 buffers of equal length required")
    for t, nm in ((s1, "s1"), (s2, "s2")):
        if t is not None:
            _req(t, nm, torch.float32, 1)
            if t.numel() != n:
                raise HipError(f"{nm} length mismatch")
    if shadow is not None:
        _req(shadow, "shadow", torch.bfloat16, 1)
        if shadow.numel() != n:
            raise HipError("shadow length mismatch")
    if step_dev is not None:
        _req(step_dev, "step_dev", torch.int32)
    if ticket is not None:
        _req(ticket, "ticket", torch.int32)
        if ticket.numel() < optim_ticket_words() or not ticket.is_contiguous():
            raise HipError(f"ticket: {optim_ticket_words()} contiguous zeroed int32 words required (ib_optim_ticket_words)")
    if sources is not None:
        items, part, rows, segs = sources[:4]
        done = sources[4] if len(sources) > 4 else ()       # views into g whose parameters an earlier launch updated
        ld = 0
        if part is not None:
            pr, pc, ld = _mat(part, "part", torch.float32)
        g0, g1 = g.data_ptr(), g.data_ptr() + 4 * n
        ent, loss = [], None                      # (start, len, kind, base, stride, count, scale)

        def flat_off(t, what):
            a = t.data_ptr()
            if not (g0 <= a < g1) or (a - g0) % 16 or not t.is_contiguous():
                raise HipError(f"optim_step sources: {what} must be a contiguous, 16-byte aligned view into g")
            return (a - g0) // 4
        for ws, ns, dw in items:
            ent.append((flat_off(dw, "dw"), dw.numel(), 1, ws.data_ptr(), dw.numel(), int(ns), 1.0))
        for seg in segs:
            c0, nc, d, d2, sc = seg[:5]
            if len(seg) > 5:                      # a segment with its own partial matrix: (.., part_i, rows_i)
                _, _, ld_i = _mat(seg[5], "part_i", torch.float32)
                for dd in (d, d2):
                    if dd is not None:
                        ent.append((flat_off(dd, "dst"), nc, 2, seg[5].data_ptr() + 4 * c0, ld_i, int(seg[6]), float(sc)))
                continue
            if not (g0 <= d.data_ptr() < g1):     # the loss scalar lives outside the flat buffer
                if nc != 1 or loss is not None:
                    raise HipError("optim_step sources: at most one scalar destination outside g")
                loss = (part.data_ptr() + 4 * c0, float(sc), d)
                continue
            for dd in (d, d2):
                if dd is not None:
                    ent.append((flat_off(dd, "dst"), nc, 2, part.data_ptr() + 4 * c0, ld, int(rows), float(sc)))
        for dv in done:
            ent.append((flat_off(dv, "done range"), dv.numel(), 3, 0, 0, 0, 1.0))
        ent.sort(key=lambda e: e[0])              # the kernel's range scan stops at the first source beyond the element
        m = len(ent)
        cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
        A = lambda ct, k: (ct * m)(*[e[k] for e in ent])
        start, ln, kind = A(ctypes.c_int64, 0), A(ctypes.c_int64, 1), A(ctypes.c_int32, 2)
        base, stride, count, scale = A(ctypes.c_void_p, 3), A(ctypes.c_int64, 4), A(ctypes.c_int32, 5), A(ctypes.c_float, 6)
        _check(lib().ib_optim_step_sources(OPT[opt], _ptr(p), _ptr(g), _ptr(s1), _ptr(s2), n, float(lr), float(grad_scale),
                                           int(step), _ptr(step_dev), _ptr(ticket), _ptr(shadow), m, cv(start), cv(ln),
                                           cv(kind), cv(base), cv(stride), cv(count), cv(scale),
                                           ctypes.c_void_p(loss[0]) if loss else None, ld, int(rows),
                                           loss[1] if loss else 0.0, _ptr(loss[2]) if loss else None, stream_ptr()),
               "ib_optim_step_sources")
        return
    _check(lib().ib_optim_step(OPT[opt], _ptr(p), _ptr(g), _ptr(s1), _ptr(s2), n, float(lr), float(grad_scale),
                               int(step), _ptr(step_dev), _ptr(ticket), _ptr(shadow), stream_ptr()), "ib_optim_step")


# --------------------------------------------------------------------------------------------
# diffusion
# --------------------------------------------------------------------------------------------
def gather_rows(table, idx, out):
    _req(table, "table", torch.float32, 2)
    _req(idx, "idx", torch.int64, 1)
    _req(out, "out", None, 2)
    B, dim = out.shape
    if idx.numel() != B or table.shape[1] != dim or not table.is_contiguous() or not out.is_contiguous():
        raise HipError("gather_rows shape mismatch")
    _check(lib().ib_gather_rows(_ptr(table), _ptr(idx), _ptr(out), B, dim, table.shape[0], dtype_code(out.dtype),
                                stream_ptr()), "ib_gather_rows")
    return out


# --------------------------------------------------------------------------------------------
# fused feed-forward sublayer (csrc/ffn_chain.hip)
# --------------------------------------------------------------------------------------------
def ffn_chain_supported(d: int, ffn: int) -> bool:
    return bool(lib().ib_ffn_chain_supported(int(d), int(ffn)))


def ffn_chain_packed_elems(d: int, ffn: int) -> int:
    return int(lib().ib_ffn_chain_packed_elems(int(d), int(ffn)))


def ffn_chain_workgroups(M: int, d: int, ffn: int, T: int = 0) -> int:
    """T > 0: the launches with the attention inside (panels of exactly one window of T frames); 0 = not supported"""
    if T:
        return int(lib().ib_ffn_chain_attn_workgroups(int(M), int(d), int(ffn), int(T)))
    return int(lib().ib_ffn_chain_workgroups(int(M), int(d), int(ffn), None))


def ffn_chain_mask_bytes(M: int, d: int, ffn: int, T: int = 0) -> int:
    if T:
        return int(lib().ib_ffn_chain_attn_mask_bytes(int(M), int(d), int(ffn), int(T)))
    return int(lib().ib_ffn_chain_mask_bytes(int(M), int(d), int(ffn)))


def ffn_chain_pack(layers):
    """layers: [(w1 [ffn, d] bf16, w2 [d, ffn] bf16, packed bf16 [ffn_chain_packed_elems])], optionally with a 4th element,
    the attention out-projection weight [d, d] (the attention epilogue's images), and a 5th, the layer's in-projection
    weight [3 d, d] (the QKV tail / head of the NEIGHBOURING layer's launches) -- ONE launch for all of them"""
    n = len(layers)
    ffn, d = layers[0][0].shape
    for it in layers:
        w1, w2, pk = it[0], it[1], it[2]
        _mat(w1, "w1", torch.bfloat16)
        _mat(w2, "w2", torch.bfloat16)
        _req(pk, "packed", torch.bfloat16, 1)
        if tuple(w1.shape) != (ffn, d) or tuple(w2.shape) != (d, ffn) or pk.numel() < ffn_chain_packed_elems(d, ffn) \
                or not pk.is_contiguous():
            raise HipError("ffn_chain_pack: weight / packed-image shapes do not agree")
        if len(it) > 3 and it[3] is not None:
            _mat(it[3], "wo", torch.bfloat16)
            if tuple(it[3].shape) != (d, d):
                raise HipError("ffn_chain_pack: the out-projection weight must be [d, d]")
        if len(it) > 4 and it[4] is not None:
            _mat(it[4], "wqkv", torch.bfloat16)
            if tuple(it[4].shape) != (3 * d, d):
                raise HipError("ffn_chain_pack: the in-projection weight must be [3 d, d]")
    arr = lambda ts: ctypes.cast((ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts]), ctypes.c_void_p)
    lds = lambda ts: ctypes.cast((ctypes.c_int64 * n)(*[(t.stride(0) if t is not None else 0) for t in ts]), ctypes.c_void_p)
    w1s, w2s, pks = [l[0] for l in layers], [l[1] for l in layers], [l[2] for l in layers]
    wos = [(l[3] if len(l) > 3 else None) for l in layers]
    wqs = [(l[4] if len(l) > 4 else None) for l in layers]
    _check(lib().ib_ffn_chain_pack(arr(w1s), lds(w1s), arr(w2s), lds(w2s), arr(wos), lds(wos), arr(wqs), lds(wqs), arr(pks), n,
                                   d, ffn, stream_ptr()), "ib_ffn_chain_pack")


def _ffn_rows(t, name, M, N, dtype=torch.bfloat16):
    _req(t, name, dtype, 2)
    if tuple(t.shape) != (M, N) or not t.is_contiguous():
        raise HipError(f"{name}: contiguous {dtype} [{M}, {N}] required, got {tuple(t.shape)} strides {t.stride()}")


def _ffn_vec(t, name, n):
    _req(t, name, torch.float32, 1)
    if t.numel() != n or not t.is_contiguous():
        raise HipError(f"{name} must be contiguous fp32 [{n}]")


def ffn_chain_fwd(x1, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask, eps: float = 1e-5, attn_out=None,
                  qkv_next=None, attn_next=None, panel_T: int = 0):
    """panel_T > 0: panels of exactly one window of panel_T frames (`mask` sized with ffn_chain_mask_bytes(.., panel_T)) --
    the geometry of a layer whose attention rides inside its launches, in both directions.
    attn_next = (attn_out_next [M, d], lse_next fp32 [M / T, 8, T], T): the NEXT layer's attention core rides behind the
    QKV tail (needs qkv_next; T = panel_T).
    attn_out = (attn [M, d], bo, gamma1, beta1, s1, x1_out, mean1, rstd1): the attention epilogue -- `x1` is then the layer
    input x and x1_out = LN1(x + attn Wo^T + bo) is computed (and stored) here.
    qkv_next = (packed image of the NEXT layer, its in-projection bias fp32 [3 d], qkv_out [M, 3 d]): that layer's
    in-projection of y rides behind LayerNorm2 (needs attn_out)"""
    M, d = x1.shape
    ffn = f1.shape[1]
    for t, n, w in ((x1, "x1", d), (f1, "f1", ffn), (s2, "s2", d), (y, "y", d)):
        _ffn_rows(t, n, M, w)
    for t, n, w in ((b1, "b1", ffn), (b2, "b2", d), (gamma, "gamma", d), (beta, "beta", d), (mean, "mean", M), (rstd, "rstd", M)):
        _ffn_vec(t, n, w)
    _req(packed, "packed", torch.bfloat16, 1)
    _req(mask, "mask", torch.uint8, 1)
    global _work_note
    if isinstance(_lib, _RecordingLib):
        # algorithmic work of THIS launch form (bench.py's roofline leg): the token-local GEMMs it contains (out-projection
        # d^2, feed-forward 2 d ffn, the next layer's in-projection 3 d^2 per row) + 4 T d per row of attention; bytes = what
        # must cross HBM: rows in (x, attn), rows out (x1, s1, s2, y; f1; qkv, attn_next), the packed images once
        mac = 2 * d * ffn + (d * d if attn_out is not None else 0) + (3 * d * d if qkv_next is not None else 0)
        Tn = int(attn_next[2]) if attn_next is not None else 0
        rows = (6 if attn_out is not None else 3) * d + ffn + (3 * d if qkv_next is not None else 0) + (d if Tn else 0)
        _work_note = (2 * M * mac + 4 * M * Tn * d, 2 * M * rows + 2 * mac,
                      {"M": M, "d": d, "ffn": ffn, "out_proj": attn_out is not None, "qkv_tail": qkv_next is not None,
                       "attention_T": Tn, "panel_T": int(panel_T)})
    T_att = int(attn_next[2]) if attn_next is not None else int(panel_T)
    if panel_T and T_att != int(panel_T):
        raise HipError("ffn_chain_fwd: attn_next's T must equal panel_T")
    if T_att and not ffn_chain_workgroups(M, d, ffn, T_att):
        raise HipError(f"ffn_chain_fwd: no one-window panels for M = {M}, T = {T_att} (16 <= T <= 64, M % T == 0)")
    if packed.numel() < ffn_chain_packed_elems(d, ffn) or mask.numel() < ffn_chain_mask_bytes(M, d, ffn, T_att):
        raise HipError("ffn_chain_fwd: packed image / mask buffer too small")
    extra = [None] * 8
    if attn_out is not None:
        attn, bo, g1, b1n, s1, x1o, m1, r1 = attn_out
        for t, n in ((attn, "attn"), (s1, "s1"), (x1o, "x1_out")):
            _ffn_rows(t, n, M, d)
        for t, n, w in ((bo, "bo", d), (g1, "gamma1", d), (b1n, "beta1", d), (m1, "mean1", M), (r1, "rstd1", M)):
            _ffn_vec(t, n, w)
        extra = [_ptr(t) for t in attn_out]
    tail = [None] * 3
    if qkv_next is not None:
        pk_n, bq, qo = qkv_next
        if attn_out is None:
            raise HipError("ffn_chain_fwd: the QKV tail needs the attention epilogue")
        _req(pk_n, "packed_next", torch.bfloat16, 1)
        _ffn_vec(bq, "bqkv_next", 3 * d)
        _ffn_rows(qo, "qkv_next", M, 3 * d)
        if pk_n.numel() < ffn_chain_packed_elems(d, ffn):
            raise HipError("ffn_chain_fwd: packed image of the next layer too small")
        tail = [_ptr(pk_n), _ptr(bq), _ptr(qo)]
    if T_att:
        ao = lse = None
        if attn_next is not None:
            ao, lse, _ = attn_next
            if qkv_next is None:
                raise HipError("ffn_chain_fwd: the attention tail needs the QKV tail")
            _ffn_rows(ao, "attn_next", M, d)
            _req(lse, "lse_next", torch.float32)
            if lse.numel() != (M // T_att) * 8 * T_att or not lse.is_contiguous() or d != 512:
                raise HipError("ffn_chain_fwd: lse_next must be contiguous fp32 [M / T, 8, T] (d = 512: eight heads of 64)")
        _check(lib().ib_ffn_chain_fwd_attn(_ptr(x1), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(beta), _ptr(f1),
                                           _ptr(s2), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(mask), *extra, *tail, _ptr(ao),
                                           _ptr(lse), T_att, M, d, ffn, float(eps), stream_ptr()), "ib_ffn_chain_fwd_attn")
        return y
    _check(lib().ib_ffn_chain_fwd(_ptr(x1), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(beta), _ptr(f1), _ptr(s2),
                                  _ptr(y), _ptr(mean), _ptr(rstd), _ptr(mask), *extra, *tail, M, d, ffn, float(eps),
                                  stream_ptr()), "ib_ffn_chain_fwd")
    return y


def ffn_chain_fwd_infer(x, packed, b1, b2, gamma, beta, y, attn, bo, gamma1, beta1, qkv_next=None, eps: float = 1e-5):
    """frozen-weight forward of a layer's token-local half in one launch (csrc/ffn_chain.hip, INFER form): y = LN2(x1 + FFN(x1)),
    x1 = LN1(x + attn Wo^T + bo); qkv_next = (packed image of the NEXT layer, its in-projection bias, qkv_out [M, 3 d]).
    Nothing is saved.  False: shape not supported."""
    M, d = x.shape
    ffn = b1.numel()
    if not ffn_chain_supported(d, ffn):
        return False
    for t, n in ((x, "x"), (y, "y"), (attn, "attn")):
        _ffn_rows(t, n, M, d)
    for t, n, w in ((b1, "b1", ffn), (b2, "b2", d), (gamma, "gamma", d), (beta, "beta", d), (bo, "bo", d), (gamma1, "gamma1", d),
                    (beta1, "beta1", d)):
        _ffn_vec(t, n, w)
    _req(packed, "packed", torch.bfloat16, 1)
    if packed.numel() < ffn_chain_packed_elems(d, ffn):
        raise HipError("ffn_chain_fwd_infer: packed image too small")
    tail = [None] * 3
    if qkv_next is not None:
        pk_n, bq, qo = qkv_next
        _req(pk_n, "packed_next", torch.bfloat16, 1)
        _ffn_vec(bq, "bqkv_next", 3 * d)
        _ffn_rows(qo, "qkv_next", M, 3 * d)
        if pk_n.numel() < ffn_chain_packed_elems(d, ffn):
            raise HipError("ffn_chain_fwd_infer: packed image of the next layer too small")
        tail = [_ptr(pk_n), _ptr(bq), _ptr(qo)]
    global _work_note
    if isinstance(_lib, _RecordingLib):
        mac = 2 * d * ffn + d * d + (3 * d * d if qkv_next is not None else 0)
        _work_note = (2 * M * mac, 2 * M * (3 * d + (3 * d if qkv_next is not None else 0)) + 2 * mac,
                      {"M": M, "d": d, "ffn": ffn, "qkv_tail": qkv_next is not None, "frozen": True})
    _check(lib().ib_ffn_chain_fwd_infer(_ptr(x), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(attn),
                                        _ptr(bo), _ptr(gamma1), _ptr(beta1), *tail, M, d, ffn, float(eps), stream_ptr()),
           "ib_ffn_chain_fwd_infer")
    return True


def ffn_chain_bwd(dy, s2, mean, rstd, gamma, packed, mask, ds2, dz1, dx1, partial, attn_out=None, qkv_head=None,
                  attn_bwd=None):
    """attn_bwd = (qkv [M, 3 d], lse fp32 [M / T, 8, T], dqkv [M, 3 d], dx [M, d], T): the launch continues through this
    layer's attention backward and in-projection dgrad (needs attn_out, whose dattn entry may be None: it is not stored;
    no qkv_head; mask / partial sized for the one-window panels: ffn_chain_workgroups(.., T)).
    attn_out = (s1, mean1, rstd1, gamma1, ds1, dattn): LayerNorm1 backward + the out-projection's dgrad in the same launch
    (dx1 may then be None: it is not stored); partial: fp32 [2 x workgroups, d], or [4 x workgroups, d] with attn_out.
    qkv_head = (packed image of the NEXT layer, dqkv_next [M, 3 d], ds1_next [M, d]): dy is computed in front of
    LayerNorm2's backward = dqkv_next . Wqkv_next + ds1_next (dy may then be None; needs attn_out)"""
    M, d = s2.shape
    ffn = dz1.shape[1]
    for t, n, w in ((s2, "s2", d), (ds2, "ds2", d), (dz1, "dz1", ffn)) + (((dx1, "dx1", d),) if dx1 is not None else ()) \
            + (((dy, "dy", d),) if dy is not None else ()):
        _ffn_rows(t, n, M, w)
    _req(partial, "partial", torch.float32, 2)
    nq = 4 if attn_out is not None else 2
    T_att = int(attn_bwd[4]) if attn_bwd is not None else 0
    global _work_note
    if isinstance(_lib, _RecordingLib):
        # dgrad GEMMs of this launch form: feed-forward 2 d ffn, out-projection d^2, in-projection 3 d^2 (own, behind the
        # attention backward, or the next layer's in front) per row + 10 T d per row of attention backward (recompute of
        # S and dP twice, dQ, dK, dV); bytes: rows in (dy | dqkv + ds1, s2, s1; qkv), rows out (ds2, ds1; dz1; dqkv, dx | dattn)
        mac = 2 * d * ffn + (d * d if attn_out is not None else 0) + (3 * d * d if (qkv_head is not None or T_att) else 0)
        rows = (5 if attn_out is not None else 4) * d + ffn + (3 * d if qkv_head is not None else 0) + ((7 * d) if T_att else (d if attn_out is not None else 0))
        _work_note = (2 * M * mac + 10 * M * T_att * d, 2 * M * rows + 2 * mac,
                      {"M": M, "d": d, "ffn": ffn, "out_proj": attn_out is not None, "qkv_head": qkv_head is not None,
                       "attention_T": T_att})
    if T_att and not ffn_chain_workgroups(M, d, ffn, T_att):
        raise HipError(f"ffn_chain_bwd: no one-window panels for M = {M}, T = {T_att} (16 <= T <= 64, M % T == 0)")
    if tuple(partial.shape) != (nq * ffn_chain_workgroups(M, d, ffn, T_att), d) or not partial.is_contiguous():
        raise HipError(f"ffn_chain_bwd: partial must be contiguous fp32 [{nq} x workgroups, d]")
    for t, n, w in ((mean, "mean", M), (rstd, "rstd", M), (gamma, "gamma", d)):
        _ffn_vec(t, n, w)
    _req(mask, "mask", torch.uint8, 1)
    if packed.numel() < ffn_chain_packed_elems(d, ffn) or mask.numel() < ffn_chain_mask_bytes(M, d, ffn, T_att):
        raise HipError("ffn_chain_bwd: packed image / mask buffer too small")
    extra = [None] * 6
    if attn_out is not None:
        s1, m1, r1, g1, ds1, dattn = attn_out
        for t, n in ((s1, "s1"), (ds1, "ds1")) + (((dattn, "dattn"),) if not (T_att and dattn is None) else ()):
            _ffn_rows(t, n, M, d)
        for t, n, w in ((m1, "mean1", M), (r1, "rstd1", M), (g1, "gamma1", d)):
            _ffn_vec(t, n, w)
        extra = [_ptr(t) for t in attn_out]
    elif dx1 is None:
        raise HipError("ffn_chain_bwd: dx1 is required without the attention epilogue")
    head = [None] * 3
    if qkv_head is not None:
        pk_n, dq, ds1n = qkv_head
        if attn_out is None:
            raise HipError("ffn_chain_bwd: the QKV head needs the attention epilogue")
        _req(pk_n, "packed_next", torch.bfloat16, 1)
        _ffn_rows(dq, "dqkv_next", M, 3 * d)
        _ffn_rows(ds1n, "ds1_next", M, d)
        if pk_n.numel() < ffn_chain_packed_elems(d, ffn):
            raise HipError("ffn_chain_bwd: packed image of the next layer too small")
        head = [_ptr(pk_n), _ptr(dq), _ptr(ds1n)]
    elif dy is None:
        raise HipError("ffn_chain_bwd: dy is required without the QKV head")
    if attn_bwd is not None:
        qkv, lse, dqkv, dx, _ = attn_bwd
        if attn_out is None or qkv_head is not None or d != 512:
            raise HipError("ffn_chain_bwd: the attention tail needs the attention epilogue, no QKV head and d = 512")
        _ffn_rows(qkv, "qkv", M, 3 * d)
        _ffn_rows(dqkv, "dqkv", M, 3 * d)
        _ffn_rows(dx, "dx", M, d)
        _req(lse, "lse", torch.float32)
        if lse.numel() != (M // T_att) * 8 * T_att or not lse.is_contiguous():
            raise HipError("ffn_chain_bwd: lse must be contiguous fp32 [M / T, 8, T]")
        _check(lib().ib_ffn_chain_bwd_attn(_ptr(dy), _ptr(s2), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(packed), _ptr(mask),
                                           _ptr(ds2), _ptr(dz1), _ptr(partial), *extra[:5], _ptr(qkv), _ptr(lse), _ptr(dqkv),
                                           _ptr(dx), T_att, M, d, ffn, stream_ptr()), "ib_ffn_chain_bwd_attn")
        return dx
    _check(lib().ib_ffn_chain_bwd(_ptr(dy), _ptr(s2), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(packed), _ptr(mask), _ptr(ds2),
                                  _ptr(dz1), _ptr(dx1), _ptr(partial), *extra, *head, M, d, ffn, stream_ptr()),
           "ib_ffn_chain_bwd")
    return dx1


def gather_rows_bwd(dout, idx, dtable):
    """dtable[r] = sum of dout[i] over idx[i] == r (fp32, position order)"""
    _req(dout, "dout", torch.float32, 2)
    _req(idx, "idx", torch.int64, 1)
    _req(dtable, "dtable", torch.float32, 2)
    n, dim = dout.shape
    if idx.numel() != n or dtable.shape[1] != dim or not dout.is_contiguous() or not dtable.is_contiguous():
        raise HipError("gather_rows_bwd shape mismatch")
    _check(lib().ib_gather_rows_bwd(_ptr(dout), _ptr(idx), _ptr(dtable), n, dim, dtable.shape[0], stream_ptr()),
           "ib_gather_rows_bwd")
    return dtable


def _pair3(o, l, what):
    _req(o, what + " output")
    _req(l, what + " label")
    if o.dtype != l.dtype or o.shape != l.shape or o.dim() != 3 or not o.is_contiguous() or not l.is_contiguous():
        raise HipError(f"{what}: contiguous [B, F, C] tensors of one shape and dtype required")
    return o.shape


def sqdiff_mean(o, l):
    B, F, C = _pair3(o, l, "sqdiff_mean")
    out = torch.empty(C, dtype=torch.float32, device=o.device)
    _check(lib().ib_sqdiff_mean(_ptr(o), _ptr(l), _ptr(out), B * F, C, dtype_code(o.dtype), stream_ptr()), "ib_sqdiff_mean")
    return out


def sqdiff_mean_bwd(o, l, dout):
    B, F, C = _pair3(o, l, "sqdiff_mean_bwd")
    _req(dout, "dout", torch.float32, 1)
    d_o = torch.empty_like(o)
    _check(lib().ib_sqdiff_mean_bwd(_ptr(o), _ptr(l), _ptr(dout), _ptr(d_o), B * F, C, dtype_code(o.dtype), stream_ptr()),
           "ib_sqdiff_mean_bwd")
    return d_o


def mask_by_threes(t, threshold: float):
    _req(t, "tensor")
    if not t.is_contiguous() or t.numel() % 3:
        raise HipError("mask_by_threes: contiguous tensor with a multiple of 3 values required")
    mask = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    _check(lib().ib_mask_by_threes(_ptr(t), _ptr(mask), t.numel(), float(threshold), dtype_code(t.dtype), stream_ptr()),
           "ib_mask_by_threes")
    return mask


def mean_norm_error(o, l, vec_size: int = 3, fold_halves: bool = False):
    B, F, C = _pair3(o, l, "mean_norm_error")
    out = torch.empty(1, dtype=torch.float32, device=o.device)
    _check(lib().ib_mean_norm_error(_ptr(o), _ptr(l), _ptr(out), B, F, C, int(vec_size), int(bool(fold_halves)),
                                    dtype_code(o.dtype), stream_ptr()), "ib_mean_norm_error")
    return out[0]


def im2col_replicate(x, col, N: int, F: int, k: int):
    """x: contiguous [N*F, C]; col: [N*F, >= C*k] (row pitch may be padded; pad columns are zero-filled)"""
    dt = x.dtype
    M, C, ldx = _mat(x, "x", dt)
    Mc, Kc, ldc = _mat(col, "col", dt)
    if M != N * F or Mc != M or Kc < C * k or not x.is_contiguous():
        raise HipError("im2col_replicate: shapes")
    _check(lib().ib_im2col_replicate(_ptr(x), _ptr(col), ldc, N, F, C, int(k), dtype_code(dt), stream_ptr()),
           "ib_im2col_replicate")
    return col


def col2im_replicate(dcol, dx, N: int, F: int, k: int, act="none", aux=None):
    dt = dcol.dtype
    M, Kc, ldc = _mat(dcol, "dcol", dt)
    Mx, C, _ = _mat(dx, "dx", dt)
    if M != N * F or Mx != M or Kc < C * k or not dx.is_contiguous():
        raise HipError("col2im_replicate: shapes")
    if aux is not None:
        _req(aux, "aux", dt)
        if aux.shape != dx.shape or not aux.is_contiguous():
            raise HipError("col2im_replicate: aux must match dx")
    _check(lib().ib_col2im_replicate(_ptr(dcol), ldc, _ptr(aux), ACT[act], _ptr(dx), N, F, C, int(k), dtype_code(dt),
                                     stream_ptr()), "ib_col2im_replicate")
    return dx


def dropout(x, y, p: float, seed: int, step: int = 0, step_dev=None):
    """y = x * mask / (1 - p); the same (seed, step) reproduces the mask (apply to the gradient in the backward)"""
    _req(x, "x"); _req(y, "y", x.dtype)
    if x.numel() != y.numel() or not x.is_contiguous() or not y.is_contiguous():
        raise HipError("dropout: contiguous tensors of equal size required")
    if step_dev is not None:
        _req(step_dev, "step_dev", torch.int32)
    _check(lib().ib_dropout(_ptr(x), _ptr(y), x.numel(), float(p), int(seed) & 0xFFFFFFFF, int(step), _ptr(step_dev),
                            dtype_code(x.dtype), stream_ptr()), "ib_dropout")
    return y


def gather_windows(table, idx, x_out, labels):
    """table: packed fp32 [rows, row_elems] window cache; idx: int64 [B]; x_out: [B, x_elems] (fp32 / bf16);
    labels: 4 contiguous fp32 tensors [B, F, c] (cop, force, torque, wrench) -- one launch"""
    rows, row_elems, _ = _mat(table, "table", torch.float32)
    if not table.is_contiguous():
        raise HipError("gather_windows: table must be contiguous")
    _req(idx, "idx", torch.int64, 1)
    B = idx.numel()
    _req(x_out, "x_out", None, 2)
    if x_out.shape[0] != B or not x_out.is_contiguous():
        raise HipError("gather_windows: x_out must be contiguous [B, x_elems]")
    le = []
    for t in labels:
        _req(t, "label", torch.float32)
        if t.shape[0] != B or not t.is_contiguous():
            raise HipError("gather_windows: labels must be contiguous fp32 [B, ...]")
        le.append(t.numel() // B)
    if len(labels) != 4 or (x_out.shape[1] + 3) // 4 * 4 + sum((e + 3) // 4 * 4 for e in le) != row_elems:
        raise HipError("gather_windows: x_out + labels do not add up to the packed row")
    lp = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in labels])
    ln = (ctypes.c_int64 * 4)(*le)
    _check(lib().ib_gather_windows(_ptr(table), row_elems, rows, _ptr(idx), B, _ptr(x_out), x_out.shape[1],
                                   dtype_code(x_out.dtype), ctypes.cast(lp, ctypes.c_void_p),
                                   ctypes.cast(ln, ctypes.c_void_p), stream_ptr()), "ib_gather_windows")


def diffusion_draw(seed: int, step: int = 0, step_dev=None, stream_id: int = 0, eps=None, t=None, num_train_steps: int = 0,
                   table=None, idx=None, x0=None):
    """One launch that makes a diffusion batch on the device (csrc/noise.hip): `x0[b] = table[idx[b]]` (optional),
    `t[b] ~ U{0..num_train_steps-1}` (optional), `eps ~ N(0,1)` (optional) from the counter-based Philox stream keyed by
    (seed, step + *step_dev, stream_id).  eps / x0: contiguous [B, ...] in fp32 or bf16; table: [rows, pitch] with
    pitch % 8 == 0; t: int64 [B]."""
    ref = eps if eps is not None else x0
    if ref is None:
        if t is None:
            raise HipError("diffusion_draw: nothing to draw")
        B, per, code = t.numel(), 1, F32
    else:
        _req(ref, "eps/x0")
        if not ref.is_contiguous() or ref.dim() < 2:
            raise HipError("diffusion_draw: eps / x0 must be contiguous [B, ...]")
        B, per, code = ref.shape[0], ref.numel() // ref.shape[0], dtype_code(ref.dtype)
    if eps is not None and x0 is not None and (eps.shape != x0.shape or eps.dtype != x0.dtype or not x0.is_contiguous()):
        raise HipError("diffusion_draw: x0 and eps must agree in shape / dtype")
    rows = pitch = 0
    if table is not None:
        rows, pitch, _ = _mat(table, "table", x0.dtype if x0 is not None else None)
        if x0 is None or idx is None or not table.is_contiguous() or pitch % 8 or pitch < per:
            raise HipError("diffusion_draw: table needs x0 + idx, contiguous rows with pitch % 8 == 0 and pitch >= T * D")
        _req(idx, "idx", torch.int64, 1)
        if idx.numel() != B:
            raise HipError("diffusion_draw: one index per window")
    if t is not None:
        _req(t, "t", torch.int64, 1)
        if t.numel() != B or num_train_steps <= 0 or not t.is_contiguous():
            raise HipError("diffusion_draw: t must be contiguous int64 [B] and num_train_steps > 0")
    if step_dev is not None:
        _req(step_dev, "step_dev", torch.int32)
    _check(lib().ib_diffusion_draw(_ptr(table), rows, pitch, _ptr(idx), _ptr(x0), _ptr(eps), _ptr(t), B, per,
                                   int(num_train_steps), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step), _ptr(step_dev),
                                   int(stream_id) & 0xFFFFFFFF, code, stream_ptr()), "ib_diffusion_draw")


def philox_words(out, seed: int, step: int, stream_id: int, domain: int):
    """out: int32 / uint32-viewed tensor of 4 * blocks words (16-byte aligned): the raw Philox4x32-10 stream"""
    _req(out, "out", torch.int32, 1)
    if out.numel() % 4 or not out.is_contiguous():
        raise HipError("philox_words: out must hold whole 4-word blocks")
    _check(lib().ib_philox_words(_ptr(out), out.numel() // 4, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFF,
                                 int(stream_id) & 0xFFFFFFFF, int(domain), stream_ptr()), "ib_philox_words")


def q_sample(x0, eps, t, sqrt_ab, sqrt_1mab, x_t):
    """x0 / eps: contiguous [B,T,D]; x_t: contiguous [B,T,D] or a row-padded 2-D [B*T, D] view"""
    dt = x0.dtype
    for a, n in ((x0, "x0"), (eps, "eps")):
        _req(a, n, dt, 3)
        if a.shape != x0.shape or not a.is_contiguous():
            raise HipError(f"{n}: contiguous [B,T,D] tensors of one shape required")
    B, T, D = x0.shape
    _req(x_t, "x_t", dt)
    if x_t.dim() == 3:
        if x_t.shape != x0.shape or not x_t.is_contiguous():
            raise HipError("x_t: contiguous [B,T,D] or 2-D [B*T, D] required")
        ld = D
    else:
        r, c, ld = _mat(x_t, "x_t", dt)
        if (r, c) != (B * T, D):
            raise HipError("x_t shape mismatch")
    _req(t, "t", torch.int64, 1)
    if t.numel() != B:
        raise HipError("t must be int64 [B]")
    _req(sqrt_ab, "sqrt_ab", torch.float32, 1)
    _req(sqrt_1mab, "sqrt_1mab", torch.float32, 1)
    _check(lib().ib_q_sample(_ptr(x0), _ptr(eps), _ptr(t), _ptr(sqrt_ab), _ptr(sqrt_1mab), _ptr(x_t), ld, B, T, D,
                             sqrt_ab.numel(), dtype_code(dt), stream_ptr()), "ib_q_sample")
    return x_t


def time_mlp_fwd_supported(temb: int, hidden: int, out: int) -> bool:
    return bool(lib().ib_time_mlp_fwd_supported(temb, hidden, out))


def time_mlp_fwd(table, t, w1, b1, w2, b2, s, zu, u, e, pack=None, slots=None):
    """fused time-embedding MLP forward (bf16 weights as stored); fills s, zu, u (backward operands) and e.
    pack=(weights, packed, D, H): the same launch also packs the chain kernel's weights (ib_mlp_chain_prep)"""
    bt = torch.bfloat16
    rows, temb, _ = _mat(table, "table", torch.float32)
    hid, k1, ldw1 = _mat(w1, "w1", bt)
    out, k2, ldw2 = _mat(w2, "w2", bt)
    _req(t, "t", torch.int64, 1)
    B = t.numel()
    if k1 != temb or k2 != hid or not table.is_contiguous():
        raise HipError("time_mlp_fwd: weight / table shapes do not chain")
    for name, a, shape in (("s", s, (B, temb)), ("zu", zu, (B, hid)), ("u", u, (B, hid))):
        _req(a, name, bt)
        if tuple(a.shape) != shape or not a.is_contiguous():
            raise HipError(f"time_mlp_fwd: {name} must be contiguous {shape}")
    er, ec, lde = _mat(e, "e", bt)
    if (er, ec) != (B, out):
        raise HipError("time_mlp_fwd: e must be [B, out]")
    _req(b1, "b1", torch.float32, 1); _req(b2, "b2", torch.float32, 1)
    if b1.numel() != hid or b2.numel() != out:
        raise HipError("time_mlp_fwd: bias sizes")
    if pack is not None:
        weights, packed, D, H = pack
        L = len(weights) - 1
        for w in weights:
            _mat(w, "chain weight", bt)
        _req(packed, "packed", bt, 1)
        if packed.numel() < mlp_chain_packed_elems(D, H, L):
            raise HipError("mlp_chain_prep: packed buffer too small")
        keep, wp = _ptr_array(weights)
        ld = (ctypes.c_int64 * len(weights))(*[w.stride(0) for w in weights])
        _check(lib().ib_mlp_chain_prep(_ptr(table), rows, _ptr(t), _ptr(w1), ldw1, _ptr(b1), _ptr(w2), ldw2, _ptr(b2),
                                       _ptr(s), _ptr(zu), _ptr(u), _ptr(e), lde, B, temb, hid, out, wp,
                                       ctypes.cast(ld, ctypes.c_void_p), _ptr(packed), D, H, L, _ptr(slots), stream_ptr()),
               "ib_mlp_chain_prep")
        return e
    _check(lib().ib_time_mlp_fwd(_ptr(table), rows, _ptr(t), _ptr(w1), ldw1, _ptr(b1), _ptr(w2), ldw2, _ptr(b2),
                                 _ptr(s), _ptr(zu), _ptr(u), _ptr(e), lde, B, temb, hid, out, stream_ptr()),
           "ib_time_mlp_fwd")
    return e


def time_mlp_bwd_supported(temb: int, hidden: int, out: int) -> bool:
    return bool(lib().ib_time_mlp_bwd_supported(temb, hidden, out))


def time_mlp_bwd_slab_count(B: int) -> int:
    return int(lib().ib_time_mlp_bwd_slab_count(B))


def time_mlp_bwd(de, w2, zu, s, dw1_slabs, db1_slabs) -> int:
    """backward of the time-embedding MLP's hidden layer in one launch (bf16): de [B, out], w2 [out, hidden], zu [B, hidden],
    s [B, temb] -> fp32 partial slabs dw1_slabs [slabs, hidden, temb], db1_slabs [slabs, hidden] (slabs =
    time_mlp_bwd_slab_count(B); summed in order by optim_step(sources=...) / step_reduce).  Returns the slab count."""
    bt = torch.bfloat16
    B, out, ld_de = _mat(de, "de", bt)
    o2, hid, ldw2 = _mat(w2, "w2", bt)
    Bz, hz, ldzu = _mat(zu, "zu", bt)
    Bs, temb, lds = _mat(s, "s", bt)
    if o2 != out or hz != hid or Bz != B or Bs != B:
        raise HipError("time_mlp_bwd: operand shapes do not chain")
    n = time_mlp_bwd_slab_count(B)
    _req(dw1_slabs, "dw1_slabs", torch.float32); _req(db1_slabs, "db1_slabs", torch.float32)
    if tuple(dw1_slabs.shape) != (n, hid, temb) or tuple(db1_slabs.shape) != (n, hid) or not dw1_slabs.is_contiguous() \
            or not db1_slabs.is_contiguous():
        raise HipError(f"time_mlp_bwd: slabs must be contiguous [{n}, {hid}, {temb}] and [{n}, {hid}]")
    _check(lib().ib_time_mlp_bwd(_ptr(de), ld_de, _ptr(w2), ldw2, _ptr(zu), ldzu, _ptr(s), lds, _ptr(dw1_slabs),
                                 _ptr(db1_slabs), B, temb, hid, out, stream_ptr()), "ib_time_mlp_bwd")
    return n


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


def mlp_chain_supported(D: int, H: int, L: int) -> bool:
    return bool(lib().ib_mlp_chain_supported(D, H, L))


def mlp_chain_workgroups(M: int) -> int:
    return int(lib().ib_mlp_chain_workgroups(M, None))


def mlp_chain_packed_elems(D: int, H: int, L: int) -> int:
    return int(lib().ib_mlp_chain_packed_elems(D, H, L))


def mlp_chain_pack(weights, packed, D: int, H: int):
    """weights: bf16 [H,D], [H,H] x (L-1), [D,H] (blocks then head; row-major, last dim contiguous)"""
    L = len(weights) - 1
    for w in weights:
        _mat(w, "chain weight", torch.bfloat16)
    _req(packed, "packed", torch.bfloat16, 1)
    if packed.numel() < mlp_chain_packed_elems(D, H, L):
        raise HipError("mlp_chain_pack: packed buffer too small")
    keep, wp = _ptr_array(weights)
    ld = (ctypes.c_int64 * len(weights))(*[w.stride(0) for w in weights])
    _check(lib().ib_mlp_chain_pack(wp, ctypes.cast(ld, ctypes.c_void_p), _ptr(packed), D, H, L, stream_ptr()),
           "ib_mlp_chain_pack")
    return packed


def mlp_chain_partial_width(D: int, H: int, L: int) -> int:
    return int(lib().ib_mlp_chain_partial_width(D, H, L))


def mlp_chain_rows_per_workgroup(M: int) -> int:
    p = ctypes.c_int(0)
    lib().ib_mlp_chain_workgroups(M, ctypes.cast(ctypes.pointer(p), ctypes.c_void_p))
    return p.value


def set_ptrs(slots, tensors):
    """slots: int64 device tensor (>= len(tensors)); writes the tensors' addresses into it on the current stream"""
    _req(slots, "slots", torch.int64, 1)
    n = len(tensors)
    if slots.numel() < n or n > 4:
        raise HipError("set_ptrs: at most 4 pointers, slots too small")
    arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    _check(lib().ib_set_ptrs(_ptr(slots), n, ctypes.cast(arr, ctypes.c_void_p), stream_ptr()), "ib_set_ptrs")


def mlp_chain_train(x0, eps, t, sqrt_ab, sqrt_1mab, e, packed, bias, gamma, beta, xt, u, h, dz, dpred, partial,
                    T: int, de_lp=None, ln_eps: float = 1e-5, slots=None):
    """x0 / eps: contiguous bf16 [B,T,D]; e: bf16 [B, L*H]; xt / dpred: bf16 2-D [B*T, D] (row pitch % 4 == 0);
    h / dz: L contiguous bf16 [B*T, H]; u: the same, or None / None entries -- the kernel keeps the pre-activations in
    registers and stores them only into the buffers it is given (L <= 2; deeper stacks need every u buffer);
    bias: L+1 fp32 vectors; gamma / beta: L fp32 vectors;
    partial: fp32 [workgroups, >= mlp_chain_partial_width] (column layout: include/ib_hip.h);
    de_lp: optional bf16 [B, L*H], legal only when mlp_chain_rows_per_workgroup(B*T) == T"""
    bt = torch.bfloat16
    _req(x0, "x0", bt, 3); _req(eps, "eps", bt, 3)
    if x0.shape != eps.shape or not x0.is_contiguous() or not eps.is_contiguous():
        raise HipError("mlp_chain_train: x0 / eps must be contiguous and of one shape")
    B, T_, D = x0.shape
    if T_ != T:
        raise HipError("mlp_chain_train: T mismatch")
    M, L = B * T, len(h)
    H = h[0].shape[1]
    u = [None] * L if u is None else list(u)
    from ._tuning import tuning as TU
    if any(a is None for a in u) and (L > 2 or TU.chain_v1):
        raise HipError("mlp_chain_train: u buffers are optional only for L <= 2 (pre-activations kept in registers)")
    _req(t, "t", torch.int64, 1); _req(sqrt_ab, "sqrt_ab", torch.float32, 1); _req(sqrt_1mab, "sqrt_1mab", torch.float32, 1)
    er, ec, lde = _mat(e, "e", bt)
    if (er, ec) != (B, L * H) or t.numel() != B:
        raise HipError("mlp_chain_train: e must be [B, L*H], t [B]")
    for name, a in (("xt", xt), ("dpred", dpred)):
        r, c, _ = _mat(a, name, bt)
        if (r, c) != (M, D):
            raise HipError(f"mlp_chain_train: {name} must be [B*T, D]")
    for name, lst, shape, dt in (("u", u, (M, H), bt), ("h", h, (M, H), bt), ("dz", dz, (M, H), bt),
                                 ("gamma", gamma, (H,), torch.float32), ("beta", beta, (H,), torch.float32)):
        if len(lst) != L:
            raise HipError(f"mlp_chain_train: {name} needs {L} entries")
        for a in lst:
            if a is None and name == "u":
                continue
            _req(a, name, dt)
            if tuple(a.shape) != shape or not a.is_contiguous():
                raise HipError(f"mlp_chain_train: {name} must be contiguous {shape}")
    if len(bias) != L + 1:
        raise HipError("mlp_chain_train: bias needs L+1 entries")
    for i, a in enumerate(bias):
        _req(a, "bias", torch.float32, 1)
        if a.numel() != (H if i < L else D):
            raise HipError("mlp_chain_train: bias size mismatch")
    nwg = mlp_chain_workgroups(M)
    pr, pc, ldp = _mat(partial, "partial", torch.float32)
    if pr < nwg or pc < mlp_chain_partial_width(D, H, L):
        raise HipError("mlp_chain_train: partial must be [workgroups, >= partial_width]")
    ldd = 0
    if de_lp is not None:
        dr, dc, ldd = _mat(de_lp, "de_lp", bt)
        if dr < B or dc < L * H or mlp_chain_rows_per_workgroup(M) != T:
            raise HipError("mlp_chain_train: de_lp needs [B, L*H] and panels of exactly one window")
    _req(packed, "packed", bt, 1)
    if packed.numel() < mlp_chain_packed_elems(D, H, L):
        raise HipError("mlp_chain_train: packed too small")
    k1, pb = _ptr_array(bias); k2, pg = _ptr_array(gamma); k3, pbe = _ptr_array(beta)
    k4, pu = _ptr_array(u); k5, ph = _ptr_array(h); k6, pdz = _ptr_array(dz)
    _check(lib().ib_mlp_chain_train(_ptr(x0), _ptr(eps), _ptr(t), _ptr(sqrt_ab), _ptr(sqrt_1mab), sqrt_ab.numel(),
                                    _ptr(e), lde, _ptr(packed), pb, pg, pbe, _ptr(xt), xt.stride(0), pu, ph, pdz,
                                    _ptr(dpred), dpred.stride(0), _ptr(partial), ldp, _ptr(de_lp), ldd, _ptr(slots), M, T, D, H,
                                    L, float(ln_eps), stream_ptr()), "ib_mlp_chain_train")
    return nwg


def colsum_segments(part, rows: int, segs, accumulate=False):
    """segs: [(col0, ncols, dst, dst2_or_None, scale)]; dst[c] (+)= scale * sum_{r < rows} part[r, col0 + c]"""
    pr, pc, ld = _mat(part, "part", torch.float32)
    if rows > pr:
        raise HipError("colsum_segments: rows exceed the partial array")
    n = len(segs)
    for c0, nc, d, d2, sc in segs:
        _req(d, "dst", torch.float32)
        if d.numel() < nc or not d.is_contiguous() or c0 + nc > pc:
            raise HipError("colsum_segments: segment does not fit")
        if d2 is not None:
            _req(d2, "dst2", torch.float32)
            if d2.numel() < nc or not d2.is_contiguous():
                raise HipError("colsum_segments: dst2 too small")
    col0 = (ctypes.c_int32 * n)(*[s[0] for s in segs])
    ncols = (ctypes.c_int32 * n)(*[s[1] for s in segs])
    dst = (ctypes.c_void_p * n)(*[s[2].data_ptr() for s in segs])
    dst2 = (ctypes.c_void_p * n)(*[(s[3].data_ptr() if s[3] is not None else None) for s in segs])
    scale = (ctypes.c_float * n)(*[float(s[4]) for s in segs])
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _check(lib().ib_colsum_segments(_ptr(part), ld, rows, n, cv(col0), cv(ncols), cv(dst), cv(dst2), cv(scale),
                                    int(accumulate), stream_ptr()), "ib_colsum_segments")


def sum_partials(partial, parts: int, scale: float, out):
    _req(partial, "partial", torch.float32); _req(out, "out", torch.float32)
    if partial.numel() < parts:
        raise HipError("sum_partials: partial too small")
    _check(lib().ib_sum_partials(_ptr(partial), parts, float(scale), _ptr(out), stream_ptr()), "ib_sum_partials")
    return out


def ddim_step(x, eps, coef, timesteps, step=0, step_dev=None, t_out=None):
    dt = x.dtype
    _req(x, "x", dt)
    _req(eps, "eps", dt)
    if eps.shape != x.shape or not x.is_contiguous() or not eps.is_contiguous():
        raise HipError("ddim_step: x/eps must be contiguous with one shape")
    _req(coef, "coef", torch.float32, 2)
    _req(timesteps, "timesteps", torch.int64, 1)
    S = coef.shape[0]
    if coef.shape[1] != 2 or timesteps.numel() != S or not coef.is_contiguous():
        raise HipError("coef must be [S,2] fp32, timesteps [S] int64")
    B = x.shape[0]
    if t_out is not None:
        _req(t_out, "t_out", torch.int64, 1)
        if t_out.numel() != B:
            raise HipError("t_out must be int64 [B]")
    _check(lib().ib_ddim_step(_ptr(x), _ptr(eps), _ptr(coef), _ptr(timesteps), S, int(step), _ptr(step_dev),
                              _ptr(t_out), B, x.numel(), dtype_code(dt), stream_ptr()), "ib_ddim_step")
    return x


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, num_batches_tracked, y, save_mean, save_rstd, training: bool,
                  momentum: float = 0.1, eps: float = 1e-5):
    """nn.BatchNorm1d over [B, C] rows (torch defaults); in training mode the running statistics are updated in place"""
    dt = x.dtype
    B, C, ldx = _mat(x, "x", dt)
    By, Cy, ldy = _mat(y, "y", dt)
    if (By, Cy) != (B, C):
        raise HipError("batchnorm_fwd: x / y shapes differ")
    for name, t in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var),
                    ("save_mean", save_mean), ("save_rstd", save_rstd)):
        _req(t, name, torch.float32, 1)
        if t.numel() != C or not t.is_contiguous():
            raise HipError(f"batchnorm_fwd: {name} must be a contiguous [{C}] fp32 vector")
    if num_batches_tracked is not None:
        _req(num_batches_tracked, "num_batches_tracked", torch.int64)
    if training and B < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size " + str(tuple(x.shape)))
    _check(lib().ib_batchnorm_fwd(_ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                  _ptr(num_batches_tracked), _ptr(y), ldy, _ptr(save_mean), _ptr(save_rstd), B, C,
                                  float(momentum), float(eps), int(bool(training)), dtype_code(dt), stream_ptr()),
           "ib_batchnorm_fwd")


def batchnorm_bwd(dy, x, gamma, save_mean, save_rstd, dx, dgamma, dbeta, training: bool, accumulate=False,
                  act_below="none", aux=None):
    dt = dy.dtype
    B, C, lddy = _mat(dy, "dy", dt)
    Bx, Cx, ldx = _mat(x, "x", dt)
    if (Bx, Cx) != (B, C):
        raise HipError("batchnorm_bwd: dy / x shapes differ")
    lddx = 0
    if dx is not None:
        Bd, Cd, lddx = _mat(dx, "dx", dt)
        if (Bd, Cd) != (B, C):
            raise HipError("batchnorm_bwd: dx shape differs")
    for name, t in (("gamma", gamma), ("save_mean", save_mean), ("save_rstd", save_rstd), ("dgamma", dgamma),
                    ("dbeta", dbeta)):
        if t is None and name in ("dgamma", "dbeta"):
            continue
        _req(t, name, torch.float32, 1)
        if t.numel() != C or not t.is_contiguous():
            raise HipError(f"batchnorm_bwd: {name} must be a contiguous [{C}] fp32 vector")
    ldaux = 0
    if ACT[act_below] != 0:
        if aux is None:
            raise HipError("batchnorm_bwd: act_below needs aux")
        Ba, Ca, ldaux = _mat(aux, "aux", dt)
        if (Ba, Ca) != (B, C):
            raise HipError("batchnorm_bwd: aux shape differs")
    _check(lib().ib_batchnorm_bwd(_ptr(dy), lddy, _ptr(x), ldx, _ptr(gamma), _ptr(save_mean), _ptr(save_rstd), _ptr(dx), lddx,
                                  _ptr(dgamma), _ptr(dbeta), int(accumulate), ACT[act_below],
                                  _ptr(aux) if ACT[act_below] != 0 else None, ldaux, B, C, int(bool(training)), dtype_code(dt),
                                  stream_ptr()), "ib_batchnorm_bwd")


def scale_by_device_scalar(y: torch.Tensor, scale: torch.Tensor):
    """y *= scale (a one-element fp32 device tensor), in place; returns y"""
    _req(y, "y")
    _req(scale, "scale", torch.float32)
    if not y.is_contiguous() or scale.numel() != 1:
        raise HipError("scale_by_device_scalar: y must be contiguous, scale a single fp32 element")
    _check(lib().ib_scale_by_device_scalar(_ptr(y), _ptr(scale), y.numel(), dtype_code(y.dtype), stream_ptr()),
           "ib_scale_by_device_scalar")
    return y


def counter_add(counter, delta=1):
    _req(counter, "counter", torch.int32)
    _check(lib().ib_counter_add(_ptr(counter), int(delta), stream_ptr()), "ib_counter_add")


def fill_i64(dst, value):
    _req(dst, "dst", torch.int64)
    _check(lib().ib_fill_i64(_ptr(dst), int(value), dst.numel(), stream_ptr()), "ib_fill_i64")


# --------------------------------------------------------------------------------------------
# graphs / events
# --------------------------------------------------------------------------------------------
class Graph:
    """A hipGraph captured from the launches issued between begin() and end() on the current stream."""

    def __init__(self):
        self._exec = None

    def begin(self):
        _check(lib().ib_graph_begin(stream_ptr()), "ib_graph_begin")

    def end(self):
        h = ctypes.c_void_p()
        _check(lib().ib_graph_end(stream_ptr(), ctypes.byref(h)), "ib_graph_end")
        self._exec = h

    def launch(self):
        if self._exec is None:
            raise HipError("graph has not been captured")
        _check(lib().ib_graph_launch(self._exec, stream_ptr()), "ib_graph_launch")

    def __del__(self):
        try:
            if self._exec is not None and _lib is not None:
                _lib.ib_graph_destroy(self._exec)
        except Exception:
            pass


class Event:
    """hipEvent recorded on the CURRENT torch stream (the stream the kernels are launched on)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _check(lib().ib_event_create(ctypes.byref(h)), "ib_event_create")
        self._h = h

    def record(self):
        _check(lib().ib_event_record(self._h, stream_ptr()), "ib_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = ctypes.c_float()
        _check(lib().ib_event_elapsed_ms(self._h, stop._h, ctypes.byref(ms)), "ib_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if _lib is not None:
                _lib.ib_event_destroy(self._h)
        except Exception:
            pass


def selftest_tr16(inp: torch.Tensor, out: torch.Tensor):
    _check(lib().ib_selftest_tr16(_ptr(inp), _ptr(out), stream_ptr()), "ib_selftest_tr16")
    return out
