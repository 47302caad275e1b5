"""CPU-side checks: the C-ABI library loads and exports every symbol include/ib_hip.h declares (no
compute calls without a GPU), host-side logic (flat layout, schedule tables bit-exact vs the oracle,
component weights), the drop-in surface (state_dict names, registry), and loud failure off-GPU."""
import argparse
import ctypes

import pytest
import torch

from oracle import ref_cpu as R


def test_library_loads_and_exports_every_declared_symbol():
    from inferbiomechanics_amd import hip
    lib = hip.lib()
    names = hip.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libib_hip.so does not export {n}"
        assert n in hip._SIGS, f"{n} has no ctypes signature"
    assert lib.ib_version() >= 100
    assert b"workspace" in lib.ib_error_string(-4)


def test_null_arguments_return_error_codes_not_crashes():
    from inferbiomechanics_amd import hip
    lib = hip.lib()
    assert lib.ib_linear_fwd(None, 0, None, 0, None, None, 0, None, 0, 0, 0, None, 0, None, 0, 4, 4, 4, 0, None) == -1
    assert lib.ib_optim_step(9, None, None, None, None, 0, 0.0, 1.0, 1, None, None, None, None) == -1
    assert lib.ib_linear_wgrad_workspace(12800, 512, 512) > 0
    assert lib.ib_linear_wgrad_workspace(4, 8, 8) == 0


def test_whole_layer_launches_validate_their_arguments():
    """round 5: the launches with the attention inside check shape, panel geometry and pointers on the host and return error
    codes -- a NULL `mask` would have been a device fault (the forward stores its ReLU bit words through it)"""
    from inferbiomechanics_amd import hip
    lib = hip.lib()
    assert lib.ib_ffn_chain_attn_workgroups(12800, 512, 2048, 50) == 256
    assert lib.ib_ffn_chain_attn_workgroups(4096, 512, 1024, 64) == 64
    for M, d, ffn, T in ((12800, 512, 2048, 8), (12800, 512, 2048, 65), (12800, 512, 2048, 48), (12800, 256, 2048, 50),
                         (12800, 512, 2000, 50), (12800, 512, 2048, 0)):
        assert lib.ib_ffn_chain_attn_workgroups(M, d, ffn, T) == 0, (M, d, ffn, T)
    assert lib.ib_ffn_chain_attn_mask_bytes(12800, 512, 2048, 50) == 256 * 4 * 512 * 8
    buf = ctypes.create_string_buffer(4096)
    a = ctypes.cast(buf, ctypes.c_void_p).value // 16 * 16 + 16          # any aligned non-NULL address: nothing is launched
    P = lambda ok=True: ctypes.c_void_p(a) if ok else None
    fwd = lambda **k: lib.ib_ffn_chain_fwd_attn(*[P(k.get(f"a{i}", True)) for i in range(25)], k.get("T", 50), k.get("M", 12800),
                                                k.get("d", 512), k.get("ffn", 2048), 1e-5, None)
    assert fwd(T=0) == -1                                         # IB_E_ARG: no panel geometry
    assert fwd(T=48) == -5 and fwd(d=256) == -5                   # IB_E_UNSUPPORTED: M % T != 0 / d != 512
    assert fwd(a11=False) == -1                                   # mask
    assert fwd(a24=False) == -1                                   # attention tail without lse
    assert fwd(a22=False) == -1                                   # attention tail without the QKV tail
    assert lib.ib_ffn_chain_fwd(*[P(i != 11) for i in range(23)], 12800, 512, 2048, 1e-5, None) == -1       # mask NULL
    bwd = lambda **k: lib.ib_ffn_chain_bwd_attn(*[P(k.get(f"a{i}", True)) for i in range(19)], k.get("T", 50), k.get("M", 12800),
                                                512, 2048, None)
    assert bwd(T=0) == -1 and bwd(T=48) == -5
    for i in (0, 6, 10, 15, 16, 17, 18):                          # dy, mask, s1, qkv, lse, dqkv, dx
        assert bwd(**{f"a{i}": False}) == -1, i
    inf = lambda **k: lib.ib_ffn_chain_fwd_infer(*[P(k.get(f"a{i}", True)) for i in range(14)], k.get("M", 51200), k.get("d", 512),
                                                 2048, 1e-5, None)
    assert inf(d=256) == -5 and inf(a6=False) == -1 and inf(a7=False) == -1 and inf(a13=True, a11=False) == -1


def test_schedule_tables_bit_exact_vs_oracle():
    from inferbiomechanics_amd.diffusion import schedule as S
    ab = S.alphas_cumprod(1000)
    assert torch.equal(ab, R.alphas_cumprod(R.linear_beta_schedule(1000)))
    assert abs(ab[499].item() - 0.078587242881778235) < 1e-15
    assert torch.equal(S.ddim_timesteps(1000, 100), R.ddim_timesteps(1000, 100))
    assert torch.equal(S.ddim_coefficients(1000, 100), R.ddim_coeffs(1000, 100))
    t = torch.arange(1000)
    assert torch.equal(S.timestep_embedding_table(1000, 128), R.timestep_embedding(t, 128))
    tabs = S.DiffusionTables(torch.device("cpu"))
    assert tabs.sqrt_ab.dtype == torch.float32 and tabs.ddim_t.dtype == torch.int64
    assert torch.equal(tabs.sqrt_ab, R.schedule_tables()["sqrt_ab"].to(torch.float32))


def test_flat_layout_alignment():
    from collections import OrderedDict
    from inferbiomechanics_amd.module import flat_layout
    lay, total = flat_layout(OrderedDict(a=(3, 5), b=(7,), c=(64, 2)))
    assert lay["a"] == (0, 15) and lay["b"] == (64, 7) and lay["c"] == (128, 128) and total == 256


def test_component_weights():
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import component_weights
    a = argparse.Namespace(predict_grf_components=[1], predict_cop_components=[], predict_moment_components=[],
                           predict_wrench_components=[])
    w = component_weights(a)
    assert sum(w) == 1.0 and w[1] == 1.0
    a = argparse.Namespace(predict_grf_components=list(range(6)), predict_cop_components=list(range(6)),
                           predict_moment_components=list(range(6)), predict_wrench_components=list(range(12)))
    assert component_weights(a) == [1.0] * 30


def test_drop_in_surface_names_match_reference():
    from inferbiomechanics_amd.models.DiffusionDenoisers import DiffusionMLP, DiffusionTransformer
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from inferbiomechanics_amd.models.TransformerBaseline import TransformerLayer
    m = FeedForwardBaseline(23, 2, 50, 'all_frames', 'sigmoid', 5, 10)
    assert list(m.state_dict().keys()) == ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias',
                                           'net.4.weight', 'net.4.bias']
    assert m.input_size == 1470 and m.output_size == 300          # FeedForwardRegressionBaseline.py:52,63
    assert sum(p.numel() for p in m.parameters()) == 1169708
    assert list(TransformerLayer(512, 8, 2048).state_dict().keys()) == R.TL_KEYS
    d = DiffusionMLP()
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == R.denoiser_mlp_param_shapes(300, [512, 512])
    d2 = DiffusionTransformer()
    assert {k: tuple(v.shape) for k, v in d2.state_dict().items()} == R.denoiser_transformer_param_shapes(300, 50)


def test_no_cpu_fallback():
    from inferbiomechanics_amd import hip
    from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator
    from inferbiomechanics_amd.models.FeedForwardRegressionBaseline import FeedForwardBaseline
    from oracle.fixture_inputs import ff_inputs
    m = FeedForwardBaseline(23, 2, 50, 'all_frames', 'sigmoid', 5, 10)
    with pytest.raises(hip.HipError):
        m(ff_inputs(2, 10, 23, 5))
    with pytest.raises(hip.HipError):
        hip.linear_fwd(torch.zeros(2, 2), torch.zeros(2, 2), None, torch.zeros(2, 2))
    ev = RegressionLossEvaluator(None, 'train', device='cpu')
    with pytest.raises(hip.HipError):
        ev({}, {}, {}, [], [], argparse.Namespace())


def test_shipping_library_reads_no_environment_and_has_no_profiling_hooks():
    """no global mutable state in the C-ABI.  The A/B switches (IB_NO_NT, IB_TN_TARGET, ...) and the in-kernel
    stamp hooks are compiled only into the measurement build (lib/ab/libib_hip_ab.so, -DIB_AB); the shipping library imports
    no getenv at all, answers IB_E_UNSUPPORTED to every ib_debug_set_* call, and both builds export the same symbols"""
    import os
    import subprocess
    from inferbiomechanics_amd import hip
    assert os.path.basename(hip.LIB_PATH) == "libib_hip.so" and not hip.measurement_build()
    und = subprocess.run(["nm", "-D", "--undefined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    lib = hip.lib()
    for setter in ("ib_debug_set_chain_prof", "ib_debug_set_gemm_prof", "ib_debug_set_nt_prof"):
        assert getattr(lib, setter)(None) == -5, setter
    assert lib.ib_debug_set_ablate(1) == -5
    assert os.path.exists(hip.AB_LIB_PATH), "the measurement build was not built (make -C inferbiomechanics_amd/csrc)"
    ab = ctypes.CDLL(hip.AB_LIB_PATH)
    for n in hip.declared_symbols():
        assert hasattr(ab, n), f"libib_hip_ab.so does not export {n}"
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", hip.AB_LIB_PATH], capture_output=True, text=True,
                                      check=True).stdout
