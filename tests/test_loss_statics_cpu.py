"""The reference's own 24 unit-test vectors for the four static loss helpers, restated as data tables
(source of the vectors: test/loss/test_RegressionLossEvaluator.py:7-192 of the reference -- inputs and expected
outputs only) and held against BOTH implementations this repo carries:

  * the drop-in statics of inferbiomechanics_amd.loss.RegressionLossEvaluator (API compatibility), and
  * the CPU oracle's restatement (oracle/ref_cpu.py: squared_diff_mean_vector, mask_by_threes, mean_norm_error,
    com_acc_error) that the GPU parity tests check the fused kernel against.

CPU only (no GPU, no library call).  tests/test_loss_edges_gpu.py pushes the same edge inputs through ib_regression_loss.
"""
import pytest
import torch

from inferbiomechanics_amd.loss.RegressionLossEvaluator import RegressionLossEvaluator as E
from oracle import ref_cpu as R

T = torch.tensor
ar24 = torch.arange(24, dtype=torch.float32).reshape(2, 4, 3)

IMPLS = {
    "dropin": dict(sq=E.get_squared_diff_mean_vector, mask=E.get_mask_by_threes, mne=E.get_mean_norm_error,
                   com=E.get_com_acc_error),
    "oracle": dict(sq=R.squared_diff_mean_vector, mask=R.mask_by_threes, mne=R.mean_norm_error, com=R.com_acc_error),
}

# ---- (helper, args, kwargs, expected value, exact?)   reference test file lines in the trailing comment
ROW_A = [[[3., -2., 4.], [4., 5., 6.]], [[3., -2., 4.], [4., 5., 6.]]]
ROW_B = [[[3., -2., 4.], [4., 5., 6.]], [[3., -2., 4.], [4., 5., 7.]]]
LAB_AB = [[[1., 2., 3.], [4., 5., 6.]], [[1., 2., 3.], [4., 5., 6.]]]
SIX = [[[1., 2., 3., 4., 5., 6.]]]
VALUE_CASES = [
    ("sq", (ar24, ar24), {}, T([0., 0., 0.]), True),                                                    # :7-13
    ("sq", (ar24, ar24 + 1.), {}, T([1., 1., 1.]), False),                                              # :15-21
    ("mask", (T([[[1., 0., 0.], [0., 2., 0.]], [[0., 0., 0.], [3., 0., 4.]]]),), {},
     T([[[1., 1., 1.], [1., 1., 1.]], [[0., 0., 0.], [1., 1., 1.]]]), True),                            # :37-48
    ("mask", (T([[[1., 0., 0.], [0., 2., 0.]]]),), {"threshold": 1.5}, T([[[0., 0., 0.], [1., 1., 1.]]]), True),  # :50-55
    ("mask", (T([[[0., 0., 0.], [0., 0., 0.]]]),), {}, T([[[0., 0., 0.], [0., 0., 0.]]]), True),        # :75-80
    ("mask", (T([[[0., 0., 1., 0., 0., 0.], [0., 0., 0., 1., 0., 0.]]]),), {},
     T([[[1., 1., 1., 0., 0., 0.], [0., 0., 0., 1., 1., 1.]]]), True),                                  # :82-87
    ("mne", (T(ROW_A), T(LAB_AB)), {}, T(0.), False),            # differs in the FIRST frame only -> 0     :113-126
    ("mne", (T(ROW_B), T(LAB_AB)), {}, T(0.5), False),           # last frame of one of two windows off by 1 :128-141
    ("mne", (T(SIX), T(SIX)), {"vec_size": 6}, T(0.), False),                                           # :143-150
    ("mne", (T(SIX), torch.zeros(1, 1, 6)), {"vec_size": 6}, torch.linalg.vector_norm(T(SIX)), False),  # :152-159
    ("com", (T([[[1., 2., 3., 0., 0., 0.], [0., 0., 0., 1., 2., 3.]]]),
             T([[[0., 0., 0., 1., 2., 3.], [1., 2., 3., 0., 0., 0.]]])), {}, T(0.), False),             # :185-192
]

# ---- inputs every implementation must refuse with ValueError
g = torch.Generator().manual_seed(7)
rnd = lambda *s: torch.rand(*s, generator=g)
ERROR_CASES = [
    ("sq", (T([[[1., 2.], [3., 4.]]]), T([[[1., 2., 3.], [4., 5., 6.]]]))),      # shapes differ        :23-28
    ("sq", (T([]), T([]))),                                                      # empty (and not 3-D)  :30-35
    ("mask", (T([[1., 0., 0.]]),)),                                              # 2-D                  :57-61
    ("mask", (torch.empty(0),)),                                                 # empty                :63-67
    ("mask", (T([[[1., 0.], [0., 2.]]]),)),                                      # last dim % 3 != 0    :69-73
    ("mne", (rnd(3, 2, 6), rnd(3, 2, 9))),                                       # shapes differ        :89-93
    ("mne", (rnd(2, 6), rnd(2, 6))),                                             # 2-D                  :95-99
    ("mne", (rnd(0, 6), rnd(0, 6))),                                             # empty                :101-105
    ("mne", (rnd(3, 2, 7), rnd(3, 2, 7))),                                       # last dim % 3 != 0    :107-111
    ("com", (rnd(3, 2, 6), rnd(4, 2, 6))),                                       # shapes differ        :161-165
    ("com", (rnd(2, 6), rnd(2, 6))),                                             # 2-D                  :167-171
    ("com", (torch.empty(0, 0), rnd(3, 6))),                                     # empty / mismatch     :173-177
    ("com", (rnd(3, 2, 5), rnd(3, 2, 5))),                                       # last dim != 6        :179-183
]


def test_the_tables_hold_all_24_reference_cases():
    assert len(VALUE_CASES) + len(ERROR_CASES) == 24


@pytest.mark.parametrize("impl", list(IMPLS))
@pytest.mark.parametrize("case", range(len(VALUE_CASES)))
def test_reference_value_vectors(impl, case):
    fn, args, kw, want, exact = VALUE_CASES[case]
    got = IMPLS[impl][fn](*args, **kw)
    if exact:
        assert torch.equal(got, want), (got, want)
    else:
        assert torch.allclose(got.reshape(want.shape), want), (got, want)


@pytest.mark.parametrize("impl", list(IMPLS))
@pytest.mark.parametrize("case", range(len(ERROR_CASES)))
def test_reference_error_vectors(impl, case):
    fn, args = ERROR_CASES[case]
    with pytest.raises(ValueError):
        IMPLS[impl][fn](*args)


def test_mask_threshold_is_strict():
    """norm == threshold is masked OUT (`>`; RegressionLossEvaluator.py:96-108): (6, 8, 0) has norm exactly 10"""
    t = T([[[6., 8., 0., 6., 8., 0.05]]])
    for impl in IMPLS.values():
        assert impl["mask"](t, threshold=10.0).reshape(-1).tolist() == [0., 0., 0., 1., 1., 1.]
