"""The large-M bf16 NT GEMM (csrc/gemm_nt.hip: 256 x 128 tiles, LDS-DMA staging, XOR-swizzled LDS image, persistent
workgroups, epilogue through LDS) behind ib_linear_fwd / ib_linear_dgrad_wt, against fp32 matmuls of the same bf16 operands.
Integer-valued operands make every product and partial sum exact in fp32, so the result is bit-for-bit the reference's:
a wrong lane map, swizzle or tile index shows up as a mismatch, not as "rounding".  Ragged M, N not a multiple of the tile,
the persistent multi-tile walk, every epilogue operand.  -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferbiomechanics_amd import hip
    hip.lib()


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


SHAPES = [(4096, 512, 512), (4100, 128, 128), (5000, 200, 192), (12800, 2048, 512), (12800, 512, 2048), (4352, 1536, 512),
          (70000, 128, 128)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nt_forward_exact_on_integer_operands(M, N, K):
    from inferbiomechanics_amd import hip
    x = ints((M, K), -3, 3, 1).to(DEV, torch.bfloat16)
    w = ints((N, K), -2, 2, 2).to(DEV, torch.bfloat16)       # |sum| <= 6 K <= 12288: exact in fp32, bf16 output rounds
    b = ints((N,), -4, 4, 3).to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    hip.linear_fwd(x, w, b, y, act="relu")
    ref = torch.relu(x.float() @ w.float().t() + b).to(torch.bfloat16)
    assert torch.equal(y, ref), (y.float() - ref.float()).abs().max()
    y2 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    hip.linear_fwd(x, w, None, y2)
    assert torch.equal(y2, (x.float() @ w.float().t()).to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(4096, 512, 512), (12800, 512, 2048), (12800, 2048, 512), (4100, 1536, 512)])
def test_nt_dgrad_with_transposed_weight_matches_generic_path(M, N, K):
    """dx = (dz w) * relu'(aux) + addend: the transposed-weight path against an fp32 reference and against the generic
    ib_linear_dgrad (k-strided weight) on the same operands"""
    from inferbiomechanics_amd import hip
    dz = ints((M, N), -3, 3, 4).to(DEV, torch.bfloat16)
    w = ints((N, K), -2, 2, 5).to(DEV, torch.bfloat16)
    aux = ints((M, K), -1, 1, 6).to(DEV, torch.bfloat16)
    add = ints((M, K), -5, 5, 7).to(DEV, torch.bfloat16)
    wt = torch.empty(K, N, device=DEV, dtype=torch.bfloat16)
    hip.transpose_multi([(w, wt)])
    assert torch.equal(wt, w.t().contiguous())
    dx = torch.full((M, K), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert hip.linear_dgrad_wt(dz, wt, dx, act_below="relu", aux=aux, addend=add)
    prod = (dz.float() @ w.float()).to(torch.bfloat16).float()          # the kernel rounds the product to bf16 first
    ref = ((prod * (aux.float() > 0)).to(torch.bfloat16).float() + add.float()).to(torch.bfloat16)
    assert torch.equal(dx, ref), (dx.float() - ref.float()).abs().max()
    dx2 = torch.empty_like(dx)
    assert hip.linear_dgrad_wt(dz, wt, dx2)
    assert torch.equal(dx2, (dz.float() @ w.float()).to(torch.bfloat16))


def test_nt_random_operands_close_to_fp32_reference():
    from inferbiomechanics_amd import hip
    g = torch.Generator().manual_seed(11)
    M, N, K = 12800, 1536, 512
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.linear_fwd(x, w, b, y)
    ref = x.float() @ w.float().t() + b
    err = (y.float() - ref).abs().max().item()
    assert err <= 2 ** -7 * ref.abs().max().item(), err            # one bf16 rounding of the fp32-accumulated result


def test_transpose_multi_many_matrices_and_ragged_edges():
    from inferbiomechanics_amd import hip
    g = torch.Generator().manual_seed(3)
    pairs = []
    for i, (r, c) in enumerate([(512, 2048), (2048, 512), (70, 130), (1, 64), (65, 1)] * 7):
        src = torch.randn(r, c, generator=g).to(DEV, torch.bfloat16)
        pairs.append((src, torch.zeros(c, r, device=DEV, dtype=torch.bfloat16)))
    hip.transpose_multi(pairs)            # 35 matrices: two launches
    for src, dst in pairs:
        assert torch.equal(dst, src.t().contiguous())
