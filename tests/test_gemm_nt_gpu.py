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
          (70000, 128, 128),
          # the NT kernel takes over from 640 rows (IB_NT_MIN_M): fewer tiles than workgroups, ragged last row panel,
          # N not a multiple of the 128-column tile, K = 256 (the shortest stage stream) and K = 320
          (640, 512, 512), (641, 136, 256), (700, 304, 320), (1000, 2048, 512), (3200, 1536, 512), (1023, 128, 1024)]


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


# ---- weight gradients of long reductions (csrc/gemm_tn.hip: transposing LDS reads, swizzled [m][column] images) -------------
def _wgrad_ref(dz, x):
    return dz.float().t() @ x.float()


@pytest.mark.parametrize("M,N,K", [(4096, 512, 512), (12800, 2048, 512), (12800, 512, 2048), (12800, 1536, 512),
                                   (12800, 256, 128), (6400, 320, 200)])
def test_tn_wgrad_slabs_exact_on_integer_operands(M, N, K):
    """sum of the split slabs == dz^T x bit for bit (integer operands: every partial sum is exact in fp32)"""
    from inferbiomechanics_amd import hip
    dz = ints((M, N), -3, 3, 21).to(DEV, torch.bfloat16)
    x = ints((M, K), -2, 2, 22).to(DEV, torch.bfloat16)
    ws = torch.full((int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)),), 0x7F, dtype=torch.uint8, device=DEV)
    ns = hip.linear_wgrad_slabs(dz, x, ws)
    slabs = ws[:ns * N * K * 4].view(torch.float32).view(ns, N, K)
    assert torch.equal(slabs.sum(0), _wgrad_ref(dz, x)), (ns, (slabs.sum(0) - _wgrad_ref(dz, x)).abs().max())
    dw = torch.empty(N, K, device=DEV)
    hip.slab_reduce_multi([(ws, ns, dw)])
    assert torch.equal(dw, _wgrad_ref(dz, x))


def test_tn_grouped_launch_with_ragged_widths_and_bias_partials():
    """the MLP denoiser's problems (D = 300 operands in 304-pitch buffers) and a transformer layer's four problems, each group
    in ONE launch, with the bias gradients' per-split partial sums"""
    from inferbiomechanics_amd import hip
    M = 12800
    groups = [[(300, 512), (512, 512), (512, 300)], [(512, 2048), (2048, 512), (512, 512), (1536, 512)]]
    for gi, shapes in enumerate(groups):
        probs, parts, refs = [], [], []
        for j, (N, K) in enumerate(shapes):
            dzb = torch.zeros(M, (N + 7) // 8 * 8, device=DEV, dtype=torch.bfloat16)
            xb = torch.zeros(M, (K + 7) // 8 * 8, device=DEV, dtype=torch.bfloat16)
            dzb[:, N:] = 7.0                                   # pad columns hold junk that must never reach an output
            xb[:, K:] = -5.0
            dz, x = dzb[:, :N], xb[:, :K]
            dz.copy_(ints((M, N), -3, 3, 30 + 10 * gi + j).to(DEV, torch.bfloat16))
            x.copy_(ints((M, K), -2, 2, 40 + 10 * gi + j).to(DEV, torch.bfloat16))
            ws = torch.full((int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)),), 0x7F, dtype=torch.uint8, device=DEV)
            probs.append((dz, x, ws))
            parts.append(torch.full((32, N), float("nan"), device=DEV))
            refs.append((_wgrad_ref(dz, x), dz.float().sum(0)))
        ns = hip.linear_wgrad_slabs_multi(probs, bias_parts=parts)
        assert ns is not None
        for (dz, x, ws), n, part, (rw, rb) in zip(probs, ns, parts, refs):
            N, K = dz.shape[1], x.shape[1]
            slabs = ws[:n * N * K * 4].view(torch.float32).view(n, N, K)
            assert torch.equal(slabs.sum(0), rw), (N, K, n)
            assert torch.equal(part[:n].sum(0), rb), (N, K, n)


def test_tn256_persistent_walk_over_more_items_than_workgroups():
    """csrc/gemm_tn256.hip (256 x 256 tiles, one split count per group): a group of 272 + 2 tiles on 256 persistent
    workgroups -- some workgroups take a second item, across the problem boundary, with the stage stream running on through
    the first item's epilogue; exact on integer operands, bias partial sums included"""
    from inferbiomechanics_amd import hip
    M = 4096
    shapes = [(4096, 4352), (512, 256)]
    probs, parts, refs = [], [], []
    for j, (N, K) in enumerate(shapes):
        dz = ints((M, N), -3, 3, 70 + j).to(DEV, torch.bfloat16)
        x = ints((M, K), -2, 2, 80 + j).to(DEV, torch.bfloat16)
        ws = torch.full((int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)),), 0x7F, dtype=torch.uint8, device=DEV)
        probs.append((dz, x, ws))
        parts.append(torch.full((32, N), float("nan"), device=DEV))
        refs.append((_wgrad_ref(dz, x), dz.float().sum(0)))
    hip.lib().ib_debug_last_path()                       # read-and-clear
    ns = hip.linear_wgrad_slabs_multi(probs, bias_parts=parts)
    path = hip.PATH_NAMES[int(hip.lib().ib_debug_last_path())]
    assert ns is not None and path == "tn256x256", (ns, path)
    for (dz, x, ws), n, part, (rw, rb) in zip(probs, ns, parts, refs):
        N, K = dz.shape[1], x.shape[1]
        slabs = ws[:n * N * K * 4].view(torch.float32).view(n, N, K)
        assert torch.equal(slabs.sum(0), rw), (N, K, n, (slabs.sum(0) - rw).abs().max())
        assert torch.equal(part[:n].sum(0), rb), (N, K, n)


def test_tn256_takes_the_transformer_layer_group_with_one_split_count():
    from inferbiomechanics_amd import hip
    M = 12800
    g = torch.Generator().manual_seed(9)
    probs = []
    for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
        dz = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
        x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
        ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=DEV)
        probs.append((dz, x, ws))
    hip.lib().ib_debug_last_path()
    ns = hip.linear_wgrad_slabs_multi(probs)
    path = hip.PATH_NAMES[int(hip.lib().ib_debug_last_path())]
    assert path == "tn256x256" and len(set(ns)) == 1 and ns[0] * 48 <= 256, (path, ns)
    for (dz, x, ws), n in zip(probs, ns):
        N, K = dz.shape[1], x.shape[1]
        got = ws[:n * N * K * 4].view(torch.float32).view(n, N, K).sum(0)
        ref = _wgrad_ref(dz, x)
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_tn_wgrad_is_bitwise_reproducible():
    from inferbiomechanics_amd import hip
    g = torch.Generator().manual_seed(5)
    M, N, K = 12800, 2048, 512
    dz = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    outs = []
    for _ in range(2):
        ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)), dtype=torch.uint8, device=DEV)
        ns = hip.linear_wgrad_slabs(dz, x, ws)
        dw = torch.empty(N, K, device=DEV)
        hip.slab_reduce_multi([(ws, ns, dw)])
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    ref = _wgrad_ref(dz, x)
    assert (outs[0] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()      # fp32 summation order only
