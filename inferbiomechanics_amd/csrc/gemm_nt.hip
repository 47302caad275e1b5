// Large-M bf16 "NT" GEMM for gfx950: C[M,N] = epilogue( sum_k A[m][k] * B[n][k] ), both operands k-contiguous.
//
// The training shapes of the transformer denoiser (BASELINE.json configs[2]/[3]: M = B*T = 12800 token rows,
// N, K in {512, 1536, 2048}) are what the 128 x 128 ring kernel of gemm.hip handles worst: with K = 512 a tile is 16
// K steps long, so its prologue (first-stage latency) and its epilogue (16 x 8-byte-per-lane stores per wave: store-ISSUE
// bound, ~as long as the K loop) weigh as much as the loop.  This kernel is built around those two ends:
//
//   * 256 x 128 output tile per 512-thread workgroup (8 waves as 4 x 2, 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles):
//     twice the flops per staged byte and per barrier of the 128^2 tile, K steps of 64 (32 MFMAs per wave per barrier);
//   * operands HBM/L2 -> LDS by global_load_lds_dwordx4 only (no VGPR round trip), three 48-KiB stages: two in flight
//     while one is consumed, counted s_waitcnt vmcnt, raw s_barrier, fragment reads as inline-asm ds_read_b128;
//   * LDS image = plain 128-byte rows (64 k) whose eight 16-byte pieces are XOR-swizzled by (row & 7): the fragment
//     reads of a 16-lane group then cover all sixteen 16-byte slots of the 256-byte bank row exactly once
//     (conflict-free), and because LDS-DMA writes lane-linearly the permutation is applied on the per-lane SOURCE
//     address (piece = slot ^ row) -- each row's 8 lanes still fetch one whole 128-byte line;
//   * PERSISTENT workgroups (one per CU) walk the tiles in an XCD-aware order (the column tiles of one row panel stay on
//     one XCD's L2), so consecutive tiles of a workgroup need no new launch and no LDS re-allocation;
//   * epilogue through LDS: every wave drops its 64 x 64 accumulators (bias / activation applied) as bf16 into a
//     [256][128] image, then all 512 threads move whole 16-byte pieces: the activation-derivative operand and the
//     residual addend are read, and C is written, in fully coalesced 256-byte row segments.
//
// Epilogues: forward  C = act(A B^T + bias)                  (nn.Linear + activation, src/models/TransformerBaseline.py:14-18)
//            backward C = (A B^T) * act'(aux) + addend       (autograd of the layer below + residual path; B = W^T here: the
//                                                            caller keeps a transposed bf16 copy of the weight, see
//                                                            ib_transpose_multi)
#include "ib_common.h"
#include "gemm_nt.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 64, NS = 3, NT_THREADS = 512;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;      // 32768 + 16384
constexpr int LDS_BYTES = NS * STAGE;                                                       // 147456
constexpr int CS = BN * 2 + 16;                                                             // C staging row stride (bytes)
static_assert(BM * CS <= 2 * STAGE, "the C staging image overlays two stages");

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
template <int OFF>
__device__ __forceinline__ bf16x8_t lds_read16(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void frags_ready(bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct NtParams {
  const bf16_t* A; const bf16_t* B; int64_t lda, ldb;
  int M, N, K;
  bf16_t* C; int64_t ldc;
  const float* bias;
  const bf16_t* aux; int64_t ldaux;
  const bf16_t* addend; int64_t ldadd;
  int tiles_m, tiles_n;
};

template <int ACT> __device__ __forceinline__ float nt_act(float v) {
  if constexpr (ACT == IB_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return tanhf(v);
  else if constexpr (ACT == IB_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  else if constexpr (ACT == IB_ACT_SILU) return v / (1.f + __expf(-v));
  else if constexpr (ACT == IB_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
  else return v;
}
template <int ACT> __device__ __forceinline__ float nt_act_bwd(float aux) {
  if constexpr (ACT == IB_ACT_RELU) return aux > 0.f ? 1.f : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return 1.f - aux * aux;
  else if constexpr (ACT == IB_ACT_SIGMOID) return aux * (1.f - aux);
  else if constexpr (ACT == IB_ACT_ELU) return aux > 0.f ? 1.f : aux + 1.f;
  else return 1.f;
}

// FWD_ACT: activation of the forward epilogue (bias added first); BWD_ACT: derivative factor taken from `aux`
// (IB_ACT_NONE = none); addend / bias are runtime-optional.
template <int FWD_ACT, int BWD_ACT>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_nt_kernel(NtParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.K / BK;
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = (int)gridDim.x;
  // XCD-aware walk: workgroups b and b + 8 share an XCD (round-robin dispatch) -> give each XCD a contiguous run of
  // the logical tile order (column tiles of one row panel are neighbours in it).  Speed only.
  const int slot_in_round = ib_xcd_remap((int)blockIdx.x, nwg);       // bijective for every grid size

  // per-lane constants of the staging and of the fragment reads
  const int srow = lane >> 3;                               // row inside an 8-row chunk
  const int spc = (lane & 7) ^ srow;                        // the 16-byte piece of that row this lane fetches
  const unsigned smem0 = lds_off(smem);
  const unsigned fragx = (unsigned)((((lane >> 4) ^ (lane & 7)) << 4));
  const unsigned a_base0 = smem0 + (unsigned)((wm * 64 + (lane & 15)) * 128) + fragx;
  const unsigned b_base0 = smem0 + A_BYTES + (unsigned)((wn * 64 + (lane & 15)) * 128) + fragx;

  for (int tile = slot_in_round; tile < tiles; tile += nwg) {
    const int ti = tile / p.tiles_n, tj = tile % p.tiles_n;
    const int i0 = ti * BM, j0 = tj * BN;
    const bf16_t* pa[4];
    const bf16_t* pb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(i0 + 8 * (wave + 8 * j) + srow, p.M - 1);
      pa[j] = p.A + (int64_t)row * p.lda + 8 * spc;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = min(j0 + 8 * (wave + 8 * j) + srow, p.N - 1);
      pb[j] = p.B + (int64_t)row * p.ldb + 8 * spc;
    }
    auto issue = [&](int kt) {
      unsigned char* st = smem + (kt % NS) * STAGE;
      const int k0 = kt * BK;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[j] + k0), (lds_void_t*)(st + (wave + 8 * j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[j] + k0), (lds_void_t*)(st + A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vm<6>(); else wait_vm<0>();       // my six pieces of stage kt have landed
      __builtin_amdgcn_s_barrier();                            // everyone's have; everyone finished reading stage kt-1
      if (kt + 2 < nk) issue(kt + 2);                          // into the slot stage kt-1 occupied
      const unsigned so = (unsigned)((kt % NS) * STAGE);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned aa = (a_base0 + so) ^ (ks ? 64u : 0u), bb = (b_base0 + so) ^ (ks ? 64u : 0u);
        bf16x8_t fa[4], fb[4];
        fa[0] = lds_read16<0>(aa); fa[1] = lds_read16<2048>(aa); fa[2] = lds_read16<4096>(aa); fa[3] = lds_read16<6144>(aa);
        fb[0] = lds_read16<0>(bb); fb[1] = lds_read16<2048>(bb); fb[2] = lds_read16<4096>(bb); fb[3] = lds_read16<6144>(bb);
        frags_ready(fa, fb);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[t], acc[t][u], 0, 0, 0);
      }
    }

    // ---- epilogue.  The C image overlays stages 1 and 2; stage slots are only reused after the barrier below, and the
    // next tile's first issue comes after the trailing barrier.
    __builtin_amdgcn_s_barrier();                              // every wave is done reading the last stages
    unsigned char* cimg = smem + STAGE;
    {
      const int rl = wm * 64 + (lane & 15), cl = wn * 64 + 4 * (lane >> 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          const int n = min(j0 + cl + 16 * u, p.N - 4);
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
          b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bf16x4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)nt_act<FWD_ACT>(acc[t][u][r] + b4[r]);
          *reinterpret_cast<bf16x4_t*>(cimg + (rl + 16 * t) * CS + (cl + 16 * u) * 2) = o;
        }
      }
    }
    __syncthreads();
    {
      // 256 rows x 16 pieces of 16 bytes; thread -> piece (tid & 15) of rows (tid >> 4) + 32 q
      const int pc = tid & 15, r0 = tid >> 4;
      const int n = j0 + 8 * pc;
      if (n < p.N) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int row = r0 + 32 * q, m = i0 + row;
          if (m < p.M) {
            bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(cimg + row * CS + pc * 16);
            if constexpr (BWD_ACT != IB_ACT_NONE) {
              const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(p.aux + (int64_t)m * p.ldaux + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * nt_act_bwd<BWD_ACT>((float)a8[e]));
            }
            if (p.addend) {
              const bf16x8_t d8 = *reinterpret_cast<const bf16x8_t*>(p.addend + (int64_t)m * p.ldadd + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)d8[e]);
            }
            *reinterpret_cast<bf16x8_t*>(p.C + (int64_t)m * p.ldc + n) = v;
          }
        }
      }
    }
    __syncthreads();                                           // the C image is consumed: stages may be refilled
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int FWD_ACT, int BWD_ACT>
int launch(const NtParams& p, hipStream_t s) {
  const int tiles = p.tiles_m * p.tiles_n;
  const int grid = tiles < 256 ? tiles : 256;                 // one persistent workgroup per CU
  hipLaunchKernelGGL((gemm_nt_kernel<FWD_ACT, BWD_ACT>), dim3(grid), dim3(NT_THREADS), 0, s, p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

}  // namespace

// Returns IB_E_UNSUPPORTED when the problem does not qualify (the caller then takes the generic kernels).
int ib_gemm_nt_try(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, int fwd_act,
                   const void* aux, int64_t ldaux, int bwd_act, const void* addend, int64_t ldadd, int64_t M, int64_t N,
                   int64_t K, hipStream_t s) {
  static const int off = []() { const char* e = getenv("IB_NO_NT"); return e ? atoi(e) : 0; }();
  static const int min_m = []() { const char* e = getenv("IB_NT_MIN_M"); return e ? atoi(e) : 4096; }();
  if (off || M < min_m || N < 128 || N % 8 != 0 || K < 128 || K % BK != 0) return IB_E_UNSUPPORTED;
  if (!al16(A) || !al16(B) || !al16(C) || lda % 8 || ldb % 8 || ldc % 8) return IB_E_UNSUPPORTED;
  if (bias && !al16(bias)) return IB_E_UNSUPPORTED;
  if (aux && (!al16(aux) || ldaux % 8)) return IB_E_UNSUPPORTED;
  if (addend && (!al16(addend) || ldadd % 8)) return IB_E_UNSUPPORTED;
  if (bwd_act != IB_ACT_NONE && (bwd_act == IB_ACT_SILU || !aux || fwd_act != IB_ACT_NONE)) return IB_E_UNSUPPORTED;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.C = (bf16_t*)C; p.ldc = ldc; p.bias = bias; p.aux = (const bf16_t*)aux; p.ldaux = ldaux;
  p.addend = (const bf16_t*)addend; p.ldadd = ldadd;
  p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
  if (bwd_act != IB_ACT_NONE) {
    switch (bwd_act) {
      case IB_ACT_RELU: return launch<IB_ACT_NONE, IB_ACT_RELU>(p, s);
      case IB_ACT_TANH: return launch<IB_ACT_NONE, IB_ACT_TANH>(p, s);
      case IB_ACT_SIGMOID: return launch<IB_ACT_NONE, IB_ACT_SIGMOID>(p, s);
      case IB_ACT_ELU: return launch<IB_ACT_NONE, IB_ACT_ELU>(p, s);
      default: return IB_E_UNSUPPORTED;
    }
  }
  switch (fwd_act) {
    case IB_ACT_NONE: return launch<IB_ACT_NONE, IB_ACT_NONE>(p, s);
    case IB_ACT_RELU: return launch<IB_ACT_RELU, IB_ACT_NONE>(p, s);
    case IB_ACT_TANH: return launch<IB_ACT_TANH, IB_ACT_NONE>(p, s);
    case IB_ACT_SIGMOID: return launch<IB_ACT_SIGMOID, IB_ACT_NONE>(p, s);
    case IB_ACT_SILU: return launch<IB_ACT_SILU, IB_ACT_NONE>(p, s);
    case IB_ACT_ELU: return launch<IB_ACT_ELU, IB_ACT_NONE>(p, s);
    default: return IB_E_UNSUPPORTED;
  }
}

// ---- dst_i[c][r] = src_i[r][c] for several bf16 matrices in ONE launch (the transposed weight copies the backward
// GEMMs read k-contiguously; refreshed once per step after the optimizer moved the weights).  64 x 64 tiles through LDS.
namespace {
constexpr int TR_MAX = 32;
struct TrMulti { const bf16_t* src[TR_MAX]; bf16_t* dst[TR_MAX]; int rows[TR_MAX], cols[TR_MAX], lds[TR_MAX], ldd[TR_MAX], blk0[TR_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(TrMulti m) {
  __shared__ bf16_t tile[64][66];
  int e = 0;
  for (int j = 1; j < m.n; ++j)
    if ((int)blockIdx.x >= m.blk0[j]) e = j;
  const int b = (int)blockIdx.x - m.blk0[e];
  const int R = m.rows[e], Cc = m.cols[e];
  const int tc = (Cc + 63) / 64;
  const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
  const bf16_t* src = m.src[e];
  bf16_t* dst = m.dst[e];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    tile[r][tx] = (r0 + r < R && c0 + tx < Cc) ? src[(int64_t)(r0 + r) * m.lds[e] + c0 + tx] : (bf16_t)0.f;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < Cc && r0 + tx < R) dst[(int64_t)(c0 + c) * m.ldd[e] + r0 + tx] = tile[tx][c];
}
}  // namespace

extern "C" int ib_transpose_multi(int n, const void* const* src, const int64_t* lds, void* const* dst, const int64_t* ldd,
                                  const int64_t* rows, const int64_t* cols, int dtype, ib_stream_t stream) {
  if (n <= 0 || n > TR_MAX || !src || !dst || !lds || !ldd || !rows || !cols) return IB_E_ARG;
  if (dtype != IB_BF16) return IB_E_DTYPE;
  TrMulti m{};
  m.n = n;
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0 || lds[i] < cols[i] || ldd[i] < rows[i]) return IB_E_ARG;
    m.src[i] = (const bf16_t*)src[i]; m.dst[i] = (bf16_t*)dst[i];
    m.rows[i] = (int)rows[i]; m.cols[i] = (int)cols[i]; m.lds[i] = (int)lds[i]; m.ldd[i] = (int)ldd[i];
    m.blk0[i] = blk;
    blk += (int)(((rows[i] + 63) / 64) * ((cols[i] + 63) / 64));
  }
  m.blk0[n] = blk;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(blk), dim3(256), 0, ib_s(stream), m);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
