// fp32 (exact-f32 MFMA) forward / dgrad for batches of a few rows: the reference's own training shape -- FeedForwardBaseline
// 1470 -> 512 -> 512 -> 300 at --batch-size 4 .. 64 in fp32 (BASELINE.json configs[0]; src/cli/train.py:52,
// src/models/FeedForwardRegressionBaseline.py:52,63).
//
// The generic fp32 kernel of gemm.hip tiles 128 x 128: [4, 512, 1470] is 4 workgroups walking 46 K steps in sequence
// (110 us), the whole fp32 regression step 0.38 ms -- launch + latency, 0.3 % of anything.  Here a 256-thread workgroup owns
// a 16 x 16 output tile, its four waves SPLIT THE REDUCTION (super-steps of 8 k: wave w takes steps w, w + 4, ...) and
// combine their accumulators through LDS in wave order (deterministic), so [4, 512, 1470] is 32 workgroups x 46 super-steps
// per wave and [64, ...] 128 workgroups.  v_mfma_f32_16x16x4_f32 (exact f32: one rounding per product, like an fmaf chain;
// MI355X_MICROARCH.md, Matrix cores) with both operands loaded as 8-byte pieces: lane (r, kq) loads k = 8 S + 2 kq + e,
// e = 0, 1, and MFMA step e of super-step S multiplies the k's {8 S + 2 kq' + e}: the reduction index is permuted
// identically on both operands, so no lane movement is needed.
#include "ib_common.h"
#include <stdlib.h>

namespace {

constexpr int FS_PD = 4;         // super-steps in flight per wave

template <int ACT>
__device__ __forceinline__ float fs_act(float v) {
  if constexpr (ACT == IB_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return tanhf(v);
  else if constexpr (ACT == IB_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  else if constexpr (ACT == IB_ACT_SILU) return v / (1.f + expf(-v));
  else if constexpr (ACT == IB_ACT_ELU) return v > 0.f ? v : expf(v) - 1.f;
  else return v;
}
template <int ACT>
__device__ __forceinline__ float fs_act_bwd(float aux) {
  if constexpr (ACT == IB_ACT_RELU) return aux > 0.f ? 1.f : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return 1.f - aux * aux;
  else if constexpr (ACT == IB_ACT_SIGMOID) return aux * (1.f - aux);
  else if constexpr (ACT == IB_ACT_ELU) return aux > 0.f ? 1.f : aux + 1.f;
  else if constexpr (ACT == IB_ACT_SILU) {
    const float sg = 1.f / (1.f + expf(-aux));
    return sg * (1.f + aux * (1.f - sg));
  } else return 1.f;
}

// y[M,N] = act(x[M,K] w[N,K]^T + bias)  (+ optional pre-activation z)
template <int ACT>
__global__ __launch_bounds__(256) void f32_small_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                            int64_t ldw, const float* __restrict__ bias, float* __restrict__ y,
                                                            int64_t ldy, float* __restrict__ z, int64_t ldz, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float red[4][16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const float* arow = x + (int64_t)min(i0 + r, M - 1) * ldx + 2 * kq;      // B operand: x row m = r
  const float* brow = w + (int64_t)min(j0 + r, N - 1) * ldw + 2 * kq;      // A operand: w row n = r
  const int nS = (K + 7) / 8;
  const int nmine = wave < nS ? (nS - wave + 3) / 4 : 0;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float2 fa[FS_PD], fb[FS_PD];
  auto load = [&](int s, int it) {
    const int k = (wave + 4 * it) * 8;
    const bool in = k + 2 * kq < K;                                      // K even: both elements of the piece or neither
    fa[s] = in ? *reinterpret_cast<const float2*>(arow + k) : make_float2(0.f, 0.f);
    fb[s] = in ? *reinterpret_cast<const float2*>(brow + k) : make_float2(0.f, 0.f);
  };
#pragma unroll
  for (int s = 0; s < FS_PD; ++s)
    if (s < nmine) load(s, s);
  for (int it0 = 0; it0 < nmine; it0 += FS_PD) {
#pragma unroll
    for (int s = 0; s < FS_PD; ++s) {
      const int it = it0 + s;
      if (it < nmine) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s].x, fa[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s].y, fa[s].y, acc, 0, 0, 0);
        if (it + FS_PD < nmine) load(s, it + FS_PD);
      }
    }
  }
  // swapped operands (A = w, B = x): this lane holds output columns j0 + 4 kq .. +3 of row i0 + r
  *reinterpret_cast<f32x4_t*>(&red[wave][r * 16 + 4 * kq]) = acc;
  __syncthreads();
  if (tid < 64) {
    const int row = tid >> 2, c4 = (tid & 3) * 4;
    const int gi = i0 + row, gj = j0 + c4;
    if (gi < M && gj < N) {
      float v[4], pre[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = row * 16 + c4 + e;
        pre[e] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        if (bias && gj + e < N) pre[e] += bias[gj + e];
        v[e] = fs_act<ACT>(pre[e]);
      }
      float* dst = y + (int64_t)gi * ldy + gj;
      float* dz_ = z ? z + (int64_t)gi * ldz + gj : nullptr;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gj + e < N) {
          dst[e] = v[e];
          if (dz_) dz_[e] = pre[e];
        }
    }
  }
}

// dx[M,K] = (dz[M,N] w[N,K]) * act'(aux[M,K]) + addend[M,K]     (reduction over n; w rows are the reduction index)
template <int ACT>
__global__ __launch_bounds__(256) void f32_small_dgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                              const float* __restrict__ w, int64_t ldw,
                                                              const float* __restrict__ aux, int64_t ldaux,
                                                              const float* __restrict__ addend, int64_t ldadd,
                                                              float* __restrict__ dx, int64_t lddx, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float red[4][16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;                   // output rows m, output columns k
  const float* arow = dz + (int64_t)min(i0 + r, M - 1) * lddz + 2 * kq;   // B operand: dz[m = r][n]
  const float* bcol = w + min(j0 + r, K - 1);                            // A operand: w[n][k = j0 + r]
  const int nS = (N + 7) / 8;
  const int nmine = wave < nS ? (nS - wave + 3) / 4 : 0;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float2 fa[FS_PD];
  float fb0[FS_PD], fb1[FS_PD];
  auto load = [&](int s, int it) {
    const int n = (wave + 4 * it) * 8 + 2 * kq;
    const bool in = n < N;                                                // N even
    fa[s] = in ? *reinterpret_cast<const float2*>(arow + (wave + 4 * it) * 8) : make_float2(0.f, 0.f);
    fb0[s] = in ? bcol[(int64_t)n * ldw] : 0.f;
    fb1[s] = in ? bcol[(int64_t)(n + 1) * ldw] : 0.f;
  };
#pragma unroll
  for (int s = 0; s < FS_PD; ++s)
    if (s < nmine) load(s, s);
  for (int it0 = 0; it0 < nmine; it0 += FS_PD) {
#pragma unroll
    for (int s = 0; s < FS_PD; ++s) {
      const int it = it0 + s;
      if (it < nmine) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb0[s], fa[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb1[s], fa[s].y, acc, 0, 0, 0);
        if (it + FS_PD < nmine) load(s, it + FS_PD);
      }
    }
  }
  *reinterpret_cast<f32x4_t*>(&red[wave][r * 16 + 4 * kq]) = acc;      // columns j0 + 4 kq .. +3 of row i0 + r
  __syncthreads();
  if (tid < 64) {
    const int row = tid >> 2, c4 = (tid & 3) * 4;
    const int gi = i0 + row, gj = j0 + c4;
    if (gi < M && gj < K) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (gj + e < K) {
          const int o = row * 16 + c4 + e;
          float v = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
          if (ACT != IB_ACT_NONE) v *= fs_act_bwd<ACT>(aux[(int64_t)gi * ldaux + gj + e]);
          if (addend) v += addend[(int64_t)gi * ldadd + gj + e];
          dx[(int64_t)gi * lddx + gj + e] = v;
        }
      }
    }
  }
}

// dW[N,K] (+)= dz[M,N]^T x[M,K],  dbias[N] (+)= column sums of dz, for a reduction of a few rows (M <= 256): a wave owns a
// 16 (n) x 16 (k) output tile (four tiles per workgroup), both operands are read as they lie -- dz[m][n0 + r], x[m][k0 + r],
// 64-byte row segments -- and the bias gradient rides along as one more MFMA per step against an all-ones operand in the
// workgroups of the first k-tile column.  One launch instead of the generic split kernel + a column-sum launch per layer.
__global__ __launch_bounds__(256) void f32_small_wgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                              const float* __restrict__ x, int64_t ldx, float* __restrict__ dw,
                                                              int64_t lddw, float* __restrict__ dbias, int accumulate, int M,
                                                              int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.y * 16, k0 = (blockIdx.x * 4 + wave) * 16;
  if (k0 >= K) return;
  const bool with_bias = dbias != nullptr && blockIdx.x == 0 && wave == 0;     // wave-uniform
  const float* ap = dz + min(n0 + r, N - 1);          // A[row = n][kred = m]
  const float* bp = x + min(k0 + r, K - 1);           // B[kred = m][col = k]
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
  const int steps = (M + 3) / 4;
  for (int s0 = 0; s0 < steps; s0 += 8) {             // eight steps (16 loads) in flight
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = (s0 + j) * 4 + kq;
      const bool in = m < M;
      a[j] = in ? ap[(int64_t)m * lddz] : 0.f;
      b[j] = in ? bp[(int64_t)m * ldx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s0 + j < steps) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        if (with_bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], 1.f, accb, 0, 0, 0);
      }
    }
  }
  // D[row n = 4 kq + e][col k = r]
  if (k0 + r < K) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 4 * kq + e;
      if (n < N) {
        float* d = dw + (int64_t)n * lddw + k0 + r;
        *d = accumulate ? *d + acc[e] : acc[e];
      }
    }
  }
  if (with_bias && r == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 4 * kq + e;
      if (n < N) dbias[n] = accumulate ? dbias[n] + accb[e] : accb[e];
    }
  }
}

inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
// up to this many 128 x 128 tiles of the generic kernel the 16 x 16 reduction-split tiles win
inline bool few_tiles(int64_t M, int64_t Ncols) {
  static const int off = ib_ab_int("IB_NO_F32_SMALL", 0);
  return !off && ((M + 127) / 128) * ((Ncols + 127) / 128) <= 16 && M <= 1024;
}

}  // namespace

// IB_E_UNSUPPORTED = nothing launched (the caller takes the generic kernel)
int ib_f32_small_fwd_try(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int act, float* y,
                         int64_t ldy, float* z, int64_t ldz, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  if (!few_tiles(M, N) || K < 64 || K % 2 != 0 || ldx % 2 != 0 || ldw % 2 != 0 || !al8(x) || !al8(w)) return IB_E_UNSUPPORTED;
  const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16)), block(256);
#define IB_FS(ACT) hipLaunchKernelGGL((f32_small_fwd_kernel<ACT>), grid, block, 0, s, x, ldx, w, ldw, bias, y, ldy, z, ldz, (int)M, (int)N, (int)K)
  switch (act) {
    case IB_ACT_RELU: IB_FS(IB_ACT_RELU); break;
    case IB_ACT_TANH: IB_FS(IB_ACT_TANH); break;
    case IB_ACT_SIGMOID: IB_FS(IB_ACT_SIGMOID); break;
    case IB_ACT_SILU: IB_FS(IB_ACT_SILU); break;
    case IB_ACT_ELU: IB_FS(IB_ACT_ELU); break;
    default: IB_FS(IB_ACT_NONE); break;
  }
#undef IB_FS
  IB_CHECK_LAUNCH();
  return IB_OK;
}

int ib_f32_small_dgrad_try(const float* dz, int64_t lddz, const float* w, int64_t ldw, int act, const float* aux,
                           int64_t ldaux, const float* addend, int64_t ldadd, float* dx, int64_t lddx, int64_t M, int64_t N,
                           int64_t K, hipStream_t s) {
  // N = reduction length (dz columns), K = output columns
  if (!few_tiles(M, K) || N < 64 || N % 2 != 0 || lddz % 2 != 0 || !al8(dz)) return IB_E_UNSUPPORTED;
  if (act != IB_ACT_NONE && !aux) return IB_E_UNSUPPORTED;
  const dim3 grid((unsigned)((K + 15) / 16), (unsigned)((M + 15) / 16)), block(256);
#define IB_FS(ACT) hipLaunchKernelGGL((f32_small_dgrad_kernel<ACT>), grid, block, 0, s, dz, lddz, w, ldw, aux, ldaux, addend, ldadd, dx, lddx, (int)M, (int)N, (int)K)
  switch (act) {
    case IB_ACT_RELU: IB_FS(IB_ACT_RELU); break;
    case IB_ACT_TANH: IB_FS(IB_ACT_TANH); break;
    case IB_ACT_SIGMOID: IB_FS(IB_ACT_SIGMOID); break;
    case IB_ACT_SILU: IB_FS(IB_ACT_SILU); break;
    case IB_ACT_ELU: IB_FS(IB_ACT_ELU); break;
    default: IB_FS(IB_ACT_NONE); break;
  }
#undef IB_FS
  IB_CHECK_LAUNCH();
  return IB_OK;
}

int ib_f32_small_wgrad_bias_try(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, int64_t lddw, float* dbias,
                                int accumulate, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  static const int off = ib_ab_int("IB_NO_F32_SMALL", 0);
  if (off || M > 256) return IB_E_UNSUPPORTED;
  const dim3 grid((unsigned)((K + 63) / 64), (unsigned)((N + 15) / 16)), block(256);
  hipLaunchKernelGGL(f32_small_wgrad_kernel, grid, block, 0, s, dz, lddz, x, ldx, dw, lddw, dbias, accumulate, (int)M, (int)N, (int)K);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
