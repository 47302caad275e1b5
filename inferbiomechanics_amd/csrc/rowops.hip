// Row-wise HBM-bound kernels for gfx950: LayerNorm (+residual, +pre-activation) fwd/bwd with
// wave-shuffle reductions, segmented column sums (bias / embedding grads), input packing, casts.
// One wave (64 lanes) owns one row; a lane holds CH chunks of 4 consecutive elements in registers
// (8- or 16-byte loads), so every tensor crosses HBM exactly once per kernel.
#include "ib_common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float act_bwd_pre(int act, float z) {
  // derivative of act at pre-activation z
  switch (act) {
    case IB_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case IB_ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
    case IB_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    case IB_ACT_SILU: { float s = 1.f / (1.f + expf(-z)); return s * (1.f + z * (1.f - s)); }
    default: return 1.f;
  }
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, int nv, bool vec, float (&v)[4]) {
  if (nv == 4 && vec) {
    if constexpr (sizeof(T) == 2) {
      bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(p);
      v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    } else {
      float4 t = *reinterpret_cast<const float4*>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (e < nv) ? ib_to_f32(p[e]) : 0.f;
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, int nv, bool vec, const float (&v)[4]) {
  if (nv == 4 && vec) {
    if constexpr (sizeof(T) == 2) {
      bf16x4_t o;
      o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
      *reinterpret_cast<bf16x4_t*>(p) = o;
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nv) p[e] = ib_from_f32<T>(v[e]);
  }
}

// ---------------------------------------------------------------- LayerNorm
// LPR lanes share a row (16 for N <= 512: four rows in flight per wave, 64 for wider rows); a lane owns
// CH chunks of 8 consecutive elements (16-byte bf16 / 2 x 16-byte fp32 accesses) kept in registers, so x,
// res, dy are read once and y / dx written once.  Row reductions are xor-shuffles inside the LPR-lane
// group.  ACT is compile-time (IB_ACT_NONE / IB_ACT_SILU; -1 = runtime switch for the rare others).
// sigmoid: accurate libm form for fp32 storage (parity mode), hardware exp + rcp for bf16 storage (their
// error is orders of magnitude below bf16 resolution; the accurate form made LN-backward VALU-bound)
template <typename T> __device__ __forceinline__ float ln_sigmoid(float v) {
  if constexpr (sizeof(T) == 2) return __frcp_rn(1.f + __expf(-v));
  else return 1.f / (1.f + expf(-v));
}
template <typename T, int ACT> __device__ __forceinline__ float ln_act(int act, float v) {
  if constexpr (ACT == IB_ACT_NONE) return v;
  else if constexpr (ACT == IB_ACT_SILU) return v * ln_sigmoid<T>(v);
  else return ib_act_fwd(act, v);
}
// activation value AND derivative from one sigmoid evaluation
template <typename T, int ACT> __device__ __forceinline__ void ln_act_both(int act, float z, float& h, float& dh) {
  if constexpr (ACT == IB_ACT_NONE) { h = z; dh = 1.f; }
  else if constexpr (ACT == IB_ACT_SILU) { const float s = ln_sigmoid<T>(z); h = z * s; dh = s * (1.f + z * (1.f - s)); }
  else { h = ib_act_fwd(act, z); dh = act_bwd_pre(act, z); }
}

template <typename T>
__device__ __forceinline__ void load8(const T* p, int nv, bool vec, float (&v)[8]) {
  if (nv == 8 && vec) {
    if constexpr (sizeof(T) == 2) {
      bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
      float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e < nv) ? ib_to_f32(p[e]) : 0.f;
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, int nv, bool vec, const float (&v)[8]) {
  if (nv == 8 && vec) {
    if constexpr (sizeof(T) == 2) {
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8_t*>(p) = o;
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < nv) p[e] = ib_from_f32<T>(v[e]);
  }
}
// sum over the LPR lanes that share a matrix row (every one of them gets the total).  The first four steps stay inside a
// 16-lane DPP row and are DPP adds (no LDS crossbar: a ds_bpermute shuffle costs ~100 cycles of dependent latency each,
// and the backward needs two such sums per matrix row); rows are combined by one shuffle (LPR = 32) or by reading the
// four row totals out as scalars (LPR = 64).
template <int CTRL> __device__ __forceinline__ float gs_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
  static_assert(LPR == 16 || LPR == 32 || LPR == 64, "lanes per row");
  v += gs_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += gs_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += gs_dpp<0x141>(v);    // row_half_mirror
  v += gs_dpp<0x140>(v);    // row_mirror
  if constexpr (LPR == 32) v += __shfl_xor(v, 16, 64);
  if constexpr (LPR == 64) {
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    v = ((r0 + r1) + r2) + r3;
  }
  return v;
}

template <typename T, int LPR, int CH, int ACT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, int act,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int M, int N, float eps, int vec, int vecp,
                                                            const T* __restrict__ add_div, int64_t ld_add, int seg) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const float invN = 1.f / (float)N;
  for (int rb = (blockIdx.x * 4 + wave) * RPW; rb < M; rb += gridDim.x * 4 * RPW) {
    const int row = rb + sub;
    const bool live = row < M;
    const int64_t ro = (int64_t)(live ? row : M - 1) * N;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * LPR + l) * 8;
      const int nv = max(0, min(8, N - col));
      load8<T>(x + ro + col, nv, vec, v[c]);
      if (add_div) {        // per-window row added BEFORE the activation (the diffusion time embedding)
        float a8[8];
        load8<T>(add_div + (int64_t)((live ? row : M - 1) / seg) * ld_add + col, nv, vec, a8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] += a8[e];
      }
      if constexpr (ACT != IB_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = (e < nv) ? ln_act<T, ACT>(act, v[c][e]) : 0.f;
      }
      if (res) {
        float r8[8];
        load8<T>(res + ro + col, nv, vec, r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] += r8[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    }
    const float mu = group_sum<LPR>(s) * invN;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * LPR + l) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (col + e < N) ? v[c][e] - mu : 0.f;
        q += d * d;
      }
    }
    const float rs = 1.f / sqrtf(group_sum<LPR>(q) * invN + eps);
    if (live && l == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
    if (live) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 8;
        const int nv = max(0, min(8, N - col));
        if (nv > 0) {
          float gm[8], bt[8], o[8];
          load8<float>(gamma + col, nv, vecp, gm);
          load8<float>(beta + col, nv, vecp, bt);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mu) * rs * gm[e] + bt[e];
          store8<T>(y + ro + col, nv, vec, o);
        }
      }
    }
  }
}

// partial[blockIdx][0..N) = sum over this block's rows of dy*xhat ; partial[gridDim + blockIdx] = sum dy
constexpr int LN_BWD_WPB = 8;   // waves per block in the backward: 256 blocks x 8 waves cover the chip with <= 256 partial rows

template <typename T, int LPR, int CH, int ACT>
__global__ __launch_bounds__(LN_BWD_WPB * 64) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ res, int act,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, T* __restrict__ dx,
                                                            T* __restrict__ dres, float* __restrict__ partial,
                                                            int M, int N, int vec, int vecp,
                                                            const T* __restrict__ add_div, int64_t ld_add, int seg) {
  constexpr int RPW = 64 / LPR;
  constexpr int NPAD = CH * LPR * 8;
  constexpr int WPB = LN_BWD_WPB;
  __shared__ float red[WPB][2][NPAD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const float invN = 1.f / (float)N;
  float ag[CH][8], ab[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }
  for (int rb = (blockIdx.x * WPB + wave) * RPW; rb < M; rb += gridDim.x * WPB * RPW) {
    const int row = rb + sub;
    const bool live = row < M;
    const int64_t ro = (int64_t)(live ? row : M - 1) * N;
    const float mu = mean[live ? row : M - 1], rs = rstd[live ? row : M - 1];
    float xh[CH][8], g[CH][8], zr[CH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * LPR + l) * 8;
      const int nv = live ? max(0, min(8, N - col)) : 0;
      float xv[8], dyv[8], gm[8];
      load8<T>(x + ro + col, nv, vec, xv);
      load8<T>(dy + ro + col, nv, vec, dyv);
      load8<float>(gamma + col, max(0, min(8, N - col)), vecp, gm);   // L1/L2 resident; not worth 8 registers per chunk
      if (add_div) {
        float a8[8];
        load8<T>(add_div + (int64_t)((live ? row : M - 1) / seg) * ld_add + col, nv, vec, a8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] += a8[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float h, dh;
        ln_act_both<T, ACT>(act, xv[e], h, dh);
        xv[e] = (e < nv) ? h : 0.f;
        zr[c][e] = dh;                 // activation derivative, reused for dx below
      }
      if (res) {
        float r8[8];
        load8<T>(res + ro + col, nv, vec, r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] += r8[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = e < nv;
        const float h = ok ? (xv[e] - mu) * rs : 0.f;
        const float gg = ok ? dyv[e] * gm[e] : 0.f;
        xh[c][e] = h; g[c][e] = gg;
        s1 += gg; s2 += gg * h;
        ag[c][e] += ok ? dyv[e] * h : 0.f;
        ab[c][e] += ok ? dyv[e] : 0.f;
      }
    }
    const float c1 = group_sum<LPR>(s1) * invN;
    const float c2 = group_sum<LPR>(s2) * invN;
    if (live) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 8;
        const int nv = max(0, min(8, N - col));
        if (nv > 0) {
          float dv[8], dxo[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dv[e] = (g[c][e] - c1 - xh[c][e] * c2) * rs;
            dxo[e] = dv[e] * zr[c][e];
          }
          store8<T>(dx + ro + col, nv, vec, dxo);
          if (dres) store8<T>(dres + ro + col, nv, vec, dv);
        }
      }
    }
  }
  // rows of one wave that share columns (same l, different sub): fixed-order shuffle sum, then LDS across waves
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = ag[c][e], b = ab[c][e];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (sub == 0) {
        red[wave][0][(c * LPR + l) * 8 + e] = a;
        red[wave][1][(c * LPR + l) * 8 + e] = b;
      }
    }
  __syncthreads();
  for (int col = threadIdx.x; col < N; col += blockDim.x) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) { sg += red[w][0][col]; sb += red[w][1][col]; }
    partial[(int64_t)blockIdx.x * N + col] = sg;
    partial[((int64_t)gridDim.x + blockIdx.x) * N + col] = sb;
  }
}

// ---- LayerNorm backward, the transformer denoiser's shape (bf16, N = 512, whole 16-byte pieces, no activation, no
// per-window addend): one row per wave-instruction (64 lanes x 8 columns), and EVERY row of a wave requested before the
// first is used.  The generic kernel above walks its rows one at a time (load -> two row reductions -> store, 6 times per
// wave at M = 12800): replayed back to back its inputs come from L2 / MALL and it takes 13.5 us, but inside the training
// step (inputs cold: dy from the previous GEMM, x and the residual from the forward pass) every trip pays an HBM round trip
// and it takes 30 us for 52 MB = 1.8 TB/s.  Here a 16-wave workgroup per CU keeps 4 rows x 3 KiB per wave in flight.
constexpr int LNF_WAVES = 16, LNF_ROWS = 4, LNF_N = 512;
__global__ __launch_bounds__(LNF_WAVES * 64) void layernorm_bwd512_kernel(const bf16_t* __restrict__ dy,
                                                                          const bf16_t* __restrict__ x,
                                                                          const bf16_t* __restrict__ res,
                                                                          const float* __restrict__ gamma,
                                                                          const float* __restrict__ mean,
                                                                          const float* __restrict__ rstd,
                                                                          bf16_t* __restrict__ dx, float* __restrict__ partial,
                                                                          int M) {
  __shared__ float red[LNF_WAVES][2][LNF_N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane * 8;
  const int wstride = gridDim.x * LNF_WAVES;
  float gm[8], ag[8], ab[8];
  {
    const float4 a = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(gamma + col + 4);
    gm[0] = a.x; gm[1] = a.y; gm[2] = a.z; gm[3] = a.w; gm[4] = b.x; gm[5] = b.y; gm[6] = b.z; gm[7] = b.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
  for (int rb = blockIdx.x * LNF_WAVES + wave; rb < M; rb += wstride * LNF_ROWS) {
    bf16x8_t xv[LNF_ROWS], dv[LNF_ROWS], rv[LNF_ROWS];
    float mu[LNF_ROWS], rs[LNF_ROWS];
#pragma unroll
    for (int j = 0; j < LNF_ROWS; ++j) {              // every request of this trip first
      const int row = min(rb + j * wstride, M - 1);
      const int64_t ro = (int64_t)row * LNF_N + col;
      xv[j] = *reinterpret_cast<const bf16x8_t*>(x + ro);
      dv[j] = *reinterpret_cast<const bf16x8_t*>(dy + ro);
      if (res) rv[j] = *reinterpret_cast<const bf16x8_t*>(res + ro);
      mu[j] = mean[row];
      rs[j] = rstd[row];
    }
#pragma unroll
    for (int j = 0; j < LNF_ROWS; ++j) {
      const int row = rb + j * wstride;
      const bool live = row < M;                       // wave-uniform
      float h[8], gg[8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xe = (float)xv[j][e];
        if (res) xe += (float)rv[j][e];
        const float d = live ? (float)dv[j][e] : 0.f;
        h[e] = (xe - mu[j]) * rs[j];
        gg[e] = d * gm[e];
        s1 += gg[e]; s2 += gg[e] * h[e];
        ag[e] += d * h[e];
        ab[e] += d;
      }
      const float c1 = group_sum<64>(s1) * (1.f / LNF_N);
      const float c2 = group_sum<64>(s2) * (1.f / LNF_N);
      if (live) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((gg[e] - c1 - h[e] * c2) * rs[j]);
        *reinterpret_cast<bf16x8_t*>(dx + (int64_t)row * LNF_N + col) = o;
      }
    }
  }
  {
    float4* r0 = reinterpret_cast<float4*>(&red[wave][0][col]);
    float4* r1 = reinterpret_cast<float4*>(&red[wave][1][col]);
    r0[0] = make_float4(ag[0], ag[1], ag[2], ag[3]); r0[1] = make_float4(ag[4], ag[5], ag[6], ag[7]);
    r1[0] = make_float4(ab[0], ab[1], ab[2], ab[3]); r1[1] = make_float4(ab[4], ab[5], ab[6], ab[7]);
  }
  __syncthreads();
  if (threadIdx.x < LNF_N) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < LNF_WAVES; ++w) { sg += red[w][0][threadIdx.x]; sb += red[w][1][threadIdx.x]; }
    partial[(int64_t)blockIdx.x * LNF_N + threadIdx.x] = sg;
    partial[((int64_t)gridDim.x + blockIdx.x) * LNF_N + threadIdx.x] = sb;
  }
}

// LayerNorm forward at the same shape: y = LN(x + res), four rows per wave requested together (the generic kernel has one
// row per wave in flight; in the step its 40 MB take 12.4 us = 3.2 TB/s)
__global__ __launch_bounds__(256) void layernorm_fwd512_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               bf16_t* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int M, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane * 8;
  const int rb = (blockIdx.x * 4 + wave) * LNF_ROWS;
  if (rb >= M) return;
  bf16x8_t xv[LNF_ROWS], rv[LNF_ROWS];
#pragma unroll
  for (int j = 0; j < LNF_ROWS; ++j) {
    const int64_t ro = (int64_t)min(rb + j, M - 1) * LNF_N + col;
    xv[j] = *reinterpret_cast<const bf16x8_t*>(x + ro);
    if (res) rv[j] = *reinterpret_cast<const bf16x8_t*>(res + ro);
  }
  float gm[8], bt[8];
  {
    const float4 a = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(gamma + col + 4);
    const float4 c = *reinterpret_cast<const float4*>(beta + col), d = *reinterpret_cast<const float4*>(beta + col + 4);
    gm[0] = a.x; gm[1] = a.y; gm[2] = a.z; gm[3] = a.w; gm[4] = b.x; gm[5] = b.y; gm[6] = b.z; gm[7] = b.w;
    bt[0] = c.x; bt[1] = c.y; bt[2] = c.z; bt[3] = c.w; bt[4] = d.x; bt[5] = d.y; bt[6] = d.z; bt[7] = d.w;
  }
#pragma unroll
  for (int j = 0; j < LNF_ROWS; ++j) {
    const int row = rb + j;
    float v[8];
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (float)xv[j][e];
      if (res) v[e] += (float)rv[j][e];
      sm += v[e];
    }
    const float mu = group_sum<64>(sm) * (1.f / LNF_N);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q += d * d; }
    const float rs = 1.f / sqrtf(group_sum<64>(q) * (1.f / LNF_N) + eps);       // the generic kernel's two-pass arithmetic
    if (row < M) {
      if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
      }
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[e] - mu) * rs * gm[e] + bt[e]);
      *reinterpret_cast<bf16x8_t*>(y + (int64_t)row * LNF_N + col) = o;
    }
  }
}

// ---------------------------------------------------------------- segmented column sum
// rows of segment s: m = s*a + r*b, r in [0, cnt).  A 256-thread block = (256/RL) column groups of 4
// consecutive columns x RL row lanes; the row lanes' partial sums are combined through LDS in a fixed
// order.  RL = 4 for short segments, 16 for long ones (independent loads in flight instead of a serial chain).
template <typename T, int RL>
__global__ __launch_bounds__(256) void segment_colsum_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                             int64_t ldo, int M, int N, int seg, int mode, int accumulate,
                                                             int vec, float* __restrict__ out1, bf16_t* __restrict__ out_lp,
                                                             int64_t ld_lp) {
  constexpr int CG = 256 / RL;
  __shared__ float red[RL][CG][4];
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int s = blockIdx.y;
  const int col = (blockIdx.x * CG + cg) * 4;
  const int nv = max(0, min(4, N - col));
  int64_t a, b; int cnt;
  if (mode == 0) { a = seg; b = 1; cnt = min(seg, M - s * seg); }
  else { a = 1; b = seg; cnt = (M - s + seg - 1) / seg; }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (nv > 0) {
    int r = rl;
    for (; r + 3 * RL < cnt; r += 4 * RL) {      // 4 independent loads in flight per lane
      float v0[4], v1[4], v2[4], v3[4];
      load4<T>(x + ((int64_t)s * a + (int64_t)r * b) * ldx + col, nv, vec, v0);
      load4<T>(x + ((int64_t)s * a + (int64_t)(r + RL) * b) * ldx + col, nv, vec, v1);
      load4<T>(x + ((int64_t)s * a + (int64_t)(r + 2 * RL) * b) * ldx + col, nv, vec, v2);
      load4<T>(x + ((int64_t)s * a + (int64_t)(r + 3 * RL) * b) * ldx + col, nv, vec, v3);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; r < cnt; r += RL) {
      float v[4];
      load4<T>(x + ((int64_t)s * a + (int64_t)r * b) * ldx + col, nv, vec, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cg][e] = acc[e];
  __syncthreads();
  if (rl == 0 && nv > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < nv) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < RL; ++w) t += red[w][cg][e];
        float* o = (out1 && s == 1) ? out1 + col + e : out + (int64_t)s * ldo + col + e;   // out1: segment 1 elsewhere
        const float r = accumulate ? *o + t : t;
        *o = r;
        if (out_lp) out_lp[(int64_t)s * ld_lp + col + e] = (bf16_t)r;   // bf16 copy = next GEMM's operand (no cast launch)
      }
    }
  }
}

template <typename T>
void launch_segment_colsum(const T* x, int64_t ldx, float* out, int64_t ldo, int M, int N, int seg, int mode,
                           int accumulate, int vec, int nseg, int rows_per_seg, hipStream_t s, float* out1 = nullptr,
                           bf16_t* out_lp = nullptr, int64_t ld_lp = 0) {
  if (rows_per_seg > 32) {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)nseg);
    hipLaunchKernelGGL((segment_colsum_kernel<T, 16>), grid, dim3(256), 0, s, x, ldx, out, ldo, M, N, seg, mode,
                       accumulate, vec, out1, out_lp, ld_lp);
  } else {
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)nseg);
    hipLaunchKernelGGL((segment_colsum_kernel<T, 4>), grid, dim3(256), 0, s, x, ldx, out, ldo, M, N, seg, mode,
                       accumulate, vec, out1, out_lp, ld_lp);
  }
}

// ---------------------------------------------------------------- input packing / casts
struct ConcatArgs {
  const float* in[16];
  int width[16];
  int offset[16];
  int nkeys;
  int total;
};
template <typename T>
__global__ void concat_keys_kernel(ConcatArgs a, T* __restrict__ out, int64_t rows) {
  const int64_t n = rows * a.total;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / a.total;
    const int col = (int)(i % a.total);
    int k = 0;
#pragma unroll
    for (int j = 1; j < 16; ++j)
      if (j < a.nkeys && col >= a.offset[j]) k = j;
    out[i] = ib_from_f32<T>(a.in[k][row * a.width[k] + (col - a.offset[k])]);
  }
}

template <typename S, typename D>
__global__ void cast2d_kernel(const S* __restrict__ src, int64_t lds, D* __restrict__ dst, int64_t ldd, int64_t rows,
                              int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    dst[r * ldd + c] = ib_from_f32<D>(ib_to_f32(src[r * lds + c]));
  }
}

template <typename T, int LPR, int CH>
int launch_ln_fwd(const void* x, const void* res, int act, const float* gamma, const float* beta, void* y, float* mean,
                  float* rstd, int64_t M, int64_t N, float eps, int vec, int vecp, const void* add_div, int64_t ld_add,
                  int seg, hipStream_t s) {
  constexpr int RPB = 4 * (64 / LPR);
  const int grid = ib_grid_1d(M, RPB);
#define IB_LN_FWD(ACT)                                                                                              \
  hipLaunchKernelGGL((layernorm_fwd_kernel<T, LPR, CH, ACT>), dim3(grid), dim3(256), 0, s, (const T*)x, (const T*)res, \
                     act, gamma, beta, (T*)y, mean, rstd, (int)M, (int)N, eps, vec, vecp, (const T*)add_div, ld_add, seg)
  if (act == IB_ACT_NONE) IB_LN_FWD(IB_ACT_NONE);
  else if (act == IB_ACT_SILU) IB_LN_FWD(IB_ACT_SILU);
  else IB_LN_FWD(-1);
#undef IB_LN_FWD
  IB_CHECK_LAUNCH();
  return IB_OK;
}

int ln_bwd_parts(int64_t M) {
  int64_t g = (M + 15) / 16;
  // one 8-wave block per CU; the fixed-order reduction over the partial rows stays short.  [12800, 512] bf16: 128 blocks
  // 31 us, 256: 24.5, 512: 27.8 (the second-stage sum grows with the partial rows); the row loop is VALU-bound
  // (a next-row prefetch changed nothing), N = 512 as one row per wave (64 lanes x 8) is 11 % faster than two
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T, int LPR, int CH>
int launch_ln_bwd(const void* dy, const void* x, const void* res, int act, const float* gamma, const float* mean,
                  const float* rstd, void* dx, void* dres, float* partial, int64_t M, int64_t N, int vec, int vecp,
                  int parts, const void* add_div, int64_t ld_add, int seg, hipStream_t s) {
#define IB_LN_BWD(ACT)                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, LPR, CH, ACT>), dim3(parts), dim3(LN_BWD_WPB * 64), 0, s, (const T*)dy, (const T*)x, \
                     (const T*)res, act, gamma, mean, rstd, (T*)dx, (T*)dres, partial, (int)M, (int)N, vec, vecp,           \
                     (const T*)add_div, ld_add, seg)
  if (act == IB_ACT_NONE) IB_LN_BWD(IB_ACT_NONE);
  else if (act == IB_ACT_SILU) IB_LN_BWD(IB_ACT_SILU);
  else IB_LN_BWD(-1);
#undef IB_LN_BWD
  IB_CHECK_LAUNCH();
  return IB_OK;
}

inline bool al(const void* p, size_t a) { return !p || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

// backward keeps 5 per-lane arrays of CH*8 floats: use more lanes per row (fewer elements per lane) so the kernel
// stays near 128 VGPRs (>= 3 waves per SIMD); at 256 VGPRs it ran at 1 wave per SIMD and 4x off the HBM roofline
#define LN_DISPATCH_BWD(FN, T, ...)                                                          \
  (N <= 128 ? FN<T, 16, 1>(__VA_ARGS__) : N <= 256 ? FN<T, 16, 2>(__VA_ARGS__)              \
   : N <= 512 ? FN<T, 64, 1>(__VA_ARGS__) : N <= 1024 ? FN<T, 64, 2>(__VA_ARGS__) : FN<T, 64, 4>(__VA_ARGS__))
#define LN_DISPATCH(FN, T, ...)                                                              \
  (N <= 128 ? FN<T, 16, 1>(__VA_ARGS__) : N <= 256 ? FN<T, 16, 2>(__VA_ARGS__)              \
   : N <= 512 ? FN<T, 16, 4>(__VA_ARGS__) : N <= 1024 ? FN<T, 64, 2>(__VA_ARGS__) : FN<T, 64, 4>(__VA_ARGS__))

extern "C" int ib_layernorm_fwd(const void* x, const void* res, int act, const float* gamma, const float* beta, void* y,
                                float* mean, float* rstd, const void* add_div, int64_t ld_add, int64_t seg, int64_t M,
                                int64_t N, float eps, int dtype, ib_stream_t stream) {
  if (!x || !gamma || !beta || !y || M <= 0 || N <= 0) return IB_E_ARG;
  if (add_div && (seg <= 0 || ld_add < N)) return IB_E_ARG;
  if (N > 2048) return IB_E_UNSUPPORTED;
  hipStream_t s = ib_s(stream);
  const int vecp = (N % 4 == 0) && al(gamma, 16) && al(beta, 16);
  if (dtype == IB_F32) {
    const int vec = (N % 4 == 0) && al(x, 16) && al(res, 16) && al(y, 16) && al(add_div, 16) && (ld_add % 4 == 0);
    return LN_DISPATCH(launch_ln_fwd, float, x, res, act, gamma, beta, y, mean, rstd, M, N, eps, vec, vecp, add_div,
                       ld_add, (int)seg, s);
  }
  if (dtype == IB_BF16) {
    const int vec = (N % 8 == 0) && al(x, 16) && al(res, 16) && al(y, 16) && al(add_div, 16) && (ld_add % 8 == 0);
    static const bool no_fast = ib_ab_set("IB_NO_LN_FAST");
    if (!no_fast && N == LNF_N && vec && vecp && act == IB_ACT_NONE && !add_div && M >= 4096) {
      hipLaunchKernelGGL(layernorm_fwd512_kernel, dim3((unsigned)((M + 4 * LNF_ROWS - 1) / (4 * LNF_ROWS))), dim3(256), 0, s,
                         (const bf16_t*)x, (const bf16_t*)res, gamma, beta, (bf16_t*)y, mean, rstd, (int)M, eps);
      IB_CHECK_LAUNCH();
      return IB_OK;
    }
    return LN_DISPATCH(launch_ln_fwd, bf16_t, x, res, act, gamma, beta, y, mean, rstd, M, N, eps, vec, vecp, add_div,
                       ld_add, (int)seg, s);
  }
  return IB_E_DTYPE;
}

extern "C" size_t ib_layernorm_bwd_workspace(int64_t M, int64_t N) {
  return (size_t)2 * ln_bwd_parts(M) * (size_t)N * sizeof(float);
}

extern "C" int ib_layernorm_bwd(const void* dy, const void* x, const void* res, int act, const float* gamma,
                                const float* mean, const float* rstd, void* dx, void* dres, float* dgamma,
                                float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                                const void* add_div, int64_t ld_add, int64_t seg, int64_t M, int64_t N, int dtype,
                                ib_stream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || M <= 0 || N <= 0) return IB_E_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return IB_E_ARG;   // both NULL = leave the partials in the workspace
  if (add_div && (seg <= 0 || ld_add < N)) return IB_E_ARG;
  if (N > 2048) return IB_E_UNSUPPORTED;
  const int parts = ln_bwd_parts(M);
  if (!workspace || workspace_bytes < (size_t)2 * parts * N * sizeof(float)) return IB_E_WORKSPACE;
  hipStream_t s = ib_s(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  int rc;
  const int vecp = (N % 4 == 0) && al(gamma, 16);
  if (dtype == IB_F32) {
    const int vec = (N % 4 == 0) && al(x, 16) && al(res, 16) && al(dy, 16) && al(dx, 16) && al(dres, 16) &&
                    al(add_div, 16) && (ld_add % 4 == 0);
    rc = LN_DISPATCH_BWD(launch_ln_bwd, float, dy, x, res, act, gamma, mean, rstd, dx, dres, partial, M, N, vec, vecp,
                     parts, add_div, ld_add, (int)seg, s);
  } else if (dtype == IB_BF16) {
    const int vec = (N % 8 == 0) && al(x, 16) && al(res, 16) && al(dy, 16) && al(dx, 16) && al(dres, 16) &&
                    al(add_div, 16) && (ld_add % 8 == 0);
    static const bool no_fast = ib_ab_set("IB_NO_LN_FAST");
    if (!no_fast && N == LNF_N && vec && vecp && act == IB_ACT_NONE && !add_div && !dres && M >= 4096) {
      hipLaunchKernelGGL(layernorm_bwd512_kernel, dim3(parts), dim3(LNF_WAVES * 64), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                         (const bf16_t*)res, gamma, mean, rstd, (bf16_t*)dx, partial, (int)M);
      IB_CHECK_LAUNCH();
      rc = IB_OK;
    } else
    rc = LN_DISPATCH_BWD(launch_ln_bwd, bf16_t, dy, x, res, act, gamma, mean, rstd, dx, dres, partial, M, N, vec, vecp,
                     parts, add_div, ld_add, (int)seg, s);
  } else {
    return IB_E_DTYPE;
  }
  if (rc != IB_OK) return rc;
  if (!dgamma) return IB_OK;        // deferred: ib_layernorm_bwd_reduce() finishes the parameter gradients later
  // fixed-order sums of the per-block partials in ONE launch: [2*parts, N] as two segments of `parts` rows,
  // segment 0 -> dgamma, segment 1 -> dbeta
  const int pvec = (N % 4 == 0);
  launch_segment_colsum<float>(partial, N, dgamma, N, 2 * parts, (int)N, parts, 0, accumulate, pvec, 2, parts, s, dbeta);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_layernorm_bwd_reduce(const void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta,
                                       int accumulate, int64_t M, int64_t N, ib_stream_t stream) {
  if (!workspace || !dgamma || !dbeta || M <= 0 || N <= 0) return IB_E_ARG;
  const int parts = ln_bwd_parts(M);
  if (workspace_bytes < (size_t)2 * parts * N * sizeof(float)) return IB_E_WORKSPACE;
  const float* partial = reinterpret_cast<const float*>(workspace);
  launch_segment_colsum<float>(partial, N, dgamma, N, 2 * parts, (int)N, parts, 0, accumulate, (N % 4 == 0), 2, parts,
                               ib_s(stream), dbeta);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_segment_colsum(const void* x, int64_t ldx, float* out, int64_t ldo, void* out_bf16, int64_t ld_bf16,
                                 int64_t M, int64_t N, int64_t seg, int mode, int accumulate, int dtype,
                                 ib_stream_t stream) {
  if (!x || !out || M <= 0 || N <= 0 || seg <= 0 || ldx < N || ldo < N || (mode != 0 && mode != 1)) return IB_E_ARG;
  if (out_bf16 && ld_bf16 < N) return IB_E_ARG;
  const int64_t nseg = (mode == 0) ? (M + seg - 1) / seg : (seg < M ? seg : M);
  if (nseg > 65535) return IB_E_UNSUPPORTED;
  hipStream_t s = ib_s(stream);
  const int rps = (int)((mode == 0) ? (seg < M ? seg : M) : (M + seg - 1) / seg);
  if (dtype == IB_F32) {
    const int vec = (ldx % 4 == 0) && al(x, 16);
    launch_segment_colsum<float>((const float*)x, ldx, out, ldo, (int)M, (int)N, (int)seg, mode, accumulate, vec,
                                 (int)nseg, rps, s, nullptr, (bf16_t*)out_bf16, ld_bf16);
  } else if (dtype == IB_BF16) {
    const int vec = (ldx % 4 == 0) && al(x, 8);
    launch_segment_colsum<bf16_t>((const bf16_t*)x, ldx, out, ldo, (int)M, (int)N, (int)seg, mode, accumulate, vec,
                                  (int)nseg, rps, s, nullptr, (bf16_t*)out_bf16, ld_bf16);
  } else {
    return IB_E_DTYPE;
  }
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_concat_keys(const float* const* inputs, const int32_t* widths, int32_t nkeys, void* out, int64_t rows,
                              int dtype_out, ib_stream_t stream) {
  if (!inputs || !widths || !out || nkeys <= 0 || nkeys > 16 || rows <= 0) return IB_E_ARG;
  ConcatArgs a{};
  a.nkeys = nkeys;
  int off = 0;
  for (int k = 0; k < nkeys; ++k) {
    if (!inputs[k] || widths[k] <= 0) return IB_E_ARG;
    a.in[k] = inputs[k]; a.width[k] = widths[k]; a.offset[k] = off; off += widths[k];
  }
  a.total = off;
  const int grid = ib_grid_1d(rows * off, 256);
  hipStream_t s = ib_s(stream);
  if (dtype_out == IB_F32) hipLaunchKernelGGL((concat_keys_kernel<float>), dim3(grid), dim3(256), 0, s, a, (float*)out, rows);
  else if (dtype_out == IB_BF16) hipLaunchKernelGGL((concat_keys_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, a, (bf16_t*)out, rows);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_cast2d(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype, int64_t rows,
                         int64_t cols, ib_stream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return IB_E_ARG;
  const int grid = ib_grid_1d(rows * cols, 256);
  hipStream_t s = ib_s(stream);
  if (src_dtype == IB_F32 && dst_dtype == IB_BF16)
    hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else if (src_dtype == IB_BF16 && dst_dtype == IB_F32)
    hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, lds, (float*)dst, ldd, rows, cols);
  else if (src_dtype == IB_F32 && dst_dtype == IB_F32)
    hipLaunchKernelGGL((cast2d_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, lds, (float*)dst, ldd, rows, cols);
  else if (src_dtype == IB_BF16 && dst_dtype == IB_BF16)
    hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else
    return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, ib_stream_t stream) {
  return ib_cast2d(src, n, src_dtype, dst, n, dst_dtype, 1, n, stream);
}

// ---- tiny matrix products ------------------------------------------------------------------------------------------
// C[M,N] (+)= sum_k A(m,k) B(k,n) with every operand addressed by two strides (so transposes and column slices of a
// weight matrix are views) and its own storage type.  For the frame-embedding projection of the transformer denoiser and
// its two gradients (TransformerBaseline.py:41-48,119-126 pattern: [T=50, 30] x [30, 512]): far below one MFMA tile
// pair, and the 30-column slice of in_proj.weight is not 16-byte aligned, so the tiled kernels took 20 - 58 us of
// scalar loads for 1.5 MFLOP.  K >= 64: one wave per output element (lanes stride the reduction, butterfly sum);
// shorter reductions: one thread per output element.  fp32 accumulation in a fixed order (bitwise reproducible).
namespace {
__device__ __forceinline__ float tiny_ld(const void* p, int dtype, int64_t i) {
  return dtype == IB_F32 ? static_cast<const float*>(p)[i] : static_cast<float>(static_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void tiny_st(void* p, int dtype, int64_t i, float v, int accumulate) {
  if (dtype == IB_F32) {
    float* q = static_cast<float*>(p) + i;
    *q = accumulate ? *q + v : v;
  } else {
    bf16_t* q = static_cast<bf16_t*>(p) + i;
    *q = static_cast<bf16_t>(accumulate ? static_cast<float>(*q) + v : v);
  }
}
template <bool WAVE>
__global__ __launch_bounds__(256) void tiny_matmul_kernel(const void* __restrict__ A, int ad, int64_t sam, int64_t sak,
                                                          const void* __restrict__ B, int bd, int64_t sbk, int64_t sbn,
                                                          void* __restrict__ C, int cd, int64_t ldc, int accumulate, int M,
                                                          int N, int K) {
  const int total = M * N;
  if (WAVE) {
    const int lane = threadIdx.x & 63;
    for (int o = blockIdx.x * 4 + (threadIdx.x >> 6); o < total; o += gridDim.x * 4) {
      const int m = o / N, n = o - m * N;
      float acc = 0.f;
      for (int k0 = lane; k0 < K; k0 += 64 * 8) {     // 8 operand pairs requested together, added in k order (see below)
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = min(k0 + 64 * e, K - 1);
          a[e] = tiny_ld(A, ad, m * sam + k * sak);
          b[e] = tiny_ld(B, bd, k * sbk + n * sbn);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + 64 * e < K) acc += a[e] * b[e];
      }
      acc = ib_wave_sum(acc);
      if (lane == 0) tiny_st(C, cd, (int64_t)m * ldc + n, acc, accumulate);
    }
  } else {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
      const int m = o / N, n = o - m * N;
      // a rolled loop is one memory round trip per k in sequence (K = 30: 31 us for the [50, 30] x [30, 512] frame-embedding
      // projection at the head of every step): batches of 8 operand pairs are requested together, the sum keeps its order
      float acc = 0.f;
      for (int k0 = 0; k0 < K; k0 += 8) {
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = min(k0 + e, K - 1);
          a[e] = tiny_ld(A, ad, m * sam + k * sak);
          b[e] = tiny_ld(B, bd, k * sbk + n * sbn);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < K) acc += a[e] * b[e];
      }
      tiny_st(C, cd, (int64_t)m * ldc + n, acc, accumulate);
    }
  }
}
}  // namespace

extern "C" int ib_tiny_matmul(const void* A, int a_dtype, int64_t sam, int64_t sak, const void* B, int b_dtype, int64_t sbk,
                              int64_t sbn, void* C, int c_dtype, int64_t ldc, int accumulate, int64_t M, int64_t N, int64_t K,
                              ib_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return IB_E_ARG;
  for (int d : {a_dtype, b_dtype, c_dtype})
    if (d != IB_F32 && d != IB_BF16) return IB_E_DTYPE;
  if (M * N > (1 << 22) || K > (1 << 16) || M * N * K > ((int64_t)1 << 28)) return IB_E_UNSUPPORTED;   // "tiny" only
  hipStream_t s = ib_s(stream);
  if (K >= 64)
    hipLaunchKernelGGL((tiny_matmul_kernel<true>), dim3(ib_grid_1d(M * N, 4, 256 * 16)), dim3(256), 0, s, A, a_dtype, sam, sak, B,
                       b_dtype, sbk, sbn, C, c_dtype, ldc, accumulate, (int)M, (int)N, (int)K);
  else
    hipLaunchKernelGGL((tiny_matmul_kernel<false>), dim3(ib_grid_1d(M * N, 256, 256 * 16)), dim3(256), 0, s, A, a_dtype, sam, sak,
                       B, b_dtype, sbk, sbn, C, c_dtype, ldc, accumulate, (int)M, (int)N, (int)K);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
