// bf16 MFMA temporal self-attention for gfx950, head dim 64 (TransformerBaseline.py:12-13,29 core).
//
// One 256-thread workgroup per (window, head).  The window is short (T <= 256 frames), so the head's
// Q, K, V (and dO in the backward) are staged ONCE into LDS as [Tp][64] bf16 images (rows padded to 144 B,
// rows >= T zero-filled), and every product runs on v_mfma_f32_16x16x32_bf16:
//
//   forward, per 16-query block (one wave):
//     S^T[key][q] = K . Q^T            (A = K rows, B = Q rows; lane owns query q = lane%16, keys in registers)
//     P = softmax over keys            (in-lane over registers + xor-shuffles across the 4 lane groups)
//     O^T[d][q]  = V^T . P^T           (A = V^T by ds_read_b64_tr_b16, B = P taken STRAIGHT from the S
//                                       accumulators: the reduction index (key) is permuted identically on
//                                       both operands, so P never goes through LDS)
//   backward = two phases, all sums kept in registers, no float atomics (bitwise reproducible):
//     phase 1 per 16-query block: S^T, dP^T = V . dO^T, dS = P (dP - D), dQ^T = K^T . dS^T
//     phase 2 per 16-key block  : S = Q . K^T, dP = dO . V^T, dS, dV^T = dO^T . P, dK^T = Q^T . dS
//   (S / dP are recomputed in phase 2 instead of being exchanged: 7 instead of 5 products, but nothing T x T
//   ever leaves registers).  D[q] = sum_key P dP is summed in phase 1 from the row's P and dP registers (fp32) and
//   passed to phase 2 through LDS; the saved output O is not read.
//
// Lane maps used (same as gemm.hip, verified on hardware by tests/test_hip_kernels.py::test_tr16...):
//   mfma(A, B, C): A[row = lane%16][k = 8*(lane/16)+j], B[k = 8*(lane/16)+j][col = lane%16],
//   D[row = 4*(lane/16)+r][col = lane%16].
#include "ib_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int LDR = 72;  // bf16 elements per LDS row (64 + 8 pad): 144 B
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ bf16x8_t row_frag(const bf16_t* img, int row, int ks, int lane) {
  // 8 consecutive d (k index) of one image row: A or B operand of a product that reduces over d
  return *reinterpret_cast<const bf16x8_t*>(img + row * LDR + ks * 32 + 8 * (lane >> 4));
}
// transposed operand: A[i = d][k = image row], 4 consecutive image rows per 16-lane group
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img, int row_lo, int row_hi, int dt, int lane) {
  const int qq = (lane & 15) >> 2, pp = lane & 3, g = lane >> 4;
  const bf16_t* a0 = img + (row_lo + 4 * g + qq) * LDR + dt * 16 + 4 * pp;
  const bf16_t* a1 = img + (row_hi + 4 * g + qq) * LDR + dt * 16 + 4 * pp;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a1));
  s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
// accumulator pair -> operand of the next product: element j<4 <- tile a reg j, j>=4 <- tile b reg j-4
__device__ __forceinline__ bf16x8_t acc_frag(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t f;
  f[0] = (bf16_t)a[0]; f[1] = (bf16_t)a[1]; f[2] = (bf16_t)a[2]; f[3] = (bf16_t)a[3];
  f[4] = (bf16_t)b[0]; f[5] = (bf16_t)b[1]; f[6] = (bf16_t)b[2]; f[7] = (bf16_t)b[3];
  return f;
}
__device__ __forceinline__ float group4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// stage one [T][64] slice (row stride `ld` elements in global memory) into a [Tp][LDR] LDS image
// Eight pieces per thread are requested before the first is stored: as a rolled loop (one load, one LDS store per trip) the
// staging of a 200-frame head was 14 dependent memory round trips per image -- most of the kernel's time at T = 200.
template <int NTHR = 256>
__device__ __forceinline__ void stage_image(bf16_t* img, const bf16_t* __restrict__ src, int64_t ld, int T, int Tp) {
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < Tp * 8; i0 += NTHR * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + NTHR * u, r = i >> 3, c = i & 7;
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      if (r < T) v[u] = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + c * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + NTHR * u, r = i >> 3, c = i & 7;
      if (i < Tp * 8) *reinterpret_cast<uint4*>(img + r * LDR + c * 8) = v[u];
    }
  }
}

// DROP: nn.MultiheadAttention's dropout on the probabilities -- the multiplier (0 or 1 / (1 - p)) of (query, key) is a hash
// (ib_common.h) applied to the P registers after the row sum; the backward regenerates it
template <int NT, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_mfma(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                     float* __restrict__ lse, int T, int H, float scale, IbAttnDrop drop,
                                                     int qsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int Tp = NT * 16;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);           // only K and V are staged: a query row is read once, by the
  bf16_t* Vs = Ks + Tp * LDR;                                  // one lane group that owns it, straight into its fragments
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  // qsplit > 1 (few windows: the sampler at B = 1 / 16 is 8 / 128 (window, head) pairs on 256 CUs): `qsplit` workgroups
  // share one (window, head), each stages K and V and takes every qsplit-th group of four query blocks
  const int bh = (int)blockIdx.x / qsplit, part = (int)blockIdx.x % qsplit;
  const int b = bh / H, h = bh % H;
  const int d = H * 64;
  const bf16_t* base = qkv + (int64_t)b * T * 3 * d + h * 64;
  stage_image(Ks, base + d, 3 * d, T, Tp);
  stage_image(Vs, base + 2 * d, 3 * d, T, Tp);
  uint32_t dkey = 0;
  if constexpr (DROP) dkey = ib_attn_drop_key(drop, bh);
  __syncthreads();
  const int nqb = (T + 15) >> 4;
  for (int qb = part * 4 + wave; qb < nqb; qb += 4 * qsplit) {
    const int q = qb * 16 + (lane & 15);
    bf16x8_t qf[2];
    {
      const bf16_t* qrow = base + (int64_t)min(q, T - 1) * 3 * d + 8 * g;     // rows beyond T: finite duplicates, never stored
      __builtin_memcpy(&qf[0], __builtin_assume_aligned(qrow, 16), 16);
      __builtin_memcpy(&qf[1], __builtin_assume_aligned(qrow + 32, 16), 16);
    }
    f32x4_t s[NT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 0, lane), qf[0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 1, lane), qf[1], a, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        a[r] = key < T ? a[r] * scale : -INFINITY;
        m = fmaxf(m, a[r]);
      }
      s[kt] = a;
    }
    m = group4_max(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(s[kt][r] - m);     // -inf -> 0
        l += pv;
        if constexpr (DROP) s[kt][r] = pv * ib_attn_drop_mult(drop, dkey, q, kt * 16 + 4 * g + r);
        else s[kt][r] = pv;
      }
    l = group4_sum(l);
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NT / 2; ++kp) {
      const bf16x8_t pf = acc_frag(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vs, 32 * kp, 32 * kp + 16, dt, lane), pf, o[dt], 0, 0, 0);
    }
    if (q < T) {
      const float inv = 1.f / l;
      bf16_t* orow = out + ((int64_t)b * T + q) * d + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4_t w;
        w[0] = (bf16_t)(o[dt][0] * inv); w[1] = (bf16_t)(o[dt][1] * inv);
        w[2] = (bf16_t)(o[dt][2] * inv); w[3] = (bf16_t)(o[dt][3] * inv);
        *reinterpret_cast<bf16x4_t*>(orow + dt * 16 + 4 * g) = w;
      }
      if (g == 0 && lse) lse[((int64_t)b * H + h) * T + q] = m + logf(l);
    }
  }
}

// Long windows (T > 64: the T = 200 sampler and training configs).  The kernel above keeps a whole score row in registers
// (NT = 14: 203 VGPRs -> two waves per SIMD, and with ~100 KB of K / V images per workgroup little else to overlap with:
// it ran at a third of its memory floor).  Here the row is walked TWICE -- pass A finds the row maximum, pass B recomputes
// each pair of score tiles, exponentiates and feeds P.V at once (two more QK^T MFMAs per tile: nothing next to the exp
// chain they unblock) -- so only one tile pair is live (< 128 VGPRs), and eight waves share one pair of K / V images:
// four waves per SIMD with two workgroups per CU.  Same arithmetic per element as the one-pass kernel.
template <int NT, bool DROP>
__global__ __launch_bounds__(512, 4) void attn_fwd_mfma_2p(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                        float* __restrict__ lse, int T, int H, float scale, IbAttnDrop drop,
                                                        int qsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int Tp = NT * 16;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vs = Ks + Tp * LDR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int bh = (int)blockIdx.x / qsplit, part = (int)blockIdx.x % qsplit;
  const int b = bh / H, h = bh % H;
  const int d = H * 64;
  const bf16_t* base = qkv + (int64_t)b * T * 3 * d + h * 64;
  stage_image<512>(Ks, base + d, 3 * d, T, Tp);
  stage_image<512>(Vs, base + 2 * d, 3 * d, T, Tp);
  uint32_t dkey = 0;
  if constexpr (DROP) dkey = ib_attn_drop_key(drop, bh);
  __syncthreads();
  const int nqb = (T + 15) >> 4;
  for (int qb = part * 8 + wave; qb < nqb; qb += 8 * qsplit) {
    const int q = qb * 16 + (lane & 15);
    bf16x8_t qf[2];
    {
      const bf16_t* qrow = base + (int64_t)min(q, T - 1) * 3 * d + 8 * g;
      __builtin_memcpy(&qf[0], __builtin_assume_aligned(qrow, 16), 16);
      __builtin_memcpy(&qf[1], __builtin_assume_aligned(qrow + 32, 16), 16);
    }
    // pass A: row maximum
    float m = -INFINITY;
#pragma unroll 2                       // (a fully unrolled walk lets the scheduler hoist every tile's fragment reads: spills)
    for (int kt = 0; kt < NT; ++kt) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 0, lane), qf[0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 1, lane), qf[1], a, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kt * 16 + 4 * g + r < T) m = fmaxf(m, a[r] * scale);
    }
    m = group4_max(m);
    // pass B: probabilities of a tile pair -> P.V
    float l = 0.f;
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kp = 0; kp < NT / 2; ++kp) {
      f32x4_t s2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int kt = 2 * kp + hh;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 0, lane), qf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, kt * 16 + (lane & 15), 1, lane), qf[1], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const float pv = key < T ? __expf(a[r] * scale - m) : 0.f;
          l += pv;
          if constexpr (DROP) a[r] = pv * ib_attn_drop_mult(drop, dkey, q, key);
          else a[r] = pv;
        }
        s2[hh] = a;
      }
      const bf16x8_t pf = acc_frag(s2[0], s2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vs, 32 * kp, 32 * kp + 16, dt, lane), pf, o[dt], 0, 0, 0);
    }
    l = group4_sum(l);
    if (q < T) {
      const float inv = 1.f / l;
      bf16_t* orow = out + ((int64_t)b * T + q) * d + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4_t w;
        w[0] = (bf16_t)(o[dt][0] * inv); w[1] = (bf16_t)(o[dt][1] * inv);
        w[2] = (bf16_t)(o[dt][2] * inv); w[3] = (bf16_t)(o[dt][3] * inv);
        *reinterpret_cast<bf16x4_t*>(orow + dt * 16 + 4 * g) = w;
      }
      if (g == 0 && lse) lse[((int64_t)b * H + h) * T + q] = m + logf(l);
    }
  }
}

template <int NT, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_mfma(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                     const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                     bf16_t* __restrict__ dqkv, int T, int H, float scale, IbAttnDrop drop) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int Tp = NT * 16;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Ks = Qs + Tp * LDR;
  bf16_t* Vs = Ks + Tp * LDR;
  bf16_t* Gs = Vs + Tp * LDR;                                  // dO
  float* Dl = reinterpret_cast<float*>(Gs + Tp * LDR);         // [Tp] rowsum(dO * O)
  float* Ll = Dl + Tp;                                         // [Tp] lse
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * 64;
  const bf16_t* base = qkv + (int64_t)b * T * 3 * d + h * 64;
  bf16_t* dbase = dqkv + (int64_t)b * T * 3 * d + h * 64;
  (void)out;      // D = rowsum(dP x P) is summed from the recomputed probabilities (phase 1), not from the rounded output
  const bf16_t* gbase = dout + (int64_t)b * T * d + h * 64;
  stage_image(Qs, base, 3 * d, T, Tp);
  stage_image(Ks, base + d, 3 * d, T, Tp);
  stage_image(Vs, base + 2 * d, 3 * d, T, Tp);
  stage_image(Gs, gbase, d, T, Tp);
  for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
    Ll[t] = t < T ? lse[((int64_t)b * H + h) * T + t] : 0.f;
    Dl[t] = 0.f;                                               // rows beyond the last query block stay 0
  }
  uint32_t dkey = 0;
  if constexpr (DROP) dkey = ib_attn_drop_key(drop, blockIdx.x);
  __syncthreads();

  // ---------------- phase 1: dQ, one 16-query block per wave iteration (lane owns query q)
  const int nb = (T + 15) >> 4;
  for (int qb = wave; qb < nb; qb += 4) {
    const int q = qb * 16 + (lane & 15);
    const bf16x8_t qf[2] = {row_frag(Qs, q, 0, lane), row_frag(Qs, q, 1, lane)};
    const bf16x8_t gf[2] = {row_frag(Gs, q, 0, lane), row_frag(Gs, q, 1, lane)};
    const float Lq = Ll[q];
    // pass 1 over the key tiles: P and (masked) dP of the whole row stay in registers, and D[q] = sum_key P dP is summed
    // from them in fp32 -- exact for the values the products see.  (rowsum(dO x O) of the SAVED output is the same number
    // only up to O's bf16 rounding, and dS = P (dP - D) cancels: with a peaked softmax that rounding was the whole dQ / dK
    // error of the bf16 path.)
    f32x4_t pr[NT], dpr[NT];
    float Dq = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const int krow = kt * 16 + (lane & 15);
      f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, krow, 0, lane), qf[0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, krow, 1, lane), qf[1], a, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, krow, 0, lane), gf[0], dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, krow, 1, lane), gf[1], dp, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const float pv = key < T ? __expf(a[r] * scale - Lq) : 0.f;
        if constexpr (DROP) dp[r] *= ib_attn_drop_mult(drop, dkey, q, key);
        Dq += pv * dp[r];
        a[r] = pv;
      }
      pr[kt] = a;
      dpr[kt] = dp;
    }
    Dq = group4_sum(Dq);
    if (g == 0) Dl[q] = Dq;                                    // phase 2 reads it (after the barrier below)
    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NT / 2; ++kp) {
      f32x4_t ds2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds2[hh][r] = pr[2 * kp + hh][r] * (dpr[2 * kp + hh][r] - Dq);
      const bf16x8_t sf = acc_frag(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ks, 32 * kp, 32 * kp + 16, dt, lane), sf, dq[dt], 0, 0, 0);
    }
    if (q < T) {
      bf16_t* row = dbase + (int64_t)q * 3 * d;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4_t w;
        w[0] = (bf16_t)(dq[dt][0] * scale); w[1] = (bf16_t)(dq[dt][1] * scale);
        w[2] = (bf16_t)(dq[dt][2] * scale); w[3] = (bf16_t)(dq[dt][3] * scale);
        *reinterpret_cast<bf16x4_t*>(row + dt * 16 + 4 * g) = w;
      }
    }
  }

  __syncthreads();                                             // D of every query is in LDS
  // ---------------- phase 2: dK, dV, one 16-key block per wave iteration (lane owns key)
  for (int kb = wave; kb < nb; kb += 4) {
    const int key = kb * 16 + (lane & 15);
    const bf16x8_t kf[2] = {row_frag(Ks, key, 0, lane), row_frag(Ks, key, 1, lane)};
    const bf16x8_t vf[2] = {row_frag(Vs, key, 0, lane), row_frag(Vs, key, 1, lane)};
    f32x4_t dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qp = 0; qp < NT / 2; ++qp) {
      f32x4_t pp2[2], ds2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qt = 2 * qp + hh;
        const int qrow = qt * 16 + (lane & 15);
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qs, qrow, 0, lane), kf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qs, qrow, 1, lane), kf[1], a, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Gs, qrow, 0, lane), vf[0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Gs, qrow, 1, lane), vf[1], dp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = qt * 16 + 4 * g + r;                  // this register's query
          const float pv = qq < T ? __expf(a[r] * scale - Ll[qq]) : 0.f;
          float mk = 1.f;
          if constexpr (DROP) mk = ib_attn_drop_mult(drop, dkey, qq, key);
          a[r] = pv * mk;                                      // dV sums the dropped probabilities
          dp[r] = pv * (mk * dp[r] - Dl[qq]);
        }
        pp2[hh] = a;
        ds2[hh] = dp;
      }
      const bf16x8_t pf = acc_frag(pp2[0], pp2[1]);
      const bf16x8_t sf = acc_frag(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Gs, 32 * qp, 32 * qp + 16, dt, lane), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qs, 32 * qp, 32 * qp + 16, dt, lane), sf, dk[dt], 0, 0, 0);
      }
    }
    if (key < T) {
      bf16_t* row = dbase + (int64_t)key * 3 * d;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4_t wk, wv;
        wk[0] = (bf16_t)(dk[dt][0] * scale); wk[1] = (bf16_t)(dk[dt][1] * scale);
        wk[2] = (bf16_t)(dk[dt][2] * scale); wk[3] = (bf16_t)(dk[dt][3] * scale);
        wv[0] = (bf16_t)dv[dt][0]; wv[1] = (bf16_t)dv[dt][1]; wv[2] = (bf16_t)dv[dt][2]; wv[3] = (bf16_t)dv[dt][3];
        *reinterpret_cast<bf16x4_t*>(row + d + dt * 16 + 4 * g) = wk;
        *reinterpret_cast<bf16x4_t*>(row + 2 * d + dt * 16 + 4 * g) = wv;
      }
    }
  }
}

template <typename K> int ensure_lds(K k, size_t need, int& cur) {
  if ((int)need <= cur) return IB_OK;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return IB_E_LAUNCH;
  cur = 160 * 1024;
  return IB_OK;
}
int g_lds_f[2][4] = {{48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}, {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}};
int g_lds_f2[2][4] = {{48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}, {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}};
int g_lds_b[2][4] = {{48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}, {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024}};

template <int NT, bool DROP>
int launch_fwd2(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, float scale, int slot,
                const IbAttnDrop& a, hipStream_t s) {
  const size_t lds = (size_t)2 * NT * 16 * LDR * 2;
  static const bool one_pass = ib_ab_set("IB_ATTN_ONE_PASS");
  // long windows: two-pass kernel, eight waves per workgroup -- when that still fills the chip (a single window is better
  // off with four query-block groups per head on the one-pass kernel: 5.5 against 7.9 us at B = 1, T = 200)
  if (NT > 4 && !one_pass && B * H * ((T + 127) / 128) >= 256) {
    auto k2 = attn_fwd_mfma_2p<NT, DROP>;
    if (ensure_lds(k2, lds, g_lds_f2[DROP][slot]) != IB_OK) return IB_E_LAUNCH;
    const int groups8 = (int)((T + 127) / 128);                // groups of eight query blocks (one per wave)
    int qs = 1;
    if (B * H < 256) qs = (int)std::min<int64_t>(groups8, std::max<int64_t>(1, 256 / (B * H)));
    hipLaunchKernelGGL(k2, dim3((unsigned)(B * H * qs)), dim3(512), lds, s, (const bf16_t*)qkv, (bf16_t*)out, lse, (int)T,
                       (int)H, scale, a, qs);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  auto k = attn_fwd_mfma<NT, DROP>;
  if (ensure_lds(k, lds, g_lds_f[DROP][slot]) != IB_OK) return IB_E_LAUNCH;
  // fewer (window, head) pairs than CUs: split the query blocks of a pair over several workgroups
  const int groups = (int)((T + 63) / 64);                     // groups of four query blocks (one per wave)
  int qsplit = 1;
  if (B * H < 256) qsplit = (int)std::min<int64_t>(groups, std::max<int64_t>(1, 256 / (B * H)));
  hipLaunchKernelGGL(k, dim3((unsigned)(B * H * qsplit)), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)out, lse, (int)T,
                     (int)H, scale, a, qsplit);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
template <int NT>
int launch_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, float scale, int slot,
               const IbAttnDrop* drop, hipStream_t s) {
  return drop ? launch_fwd2<NT, true>(qkv, out, lse, B, T, H, scale, slot, *drop, s)
              : launch_fwd2<NT, false>(qkv, out, lse, B, T, H, scale, slot, IbAttnDrop{}, s);
}
template <int NT, bool DROP>
int launch_bwd2(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T,
                int64_t H, float scale, int slot, const IbAttnDrop& a, hipStream_t s) {
  const size_t lds = (size_t)4 * NT * 16 * LDR * 2 + (size_t)2 * NT * 16 * 4;
  auto k = attn_bwd_mfma<NT, DROP>;
  if (ensure_lds(k, lds, g_lds_b[DROP][slot]) != IB_OK) return IB_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3((unsigned)(B * H)), dim3(256), lds, s, (const bf16_t*)qkv, (const bf16_t*)out,
                     (const bf16_t*)dout, lse, (bf16_t*)dqkv, (int)T, (int)H, scale, a);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
template <int NT>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T,
               int64_t H, float scale, int slot, const IbAttnDrop* drop, hipStream_t s) {
  return drop ? launch_bwd2<NT, true>(qkv, out, dout, lse, dqkv, B, T, H, scale, slot, *drop, s)
              : launch_bwd2<NT, false>(qkv, out, dout, lse, dqkv, B, T, H, scale, slot, IbAttnDrop{}, s);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

// Internal entry points used by attention.hip's dispatcher (not part of the public C-ABI).
// Return IB_E_UNSUPPORTED when the shape / alignment is outside this kernel's domain.
int ib_attention_fwd_mfma_bf16(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, int64_t dh,
                               const IbAttnDrop* drop, hipStream_t s) {
  if (dh != 64 || T > 256 || !al16(qkv) || !al16(out)) return IB_E_UNSUPPORTED;
  const float scale = 0.125f;
  if (T <= 64) return launch_fwd<4>(qkv, out, lse, B, T, H, scale, 0, drop, s);
  if (T <= 128) return launch_fwd<8>(qkv, out, lse, B, T, H, scale, 1, drop, s);
  if (T <= 224) return launch_fwd<14>(qkv, out, lse, B, T, H, scale, 2, drop, s);
  return launch_fwd<16>(qkv, out, lse, B, T, H, scale, 3, drop, s);
}

int ib_attention_bwd_mfma_bf16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                               int64_t B, int64_t T, int64_t H, int64_t dh, const IbAttnDrop* drop, hipStream_t s) {
  if (dh != 64 || T > 256 || !al16(qkv) || !al16(out) || !al16(dout) || !al16(dqkv)) return IB_E_UNSUPPORTED;
  const float scale = 0.125f;
  if (T <= 64) return launch_bwd<4>(qkv, out, dout, lse, dqkv, B, T, H, scale, 0, drop, s);
  if (T <= 128) return launch_bwd<8>(qkv, out, dout, lse, dqkv, B, T, H, scale, 1, drop, s);
  if (T <= 224) return launch_bwd<14>(qkv, out, dout, lse, dqkv, B, T, H, scale, 2, drop, s);
  return launch_bwd<16>(qkv, out, dout, lse, dqkv, B, T, H, scale, 3, drop, s);
}
