// Temporal self-attention core for gfx950 (nn.MultiheadAttention between in-proj and out-proj,
// src/models/TransformerBaseline.py:12-13,29): per (window, head) softmax(q k^T / sqrt(dh)) v, no mask.
// A window is short (T = 50 / 200 frames), so the whole K and V of one (window, head) are staged in
// LDS once (T x dh fp32, padded rows: conflict-free) and every query row is finished in one pass --
// no flash-style streaming.  One 256-thread workgroup per (window, head); a wave owns a query row:
// scores with keys on the lanes, softmax by wave shuffles, P.V with the head dimension on the lanes.
// Backward recomputes P from q, k and the saved log-sum-exp in two phases (dQ with K,V resident; then
// dK,dV with Q,dO resident), so no T x T tensor ever reaches HBM and no float atomics are used
// (bitwise reproducible).  This is the exact-fp32 VALU formulation used by both dtypes; storage may be bf16.
#include "ib_common.h"

namespace {

constexpr int MAX_T = 256;   // keys per lane <= 4
constexpr int MAX_DH = 128;

template <typename T, bool DROP>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                            float* __restrict__ lse, int Tn, int H, int dh, float scale,
                                                            IbAttnDrop drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldk = dh + 1;
  float* Ks = sm;                       // [Tn][ldk]
  float* Vs = Ks + Tn * ldk;            // [Tn][ldk]
  float* Ps = Vs + Tn * ldk;            // [4][MAX_T]
  float* Qs = Ps + 4 * MAX_T;           // [4][MAX_DH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * dh;
  const T* base = qkv + (int64_t)b * Tn * 3 * d;
  uint32_t dkey = 0;
  if constexpr (DROP) dkey = ib_attn_drop_key(drop, blockIdx.x);
  for (int i = threadIdx.x; i < Tn * dh; i += blockDim.x) {
    const int t = i / dh, c = i % dh;
    Ks[t * ldk + c] = ib_to_f32(base[(int64_t)t * 3 * d + d + h * dh + c]);
    Vs[t * ldk + c] = ib_to_f32(base[(int64_t)t * 3 * d + 2 * d + h * dh + c]);
  }
  __syncthreads();
  const int iters = (Tn + 3) / 4;
  for (int it = 0; it < iters; ++it) {
    const int t = it * 4 + wave;
    const bool on = t < Tn;
    if (on)
      for (int c = lane; c < dh; c += 64) Qs[wave * MAX_DH + c] = ib_to_f32(base[(int64_t)t * 3 * d + h * dh + c]) * scale;
    __syncthreads();
    float s[4];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = jj * 64 + lane;
      s[jj] = -INFINITY;
      if (on && j < Tn) {
        float a = 0.f;
        for (int c = 0; c < dh; ++c) a += Qs[wave * MAX_DH + c] * Ks[j * ldk + c];
        s[jj] = a;
        m = fmaxf(m, a);
      }
    }
    m = ib_wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = jj * 64 + lane;
      if (on && j < Tn) {
        const float p = expf(s[jj] - m);
        l += p;                                                     // the softmax is normalised BEFORE its dropout
        if constexpr (DROP) Ps[wave * MAX_T + j] = p * ib_attn_drop_mult(drop, dkey, t, j);
        else Ps[wave * MAX_T + j] = p;
      }
    }
    l = ib_wave_sum(l);
    __syncthreads();
    if (on) {
      const float inv = 1.f / l;
      for (int c = lane; c < dh; c += 64) {
        float o = 0.f;
        for (int j = 0; j < Tn; ++j) o += Ps[wave * MAX_T + j] * Vs[j * ldk + c];
        out[((int64_t)b * Tn + t) * d + h * dh + c] = ib_from_f32<T>(o * inv);
      }
      if (lane == 0 && lse) lse[((int64_t)b * H + h) * Tn + t] = m + logf(l);
    }
    __syncthreads();
  }
}

// with dropout on the probabilities (O = (P x mask / (1 - p)) V): dV sums the DROPPED probabilities, dP picks up the same
// multiplier, and D = rowsum(dP x P) is still rowsum(dO x O) of the saved (dropped) output
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                            const T* __restrict__ dout, const float* __restrict__ lse,
                                                            T* __restrict__ dqkv, int Tn, int H, int dh, float scale,
                                                            IbAttnDrop drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldk = dh + 1;
  float* A1 = sm;                       // phase 1: K   ; phase 2: Q
  float* A2 = A1 + Tn * ldk;            // phase 1: V   ; phase 2: dO
  float* R1 = A2 + Tn * ldk;            // [4][MAX_T]  dS row
  float* R2 = R1 + 4 * MAX_T;           // [4][MAX_T]  P row
  float* X1 = R2 + 4 * MAX_T;           // [4][MAX_DH]
  float* X2 = X1 + 4 * MAX_DH;          // [4][MAX_DH]
  float* Dl = X2 + 4 * MAX_DH;          // [MAX_T]  rowsum(dO * O)
  float* Ll = Dl + MAX_T;               // [MAX_T]  lse
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int d = H * dh;
  const T* base = qkv + (int64_t)b * Tn * 3 * d;
  T* dbase = dqkv + (int64_t)b * Tn * 3 * d;
  (void)out;
  const T* dobase = dout + (int64_t)b * Tn * d + h * dh;
  uint32_t dkey = 0;
  if constexpr (DROP) dkey = ib_attn_drop_key(drop, blockIdx.x);
  for (int i = threadIdx.x; i < Tn * dh; i += blockDim.x) {
    const int t = i / dh, c = i % dh;
    A1[t * ldk + c] = ib_to_f32(base[(int64_t)t * 3 * d + d + h * dh + c]);
    A2[t * ldk + c] = ib_to_f32(base[(int64_t)t * 3 * d + 2 * d + h * dh + c]);
  }
  for (int t = threadIdx.x; t < Tn; t += blockDim.x) Ll[t] = lse[((int64_t)b * H + h) * Tn + t];
  __syncthreads();
  const int iters = (Tn + 3) / 4;
  // ---- phase 1: dQ (K, V resident)
  for (int it = 0; it < iters; ++it) {
    const int t = it * 4 + wave;
    const bool on = t < Tn;
    if (on) {
      for (int c = lane; c < dh; c += 64) {
        X1[wave * MAX_DH + c] = ib_to_f32(base[(int64_t)t * 3 * d + h * dh + c]);
        X2[wave * MAX_DH + c] = ib_to_f32(dobase[(int64_t)t * d + c]);
      }
    }
    __syncthreads();
    // D[t] = sum_j P dP from the row's own probabilities (the saved output is not read: in bf16 storage its rounding does not
    // cancel in dS = P (dP - D)); a wave owns the row, so R1 / R2 are wave-private between the two passes
    float dsum = 0.f;
    if (on) {
      const float lt = Ll[t];
      for (int j = lane; j < Tn; j += 64) {
        float a = 0.f, dp = 0.f;
        for (int c = 0; c < dh; ++c) {
          a += X1[wave * MAX_DH + c] * A1[j * ldk + c];
          dp += X2[wave * MAX_DH + c] * A2[j * ldk + c];
        }
        const float p = expf(a * scale - lt);
        if constexpr (DROP) dp *= ib_attn_drop_mult(drop, dkey, t, j);
        R1[wave * MAX_T + j] = dp;
        R2[wave * MAX_T + j] = p;
        dsum += p * dp;
      }
    }
    dsum = ib_wave_sum(dsum);
    if (on) {
      if (lane == 0) Dl[t] = dsum;
      for (int j = lane; j < Tn; j += 64) R1[wave * MAX_T + j] = R2[wave * MAX_T + j] * (R1[wave * MAX_T + j] - dsum);
    }
    __syncthreads();
    if (on) {
      for (int c = lane; c < dh; c += 64) {
        float g = 0.f;
        for (int j = 0; j < Tn; ++j) g += R1[wave * MAX_T + j] * A1[j * ldk + c];
        dbase[(int64_t)t * 3 * d + h * dh + c] = ib_from_f32<T>(g * scale);
      }
    }
    __syncthreads();
  }
  // ---- phase 2: dK, dV (Q, dO resident)
  for (int i = threadIdx.x; i < Tn * dh; i += blockDim.x) {
    const int t = i / dh, c = i % dh;
    A1[t * ldk + c] = ib_to_f32(base[(int64_t)t * 3 * d + h * dh + c]);
    A2[t * ldk + c] = ib_to_f32(dobase[(int64_t)t * d + c]);
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int j = it * 4 + wave;
    const bool on = j < Tn;
    if (on) {
      for (int c = lane; c < dh; c += 64) {
        X1[wave * MAX_DH + c] = ib_to_f32(base[(int64_t)j * 3 * d + d + h * dh + c]);
        X2[wave * MAX_DH + c] = ib_to_f32(base[(int64_t)j * 3 * d + 2 * d + h * dh + c]);
      }
    }
    __syncthreads();
    if (on) {
      for (int t = lane; t < Tn; t += 64) {
        float a = 0.f, dp = 0.f;
        for (int c = 0; c < dh; ++c) {
          a += A1[t * ldk + c] * X1[wave * MAX_DH + c];
          dp += A2[t * ldk + c] * X2[wave * MAX_DH + c];
        }
        const float p = expf(a * scale - Ll[t]);
        float mk = 1.f;
        if constexpr (DROP) mk = ib_attn_drop_mult(drop, dkey, t, j);
        R1[wave * MAX_T + t] = p * (mk * dp - Dl[t]);
        R2[wave * MAX_T + t] = p * mk;
      }
    }
    __syncthreads();
    if (on) {
      for (int c = lane; c < dh; c += 64) {
        float gk = 0.f, gv = 0.f;
        for (int t = 0; t < Tn; ++t) {
          gk += R1[wave * MAX_T + t] * A1[t * ldk + c];
          gv += R2[wave * MAX_T + t] * A2[t * ldk + c];
        }
        dbase[(int64_t)j * 3 * d + d + h * dh + c] = ib_from_f32<T>(gk * scale);
        dbase[(int64_t)j * 3 * d + 2 * d + h * dh + c] = ib_from_f32<T>(gv);
      }
    }
    __syncthreads();
  }
}

// the multipliers (0 or 1 / (1 - p)) one (seed, step) draws, written out [B, H, T, T]: test / debugging aid (a float64
// restatement of a train-mode layer needs the masks the kernels used)
__global__ __launch_bounds__(256) void attention_drop_mask_kernel(float* __restrict__ mask, int Tn, IbAttnDrop drop) {
  const uint32_t dkey = ib_attn_drop_key(drop, blockIdx.x);
  for (int i = threadIdx.x; i < Tn * Tn; i += blockDim.x)
    mask[(int64_t)blockIdx.x * Tn * Tn + i] = ib_attn_drop_mult(drop, dkey, i / Tn, i % Tn);
}

// raise the dynamic-LDS limit of a kernel once (and again only if a later call needs more); not a
// stream operation, so it is kept out of the steady-state launch path (and of any graph capture)
template <typename K> int ensure_lds(K k, size_t need, int& cur) {
  if ((int)need <= cur) return IB_OK;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return IB_E_LAUNCH;
  cur = 160 * 1024;
  return IB_OK;
}
int g_lds_fwd[2][2] = {{48 * 1024, 48 * 1024}, {48 * 1024, 48 * 1024}};       // [dtype][DROP]
int g_lds_bwd[2][2] = {{48 * 1024, 48 * 1024}, {48 * 1024, 48 * 1024}};

// p = 0 -> the plain kernels; otherwise the arguments of the mask hash.  IB_E_ARG for p outside [0, 1)
int drop_args(float p, uint32_t seed, int32_t step, const int32_t* step_dev, IbAttnDrop& a) {
  if (!(p >= 0.f) || p >= 1.f) return IB_E_ARG;
  a.thr = (uint32_t)((double)p * 4294967296.0);
  a.keep = 1.f / (1.f - p);
  a.seed = seed; a.step = step; a.step_dev = step_dev;
  return IB_OK;
}

template <typename T, bool DROP>
int launch_fwd_valu(const void* qkv, void* out, float* lse, int64_t B, int64_t T_, int64_t H, int64_t dh, size_t lds,
                    float scale, const IbAttnDrop& a, int& cur, hipStream_t s) {
  auto k = attention_fwd_kernel<T, DROP>;
  if (ensure_lds(k, lds, cur) != IB_OK) return IB_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3((unsigned)(B * H)), dim3(256), lds, s, (const T*)qkv, (T*)out, lse, (int)T_, (int)H, (int)dh,
                     scale, a);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
template <typename T, bool DROP>
int launch_bwd_valu(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T_,
                    int64_t H, int64_t dh, size_t lds, float scale, const IbAttnDrop& a, int& cur, hipStream_t s) {
  auto k = attention_bwd_kernel<T, DROP>;
  if (ensure_lds(k, lds, cur) != IB_OK) return IB_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3((unsigned)(B * H)), dim3(256), lds, s, (const T*)qkv, (const T*)out, (const T*)dout, lse,
                     (T*)dqkv, (int)T_, (int)H, (int)dh, scale, a);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

size_t fwd_lds(int T, int dh) { return ((size_t)2 * T * (dh + 1) + 4 * MAX_T + 4 * MAX_DH) * sizeof(float); }
size_t bwd_lds(int T, int dh) {
  return ((size_t)2 * T * (dh + 1) + 8 * MAX_T + 8 * MAX_DH + 2 * MAX_T) * sizeof(float);
}

}  // namespace

// bf16 MFMA kernels (attention_mfma.hip); IB_E_UNSUPPORTED -> use the generic fp32-VALU kernels below
int ib_attention_fwd_mfma_bf16(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, int64_t dh,
                               const IbAttnDrop* drop, hipStream_t s);
int ib_attention_bwd_mfma_bf16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                               int64_t B, int64_t T, int64_t H, int64_t dh, const IbAttnDrop* drop, hipStream_t s);

extern "C" int ib_attention_fwd_drop(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, int64_t dh,
                                     float p, uint32_t seed, int32_t step, const int32_t* step_dev, int dtype,
                                     ib_stream_t stream) {
  if (!qkv || !out || B <= 0 || T <= 0 || H <= 0 || dh <= 0) return IB_E_ARG;
  if (T > MAX_T || dh > MAX_DH) return IB_E_UNSUPPORTED;
  IbAttnDrop a{};
  if (drop_args(p, seed, step, step_dev, a) != IB_OK) return IB_E_ARG;
  const bool dr = p > 0.f;
  if (dtype == IB_BF16 && lse) {
    const int rc = ib_attention_fwd_mfma_bf16(qkv, out, lse, B, T, H, dh, dr ? &a : nullptr, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) return rc;
  }
  const size_t lds = fwd_lds((int)T, (int)dh);
  if (lds > 160 * 1024) return IB_E_UNSUPPORTED;
  const float scale = 1.f / sqrtf((float)dh);
  hipStream_t s = ib_s(stream);
  if (dtype == IB_F32)
    return dr ? launch_fwd_valu<float, true>(qkv, out, lse, B, T, H, dh, lds, scale, a, g_lds_fwd[0][1], s)
              : launch_fwd_valu<float, false>(qkv, out, lse, B, T, H, dh, lds, scale, a, g_lds_fwd[0][0], s);
  if (dtype == IB_BF16)
    return dr ? launch_fwd_valu<bf16_t, true>(qkv, out, lse, B, T, H, dh, lds, scale, a, g_lds_fwd[1][1], s)
              : launch_fwd_valu<bf16_t, false>(qkv, out, lse, B, T, H, dh, lds, scale, a, g_lds_fwd[1][0], s);
  return IB_E_DTYPE;
}

extern "C" int ib_attention_bwd_drop(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                     int64_t B, int64_t T, int64_t H, int64_t dh, float p, uint32_t seed, int32_t step,
                                     const int32_t* step_dev, int dtype, ib_stream_t stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0 || dh <= 0) return IB_E_ARG;
  if (T > MAX_T || dh > MAX_DH) return IB_E_UNSUPPORTED;
  IbAttnDrop a{};
  if (drop_args(p, seed, step, step_dev, a) != IB_OK) return IB_E_ARG;
  const bool dr = p > 0.f;
  if (dtype == IB_BF16) {
    const int rc = ib_attention_bwd_mfma_bf16(qkv, out, dout, lse, dqkv, B, T, H, dh, dr ? &a : nullptr, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) return rc;
  }
  const size_t lds = bwd_lds((int)T, (int)dh);
  if (lds > 160 * 1024) return IB_E_UNSUPPORTED;
  const float scale = 1.f / sqrtf((float)dh);
  hipStream_t s = ib_s(stream);
  if (dtype == IB_F32)
    return dr ? launch_bwd_valu<float, true>(qkv, out, dout, lse, dqkv, B, T, H, dh, lds, scale, a, g_lds_bwd[0][1], s)
              : launch_bwd_valu<float, false>(qkv, out, dout, lse, dqkv, B, T, H, dh, lds, scale, a, g_lds_bwd[0][0], s);
  if (dtype == IB_BF16)
    return dr ? launch_bwd_valu<bf16_t, true>(qkv, out, dout, lse, dqkv, B, T, H, dh, lds, scale, a, g_lds_bwd[1][1], s)
              : launch_bwd_valu<bf16_t, false>(qkv, out, dout, lse, dqkv, B, T, H, dh, lds, scale, a, g_lds_bwd[1][0], s);
  return IB_E_DTYPE;
}

extern "C" int ib_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, int64_t dh,
                                int dtype, ib_stream_t stream) {
  return ib_attention_fwd_drop(qkv, out, lse, B, T, H, dh, 0.f, 0u, 0, nullptr, dtype, stream);
}

extern "C" int ib_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                int64_t B, int64_t T, int64_t H, int64_t dh, int dtype, ib_stream_t stream) {
  return ib_attention_bwd_drop(qkv, out, dout, lse, dqkv, B, T, H, dh, 0.f, 0u, 0, nullptr, dtype, stream);
}

extern "C" int ib_attention_drop_mask(float* mask, int64_t B, int64_t T, int64_t H, float p, uint32_t seed, int32_t step,
                                      const int32_t* step_dev, ib_stream_t stream) {
  if (!mask || B <= 0 || T <= 0 || H <= 0) return IB_E_ARG;
  if (T > MAX_T) return IB_E_UNSUPPORTED;
  IbAttnDrop a{};
  if (drop_args(p, seed, step, step_dev, a) != IB_OK) return IB_E_ARG;
  hipLaunchKernelGGL(attention_drop_mask_kernel, dim3((unsigned)(B * H)), dim3(256), 0, ib_s(stream), mask, (int)T, a);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
