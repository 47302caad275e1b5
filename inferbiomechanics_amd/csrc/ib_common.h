// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels. wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ib_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define IB_WAVE 64

#define IB_CHECK_LAUNCH()                                    \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return IB_E_LAUNCH;               \
  } while (0)

// dispatch-debug word (util.hip; ib_debug_last_path()): which kernel family the calling THREAD's last entry point took.
// thread_local: the library keeps no mutable state that two host threads share (re-entrant C-ABI).
extern thread_local int g_ib_last_path;
#define IB_PATH(code) (g_ib_last_path = (code))

// A/B switches and in-kernel profiling hooks exist only in MEASUREMENT builds (-DIB_AB: lib/ab/libib_hip_ab.so, loaded by
// tools/ and two tests through IB_HIP_LIB).  The shipping library reads no environment variable: every switch is its
// default, a compile-time constant.
#ifdef IB_AB
#include <stdlib.h>
static inline int ib_ab_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool ib_ab_set(const char* name) { return getenv(name) != nullptr; }
#define IB_AB_PROF(ptr) (ptr)
#else
static inline int ib_ab_int(const char*, int dflt) { return dflt; }
static inline bool ib_ab_set(const char*) { return false; }
#define IB_AB_PROF(ptr) (nullptr)
#endif

static inline hipStream_t ib_s(ib_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float ib_to_f32(float v) { return v; }
__device__ __forceinline__ float ib_to_f32(bf16_t v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T ib_from_f32(float v);
template <> __device__ __forceinline__ float ib_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t ib_from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }

// activation forward; accurate libm forms (parity mode must hold <= 1e-3 rel vs PyTorch CPU fp32)
__device__ __forceinline__ float ib_act_fwd(int act, float v) {
  switch (act) {
    case IB_ACT_RELU: return v > 0.f ? v : 0.f;
    case IB_ACT_TANH: return tanhf(v);
    case IB_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case IB_ACT_SILU: return v / (1.f + expf(-v));
    case IB_ACT_ELU: return v > 0.f ? v : expf(v) - 1.f;
    default: return v;
  }
}
// derivative factor given aux (= layer OUTPUT for relu/tanh/sigmoid/elu, PRE-activation for silu)
__device__ __forceinline__ float ib_act_bwd(int act, float aux) {
  switch (act) {
    case IB_ACT_RELU: return aux > 0.f ? 1.f : 0.f;
    case IB_ACT_TANH: return 1.f - aux * aux;
    case IB_ACT_SIGMOID: return aux * (1.f - aux);
    case IB_ACT_ELU: return aux > 0.f ? 1.f : aux + 1.f;     // aux = output y = exp(x) - 1 for x <= 0
    case IB_ACT_SILU: {
      float s = 1.f / (1.f + expf(-aux));
      return s * (1.f + aux * (1.f - s));
    }
    default: return 1.f;
  }
}

// 64-lane butterfly reductions (wave shuffles; no LDS)
__device__ __forceinline__ float ib_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ib_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// lowbias32 (Wellons): a full-avalanche 32-bit mixer -- the counter-based dropout masks are hashes of (seed, step, element)
__device__ __forceinline__ uint32_t ib_mix32(uint32_t v) {
  v ^= v >> 16; v *= 0x21f0aaadu;
  v ^= v >> 15; v *= 0x735a2d97u;
  v ^= v >> 15;
  return v;
}
// attention-probability dropout (nn.MultiheadAttention(dropout=p), TransformerBaseline.py:12-13): the mask of probability
// (window b, head h, query q, key j) is a hash of (seed, step, b*H+h, q, j), so the backward regenerates the forward's draw
struct IbAttnDrop {
  uint32_t thr;            // drop when the 32-bit draw is below p * 2^32
  float keep;              // 1 / (1 - p)
  uint32_t seed;
  int32_t step;
  const int32_t* step_dev; // device-resident step counter (graph replay), or null
};
__device__ __forceinline__ uint32_t ib_attn_drop_key(const IbAttnDrop& a, int bh) {
  const uint32_t st = (uint32_t)(a.step_dev ? *a.step_dev : a.step);
  return ib_mix32(ib_mix32(a.seed ^ ib_mix32(st + 0x9e3779b9u)) + (uint32_t)bh * 0x85ebca6bu);
}
__device__ __forceinline__ float ib_attn_drop_mult(const IbAttnDrop& a, uint32_t key, int q, int j) {   // q, j < 2^16
  return ib_mix32((((uint32_t)q << 16) | (uint32_t)j) ^ key) < a.thr ? 0.f : a.keep;
}

// XCD-aware bijective remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous chunk of the logical grid -> neighbouring tiles that
// share an operand panel hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int ib_xcd_remap(int bid, int nwg) {
  const int nx = 8;
  if (nwg < nx * 2) return bid;
  int xcd = bid % nx, q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / nx;
}

static inline int ib_grid_1d(int64_t work_items, int per_block, int cap = 256 * 8) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

// sum of `nslab` float4 slabs (element stride st4 between slabs), added strictly in slab order from +0: batches of up to 8
// slabs are REQUESTED together (predicated on the count).  A rolled one-slab-per-trip loop -- or a 4-wide loop with a
// rolled tail -- is one memory round trip per trip in sequence; the slab counts of the split-M weight gradients are 2, 8
// or 10, so the tails were most of those launches.
__device__ __forceinline__ float4 ib_slab_sum4(const float4* __restrict__ q, int64_t st4, int nslab) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < nslab; k0 += 8) {
    float4 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k0 + e < nslab) v[e] = q[(int64_t)(k0 + e) * st4];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k0 + e < nslab) { s.x += v[e].x; s.y += v[e].y; s.z += v[e].z; s.w += v[e].w; }
  }
  return s;
}

// partial column sum of one float4 column over rows r0, r0 + step, ... (< rows) of a row-major array (pitch ld), added
// strictly in that order from +0; batches of up to 8 rows requested together (see ib_slab_sum4)
__device__ __forceinline__ float4 ib_rows_sum4(const float* __restrict__ base, int64_t ld, int r0, int step, int rows) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = r0; r < rows; r += 8 * step) {
    float4 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (r + e * step < rows) v[e] = *reinterpret_cast<const float4*>(base + (int64_t)(r + e * step) * ld);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (r + e * step < rows) { s.x += v[e].x; s.y += v[e].y; s.z += v[e].z; s.w += v[e].w; }
  }
  return s;
}
