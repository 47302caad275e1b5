// Linear-layer GEMMs for gfx950: one LDS-tiled MFMA kernel family, three operand layouts.
//
//   C[i][j] = sum_k A(i,k) * B(j,k)            128 x 128 output tile per 256-thread workgroup,
//                                              4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16 tiles
//   fwd   : A = x  [M,K]  (k-contiguous), B = w  [N,K] (k-contiguous)     y  = act(x w^T + ...)
//   dgrad : A = dz [M,N]  (k-contiguous), B = w  [N,K] (k-STRIDED)        dx = (dz w) * act'(aux)
//   wgrad : A = dz [M,N]  (k-STRIDED)   , B = x  [M,K] (k-STRIDED)        dw = dz^T x   (split over M)
//
// dtype f32  : v_mfma_f32_16x16x4_f32  (exact f32 fma chain -> the parity mode)
// dtype bf16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate (the throughput mode)
//
// LDS images keep the GLOBAL orientation of each operand (so staging is always 16-byte pieces,
// coalesced along the contiguous dimension): k-contiguous operands as [row][k], k-strided operands
// as [k][row].  bf16 fragments of a [k][row] image are read with ds_read_b64_tr_b16 (the hardware
// transposing read), everything else with ds_read_b128 / ds_read_b32.
// The MFMA is issued "swapped" (a = B-tile rows, b = A-tile rows) so that each lane ends up with
// 4 CONSECUTIVE output columns of one output row -> 8/16-byte epilogue stores and bias loads.
#include <cstdlib>
#include <type_traits>

#include "ib_common.h"
#include "gemm_nt.h"
#include "time_bwd.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int OPER_BYTES = 18432;  // one operand tile image (max over layouts / dtypes)

template <typename T> struct Tile;
template <> struct Tile<float> {
  static constexpr int BK = 32, VW = 4, KC_STRIDE = 34, KS_STRIDE = 144;
};
template <> struct Tile<bf16_t> {
  static constexpr int BK = 64, VW = 8, KC_STRIDE = 72, KS_STRIDE = 144;
};

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_WGRAD = 2 };

// TIMING-ONLY ablation switch (tools/kbench.py --ablate): bit0 skip steady-state global loads, bit1 skip LDS
// stores, bit2 skip MFMAs, bit3 skip the epilogue stores.  Results are wrong when non-zero; never set by product code.
#ifdef IB_AB
int g_ablate = 0;
long long* g_gemm_prof = nullptr;   // TIMING-ONLY (tools/gemm_prof.py): [workgroups][8] wall-clock stamps of the ring kernel
#define IB_ABLATE g_ablate
#else
#define IB_ABLATE 0
#endif

struct GemmParams {
  const void* A; const void* B; int64_t lda, ldb;
  int M, N, K;                 // C is [M,N]; K is the reduction length
  void* C; int64_t ldc;
  void* Z; int64_t ldz;        // fwd: optional pre-activation output
  const float* bias;
  const void* add_div; int64_t ld_add_div;
  const void* add_mod; int64_t ld_add_mod;
  int seg;
  const void* aux; int64_t ldaux;  // dgrad: activation-derivative operand
  const void* addend; int64_t ldadd;  // dgrad: residual-path gradient added in the epilogue
  int act;
  int vecA, vecB, vecC;
  int gldsA, gldsB;            // operand pieces are 16-byte aligned: direct-to-LDS staging is legal (bf16)
  int vecBias, vecAdd, vecAux;   // 16-byte (fp32) / 8-byte (bf16) epilogue operand loads are legal
  int tiles_m, tiles_n;
  int xcd_group;               // give each XCD a contiguous run of logical block ids
  int k_chunk;                 // split over the reduction (wgrad): blockIdx.y * k_chunk
  int64_t slab_stride;         // wgrad: elements between partial slabs (0 when not split)
  int accumulate;
  int ablate;
  long long* prof;
  float* dbias_part;           // wgrad (ring kernel): optional [splits][M] fp32 per-slice row sums of A (= the bias gradient's
                               // split-M partial sums: sum over the slice's rows of dz), written by the column-tile-0 blocks
};

template <typename T>
__device__ __forceinline__ uint4 load_piece(const T* p, int nvalid, bool vec) {
  constexpr int VW = Tile<T>::VW;
  uint4 r = make_uint4(0u, 0u, 0u, 0u);
  if (nvalid >= VW && vec) {
    __builtin_memcpy(&r, __builtin_assume_aligned(p, 4), 16);
  } else if (nvalid > 0) {
    T tmp[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) tmp[e] = static_cast<T>(0.f);
#pragma unroll
    for (int e = 0; e < VW; ++e)
      if (e < nvalid) tmp[e] = p[e];
    __builtin_memcpy(&r, tmp, 16);
  }
  return r;
}

// global -> registers: 4 x 16-byte pieces per thread per operand tile
template <typename T, bool KC>
__device__ __forceinline__ void load_tile(const T* base, int64_t ld, int row0, int rows_total, int k0,
                                          int k_end, bool vec, uint4 (&r)[4], int tid) {
  constexpr int VW = Tile<T>::VW;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = tid + NTHREADS * q;
    if (KC) {
      const int row = p >> 3, kc = p & 7;
      const int gr = row0 + row, gk = k0 + kc * VW;
      const int nvalid = (gr < rows_total) ? min(VW, k_end - gk) : 0;
      r[q] = load_piece<T>(base + (int64_t)gr * ld + gk, nvalid, vec);
    } else {
      constexpr int RG = 128 / VW;
      const int kk = p / RG, rg = p % RG;
      const int gk = k0 + kk, gr = row0 + rg * VW;
      const int nvalid = (gk < k_end) ? min(VW, rows_total - gr) : 0;
      r[q] = load_piece<T>(base + (int64_t)gk * ld + gr, nvalid, vec);
    }
  }
}

// Hoisted form for the steady state: per-thread piece pointers are computed once per workgroup and
// advanced by a constant per K step; no predicates (k-contiguous operands clamp out-of-range rows to the
// last valid row -- those tile rows only feed outputs that are never stored; k-strided operands use this
// path with a per-piece row count: full pieces load 16 bytes, edge pieces are zero-filled element-wise).
template <typename T, bool KC>
__device__ __forceinline__ void init_ptrs(const T* base, int64_t ld, int row0, int rows_total, int k0,
                                          const T* (&ptr)[4], int (&nvr)[4], int tid) {
  constexpr int VW = Tile<T>::VW;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = tid + NTHREADS * q;
    if (KC) {
      const int row = p >> 3, kc = p & 7;
      const int gr = min(row0 + row, rows_total - 1);
      ptr[q] = base + (int64_t)gr * ld + k0 + kc * VW;
      nvr[q] = VW;
    } else {
      constexpr int RG = 128 / VW;
      const int kk = p / RG, rg = p % RG;
      const int gr = row0 + rg * VW;
      ptr[q] = base + (int64_t)(k0 + kk) * ld + gr;
      nvr[q] = max(0, min(VW, rows_total - gr));   // rows of this piece inside the matrix (loop-invariant)
    }
  }
}
template <typename T, bool KC>
__device__ __forceinline__ void load_tile_fast(const T* const (&ptr)[4], const int (&nvr)[4], int64_t off,
                                               uint4 (&r)[4]) {
  constexpr int VW = Tile<T>::VW;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (KC || nvr[q] == VW) {
      __builtin_memcpy(&r[q], __builtin_assume_aligned(ptr[q] + off, 4), 16);
    } else {
      r[q] = load_piece<T>(ptr[q] + off, nvr[q], false);   // edge piece of a k-strided operand: zero-filled
    }
  }
}

// registers -> LDS image
template <typename T, bool KC>
__device__ __forceinline__ void store_tile(unsigned char* tile, const uint4 (&r)[4], int tid) {
  constexpr int VW = Tile<T>::VW;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = tid + NTHREADS * q;
    if (KC) {
      const int row = p >> 3, kc = p & 7;
      const int byte = (row * Tile<T>::KC_STRIDE + kc * VW) * (int)sizeof(T);
      if (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(tile + byte) = r[q];
      } else {  // 136-byte rows: only 8-byte aligned
        *reinterpret_cast<uint2*>(tile + byte) = make_uint2(r[q].x, r[q].y);
        *reinterpret_cast<uint2*>(tile + byte + 8) = make_uint2(r[q].z, r[q].w);
      }
    } else {
      constexpr int RG = 128 / VW;
      const int kk = p / RG, rg = p % RG;
      const int byte = (kk * Tile<T>::KS_STRIDE + rg * VW) * (int)sizeof(T);
      *reinterpret_cast<uint4*>(tile + byte) = r[q];
    }
  }
}

// ---- direct-to-LDS staging (bf16): global_load_lds_dwordx4 writes 16 B per lane straight into the LDS
// image (no VGPR round trip, no ds_write: VGPR->LDS stores run at ~80 B/clk/CU and were ~1/3 of a K step).
// One wave-instruction fills 1 KiB of the image at (wave-uniform base + lane*16), so the 18 KiB image
// (128 rows x 144 B, or 64 k-rows x 288 B) is 18 chunks; wave w issues chunks w, w+4, ...  Each lane derives
// WHICH global piece belongs at its LDS slot from the slot's byte offset; lanes that fall into the row padding
// fetch a duplicate piece (the pad bytes are never read).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
constexpr int GLDS_CHUNKS = OPER_BYTES / 1024;   // 18

template <bool KC>
__device__ __forceinline__ void glds_offsets(int64_t ld, int row0, int rows_total, int wave, int lane, int64_t (&off)[5]) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int chunk = wave + 4 * j;
    const int o = (chunk < GLDS_CHUNKS ? chunk : 0) * 1024 + lane * 16;
    if (KC) {
      const int row = o / 144, c = min((o % 144) / 16, 7);
      off[j] = (int64_t)min(row0 + row, rows_total - 1) * ld + 8 * c;
    } else {
      const int kk = o / 288, c = min((o % 288) / 16, 15);
      off[j] = (int64_t)kk * ld + row0 + 8 * c;
    }
  }
}
__device__ __forceinline__ void glds_issue(const bf16_t* base, const int64_t (&off)[5], int64_t koff, unsigned char* tile,
                                           int wave) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int chunk = wave + 4 * j;
    if (chunk < GLDS_CHUNKS)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(base + off[j] + koff), (lds_void_t*)(tile + chunk * 1024), 16, 0, 0);
  }
}

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// bf16 fragment: 8 consecutive k of tile row (rbase + lane%16), k = kbase + 8*(lane/16) + j
template <bool KC, int KCS = Tile<bf16_t>::KC_STRIDE>
__device__ __forceinline__ bf16x8_t read_frag_bf16(const unsigned char* tile, int rbase, int kbase, int lane) {
  if (KC) {
    const int off = ((rbase + (lane & 15)) * KCS + kbase + 8 * (lane >> 4)) * 2;
    return *reinterpret_cast<const bf16x8_t*>(tile + off);
  } else {
    // [k][row] image; transposing read: lane 4q+p of a 16-lane group addresses row (k) q,
    // columns 4p..4p+3, and receives column (lane%16) of the 4 rows.
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int kr = kbase + 8 * (lane >> 4) + q;
    const bf16_t* a0 = reinterpret_cast<const bf16_t*>(tile) + kr * Tile<bf16_t>::KS_STRIDE + rbase + 4 * pp;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0 + 4 * Tile<bf16_t>::KS_STRIDE));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

template <bool KC>
__device__ __forceinline__ float read_frag_f32(const unsigned char* tile, int rbase, int kbase, int lane) {
  const float* t = reinterpret_cast<const float*>(tile);
  if (KC) return t[(rbase + (lane & 15)) * Tile<float>::KC_STRIDE + kbase + (lane >> 4)];
  return t[(kbase + (lane >> 4)) * Tile<float>::KS_STRIDE + rbase + (lane & 15)];
}

template <typename T, bool A_KC, bool B_KC>
__device__ __forceinline__ void compute_tile(const unsigned char* tA, const unsigned char* tB,
                                             f32x4_t (&acc)[4][4], int lane, int wi, int wj) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = read_frag_bf16<A_KC>(tA, wi * 64 + 16 * t, ks * 32, lane);
#pragma unroll
      for (int u = 0; u < 4; ++u) fb[u] = read_frag_bf16<B_KC>(tB, wj * 64 + 16 * u, ks * 32, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[t], acc[t][u], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < Tile<float>::BK / 4; ++kk) {
      float fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = read_frag_f32<A_KC>(tA, wi * 64 + 16 * t, kk * 4, lane);
#pragma unroll
      for (int u = 0; u < 4; ++u) fb[u] = read_frag_f32<B_KC>(tB, wj * 64 + 16 * u, kk * 4, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[u], fa[t], acc[t][u], 0, 0, 0);
    }
  }
}

template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4], int nv, bool vec) {
  if (nv == 4 && vec) {
    if constexpr (sizeof(T) == 2) {
      bf16x4_t o;
      o[0] = static_cast<bf16_t>(v[0]); o[1] = static_cast<bf16_t>(v[1]);
      o[2] = static_cast<bf16_t>(v[2]); o[3] = static_cast<bf16_t>(v[3]);
      *reinterpret_cast<bf16x4_t*>(p) = o;
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < nv) p[r] = ib_from_f32<T>(v[r]);
  }
}

// activation math with a compile-time selector.  fp32 storage (parity mode) uses the accurate libm
// forms; bf16 storage uses the hardware exp (v_exp_f32) -- its error is far below bf16 resolution.
template <typename T, int ACT>
__device__ __forceinline__ float act_fwd_t(float v) {
  if constexpr (ACT == IB_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return tanhf(v);
  else if constexpr (ACT == IB_ACT_SIGMOID) return 1.f / (1.f + (sizeof(T) == 2 ? __expf(-v) : expf(-v)));
  else if constexpr (ACT == IB_ACT_SILU) return v / (1.f + (sizeof(T) == 2 ? __expf(-v) : expf(-v)));
  else if constexpr (ACT == IB_ACT_ELU) return v > 0.f ? v : (sizeof(T) == 2 ? __expf(v) : expf(v)) - 1.f;
  else return v;
}
template <typename T, int ACT>
__device__ __forceinline__ float act_bwd_t(float aux) {
  if constexpr (ACT == IB_ACT_RELU) return aux > 0.f ? 1.f : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return 1.f - aux * aux;
  else if constexpr (ACT == IB_ACT_SIGMOID) return aux * (1.f - aux);
  else if constexpr (ACT == IB_ACT_ELU) return aux > 0.f ? 1.f : aux + 1.f;
  else if constexpr (ACT == IB_ACT_SILU) {
    const float sg = 1.f / (1.f + (sizeof(T) == 2 ? __expf(-aux) : expf(-aux)));
    return sg * (1.f + aux * (1.f - sg));
  } else return 1.f;
}

template <typename T>
__device__ __forceinline__ void load4f(const T* p, int nv, bool vec, float (&v)[4]) {
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
    for (int r = 0; r < 4; ++r) v[r] = (r < nv) ? ib_to_f32(p[r]) : 0.f;
  }
}

// raw 4-element operand piece (8 B bf16 / 16 B fp32) kept in registers until all loads are in flight
template <typename T> struct Raw4 { typename std::conditional<sizeof(T) == 2, uint2, uint4>::type v; };
template <typename T>
__device__ __forceinline__ Raw4<T> ldraw4(const T* __restrict__ p, int nv, bool vec) {
  Raw4<T> r;
  if (nv == 4 && vec) {
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
  } else {
    T tmp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tmp[e] = (e < nv) ? p[e] : static_cast<T>(0.f);
    __builtin_memcpy(&r.v, tmp, sizeof(r.v));
  }
  return r;
}
template <typename T>
__device__ __forceinline__ void unraw4(const Raw4<T>& r, float (&v)[4]) {
  T tmp[4];
  __builtin_memcpy(tmp, &r.v, sizeof(r.v));
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = ib_to_f32(tmp[e]);
}

// Epilogue: a lane holds C[i][jb..jb+3] for 16 (t,u) sub-tiles.  Phase 1 issues EVERY operand load (bias,
// row-broadcast adds, activation-derivative operand, residual) so they are all in flight together; phase 2
// does the math and the stores.  (A per-sub-tile load->math->store chain serialises 16 L2 round trips.)
template <typename T, int EPI, int ACT>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4_t (&acc)[4][4], int i0, int j0, int wi, int wj,
                                         int lane, int split_id) {
  const int jb0 = j0 + wj * 64 + 4 * (lane >> 4);
  const int ib0 = i0 + wi * 64 + (lane & 15);
  [[maybe_unused]] Raw4<float> rb[4];
  [[maybe_unused]] Raw4<T> r1[4][4], r2[4][4];
  [[maybe_unused]] bool h1 = false, h2 = false;
  if constexpr (EPI == EPI_FWD) {
    h1 = p.add_div != nullptr; h2 = p.add_mod != nullptr;
    if (p.bias) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jb = jb0 + 16 * u;
        rb[u] = ldraw4<float>(p.bias + jb, max(0, min(4, p.N - jb)), p.vecBias);
      }
    }
  } else if constexpr (EPI == EPI_DGRAD) {
    h1 = (ACT != IB_ACT_NONE); h2 = p.addend != nullptr;
  }
  if constexpr (EPI != EPI_WGRAD) {
    if (h1 || h2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = min(ib0 + 16 * t, p.M - 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jb = jb0 + 16 * u;
          const int nv = max(0, min(4, p.N - jb));
          if constexpr (EPI == EPI_FWD) {
            if (h1) r1[t][u] = ldraw4<T>(reinterpret_cast<const T*>(p.add_div) + (int64_t)(i / p.seg) * p.ld_add_div + jb, nv, p.vecAdd);
            if (h2) r2[t][u] = ldraw4<T>(reinterpret_cast<const T*>(p.add_mod) + (int64_t)(i % p.seg) * p.ld_add_mod + jb, nv, p.vecAdd);
          } else {
            if (h1) r1[t][u] = ldraw4<T>(reinterpret_cast<const T*>(p.aux) + (int64_t)i * p.ldaux + jb, nv, p.vecAux);
            if (h2) r2[t][u] = ldraw4<T>(reinterpret_cast<const T*>(p.addend) + (int64_t)i * p.ldadd + jb, nv, p.vecAdd);
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = ib0 + 16 * t;
    if (i >= p.M) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jb = jb0 + 16 * u;
      if (jb >= p.N) continue;
      const int nv = min(4, p.N - jb);
      float v[4] = {acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]};
      float w4[4];
      if constexpr (EPI == EPI_FWD) {
        if (p.bias) {
          unraw4<float>(rb[u], w4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += w4[r];
        }
        if (h1) {
          unraw4<T>(r1[t][u], w4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += w4[r];
        }
        if (h2) {
          unraw4<T>(r2[t][u], w4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += w4[r];
        }
        if (p.Z) store4<T>(reinterpret_cast<T*>(p.Z) + (int64_t)i * p.ldz + jb, v, nv, p.vecC);
        if constexpr (ACT != IB_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_fwd_t<T, ACT>(v[r]);
        }
        store4<T>(reinterpret_cast<T*>(p.C) + (int64_t)i * p.ldc + jb, v, nv, p.vecC);
      } else if constexpr (EPI == EPI_DGRAD) {
        if constexpr (ACT != IB_ACT_NONE) {
          unraw4<T>(r1[t][u], w4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= act_bwd_t<T, ACT>(w4[r]);
        }
        if (h2) {
          unraw4<T>(r2[t][u], w4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += w4[r];
        }
        store4<T>(reinterpret_cast<T*>(p.C) + (int64_t)i * p.ldc + jb, v, nv, p.vecC);
      } else {
        float* c = reinterpret_cast<float*>(p.C) + (int64_t)split_id * p.slab_stride + (int64_t)i * p.ldc + jb;
        if (p.accumulate && p.slab_stride == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) v[r] += c[r];
        }
        store4<float>(c, v, nv, p.vecC);
      }
    }
  }
}

template <typename T, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmParams p) {
  constexpr int BK = Tile<T>::BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * OPER_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch), so after the remap a run of
  // consecutive logical ids sits on one XCD.  fwd/dgrad: the tiles_n column tiles of one row panel; wgrad: ALL
  // output tiles of one M-chunk (they re-read the same dz / x chunk -- without this the chunk was fetched by up
  // to 8 L2s: 79 MB fetched for 21 MB algorithmic in the first profile).  Speed only, never correctness.
  const int bid = p.xcd_group ? ib_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tiles_total = p.tiles_m * p.tiles_n;
  const int split_id = bid / tiles_total, tile_id = bid % tiles_total;
  const int ti = tile_id / p.tiles_n, tj = tile_id % p.tiles_n;
  const int i0 = ti * BM, j0 = tj * BN;
  const int kb = split_id * p.k_chunk;
  const int ke = min(p.K, kb + p.k_chunk);
  const int nk = (ke - kb + BK - 1) / BK;
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);

  f32x4_t acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fast (unpredicated, pointer-bumped) staging where legal, generic predicated staging elsewhere
  const bool fastA = p.vecA, fastB = p.vecB;
  const T* pa[4]; const T* pb[4];
  int na[4], nb[4];
  init_ptrs<T, A_KC>(A, p.lda, i0, p.M, kb, pa, na, tid);
  init_ptrs<T, B_KC>(B, p.ldb, j0, p.N, kb, pb, nb, tid);
  const int64_t stepA = A_KC ? BK : (int64_t)BK * p.lda;
  const int64_t stepB = B_KC ? BK : (int64_t)BK * p.ldb;

  // Double-buffered LDS, one barrier per K step.  Per operand and K step the tile is staged either
  //  (g) directly HBM/L2 -> LDS by global_load_lds (bf16, 16-byte-aligned pieces, full K step, and -- for
  //      k-strided operands -- an interior tile), issued BEFORE the MFMA phase of the previous tile, or
  //  (r) through registers (predicated / zero-filled: K tails, edge tiles, unaligned or fp32 operands), loaded
  //      before the MFMA phase and written to LDS after it.
  // (A depth-2 register prefetch was measured slower for the k-strided operands: wgrad +25 %, dgrad +10 %.)
  bool gA = false, gB = false;
  [[maybe_unused]] int64_t oa[5], ob[5];
  if constexpr (sizeof(T) == 2) {
    // measured (tools/kbench.py): direct-to-LDS staging pays for long reductions (K = 2048 forward: 48 -> 41 us)
    // and is neutral-to-slightly-negative below ~16 K steps, where prologue/epilogue dominate
    const bool longk = nk >= 16;
    gA = longk && p.gldsA && (A_KC || i0 + BM <= p.M);
    gB = longk && p.gldsB && (B_KC || j0 + BN <= p.N);
    if (gA) glds_offsets<A_KC>(p.lda, i0, p.M, wave, lane, oa);
    if (gB) glds_offsets<B_KC>(p.ldb, j0, p.N, wave, lane, ob);
  }
  uint4 ra[4], rb[4];
  // returns bit0: A staged through registers, bit1: B staged through registers
  auto stage_issue = [&](int kt, unsigned char* tA, unsigned char* tB) -> int {
    const int k0 = kb + kt * BK;
    const bool fullk = (k0 + BK <= ke);
    int viaregs = 0;
    if constexpr (sizeof(T) == 2) {
      if (gA && fullk) glds_issue(reinterpret_cast<const bf16_t*>(A), oa, A_KC ? (int64_t)k0 : (int64_t)k0 * p.lda, tA, wave);
      else viaregs |= 1;
      if (gB && fullk) glds_issue(reinterpret_cast<const bf16_t*>(B), ob, B_KC ? (int64_t)k0 : (int64_t)k0 * p.ldb, tB, wave);
      else viaregs |= 2;
    } else {
      viaregs = 3;
    }
    if (viaregs & 1) {
      if (fastA && fullk) load_tile_fast<T, A_KC>(pa, na, stepA * kt, ra);
      else load_tile<T, A_KC>(A, p.lda, i0, p.M, k0, ke, p.vecA, ra, tid);
    }
    if (viaregs & 2) {
      if (fastB && fullk) load_tile_fast<T, B_KC>(pb, nb, stepB * kt, rb);
      else load_tile<T, B_KC>(B, p.ldb, j0, p.N, k0, ke, p.vecB, rb, tid);
    }
    return viaregs;
  };
  auto stage_commit = [&](int viaregs, unsigned char* tA, unsigned char* tB) {
    if (viaregs & 1) store_tile<T, A_KC>(tA, ra, tid);
    if (viaregs & 2) store_tile<T, B_KC>(tB, rb, tid);
  };
  if (nk > 0) {
    const int v = stage_issue(0, smem, smem + OPER_BYTES);
    stage_commit(v, smem, smem + OPER_BYTES);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* tA = smem + (kt & 1) * 2 * OPER_BYTES;
    unsigned char* tB = tA + OPER_BYTES;
    unsigned char* nA = smem + ((kt + 1) & 1) * 2 * OPER_BYTES;
    unsigned char* nB = nA + OPER_BYTES;
    int v = 0;
#ifdef IB_ABLATE
    if (kt + 1 < nk && !(p.ablate & 1)) v = stage_issue(kt + 1, nA, nB);
    if (!(p.ablate & 4)) compute_tile<T, A_KC, B_KC>(tA, tB, acc, lane, wi, wj);
    if (kt + 1 < nk && !(p.ablate & 2)) stage_commit(v, nA, nB);
#else
    if (kt + 1 < nk) v = stage_issue(kt + 1, nA, nB);
    compute_tile<T, A_KC, B_KC>(tA, tB, acc, lane, wi, wj);
    if (kt + 1 < nk) stage_commit(v, nA, nB);
#endif
    __syncthreads();   // also drains the in-flight global_load_lds (hipcc emits vmcnt(0) before the barrier)
  }
#ifdef IB_ABLATE
  if (p.ablate & 8) return;
#endif

  // ---- epilogue: lane holds C[i][jb..jb+3] for 16 (t,u) sub-tiles.  The activation is a compile-time
  // parameter of the epilogue body (one uniform switch here), so only the selected math is executed.
  if constexpr (EPI == EPI_WGRAD) {
    epilogue<T, EPI, IB_ACT_NONE>(p, acc, i0, j0, wi, wj, lane, split_id);
    return;
  }
  switch (p.act) {
    case IB_ACT_RELU: epilogue<T, EPI, IB_ACT_RELU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_TANH: epilogue<T, EPI, IB_ACT_TANH>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_SIGMOID: epilogue<T, EPI, IB_ACT_SIGMOID>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_SILU: epilogue<T, EPI, IB_ACT_SILU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_ELU: epilogue<T, EPI, IB_ACT_ELU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    default: epilogue<T, EPI, IB_ACT_NONE>(p, acc, i0, j0, wi, wj, lane, split_id); break;
  }
}

// ---- ring-pipelined variant (bf16, every operand piece 16-byte aligned, reduction a multiple of 32) -------------
// The kernel above stages ONE K step ahead and its __syncthreads() drains the in-flight loads (vmcnt(0)): every K
// step exposes a full L2/HBM round trip, ~1.2 us per 64-deep step at these sizes.  Here the operands go HBM/L2 -> LDS
// by global_load_lds only, into a ring of four 32-deep stages: three stages (2 x 54 KiB per CU at 2 workgroups/CU) are
// in flight while one is consumed, each wave waits for ITS OWN pieces of the stage with a counted s_waitcnt vmcnt(N),
// and the workgroup barrier is the raw s_barrier (no vmcnt(0) drain).  One barrier per K step:
//   wait(stage kt landed, mine) ; barrier (everyone's landed; everyone is done reading stage kt-1)
//   issue stage kt+3 into the slot stage kt-1 occupied ; MFMAs on stage kt.
// Out-of-range tile rows/columns are CLAMPED onto valid memory (they only feed outputs that are never stored); the
// reduction range itself must be exact (no zero fill with LDS-DMA), so ragged reductions use the kernel above.
namespace ring {
constexpr int BK = 32, NS = 4, KCS = 40;                   // KC image: 128 rows x (32 + 8 pad) bf16 = 10 chunks of 1 KiB
constexpr int KC_BYTES = 128 * KCS * 2, KS_BYTES = BK * Tile<bf16_t>::KS_STRIDE * 2;   // 10240 / 9216
template <bool KC> constexpr int chunks() { return (KC ? KC_BYTES : KS_BYTES) / 1024; }

// per-lane global element offsets of the (up to 3) chunks wave `w` stages for one operand tile
template <bool KC>
__device__ __forceinline__ void offsets(int64_t ld, int row0, int rows_total, int wave, int lane, int64_t (&off)[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int chunk = wave + 4 * j;
    const int o = (chunk < chunks<KC>() ? chunk : 0) * 1024 + lane * 16;
    if (KC) {
      const int row = o / (KCS * 2), c = min((o % (KCS * 2)) / 16, 3);
      off[j] = (int64_t)min(row0 + row, rows_total - 1) * ld + 8 * c;
    } else {
      const int kk = o / 288, c = min((o % 288) / 16, 15);
      off[j] = (int64_t)kk * ld + min((int64_t)row0 + 8 * c, ld - 8);
    }
  }
}
template <bool KC>
__device__ __forceinline__ void issue(const bf16_t* base, const int64_t (&off)[3], int64_t koff, unsigned char* tile, int wave) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int chunk = wave + 4 * j;
    if (chunk < chunks<KC>())
      __builtin_amdgcn_global_load_lds((glb_void_t*)(base + off[j] + koff), (lds_void_t*)(tile + chunk * 1024), 16, 0, 0);
  }
}
template <bool KC> __device__ __forceinline__ int per_wave(int wave) { return (chunks<KC>() - wave + 3) / 4; }

// LDS fragment reads as inline asm: hipcc orders every LDS read it can see behind ALL outstanding LDS-DMA writes
// (s_waitcnt vmcnt(0) before the first ds_read of the K step), which would drain the ring; reads it cannot see are
// ordered by hand -- the counted vmcnt + barrier above them, one lgkmcnt(0) (frags_ready) below them.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
template <bool KC>
__device__ __forceinline__ bf16x8_t read_frag_asm(const unsigned char* tile, int rbase, int lane) {
  if (KC) {
    const unsigned a = lds_addr(tile) + ((rbase + (lane & 15)) * KCS + 8 * (lane >> 4)) * 2;
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int kr = 8 * (lane >> 4) + q;
    const unsigned a = lds_addr(tile) + (kr * Tile<bf16_t>::KS_STRIDE + rbase + 4 * pp) * 2;
    u32x2_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * Tile<bf16_t>::KS_STRIDE * 2));
    u32x4_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8_t, v);
  }
}
// the fragments are the asm's in/out operands, so no consumer can be scheduled above the wait
__device__ __forceinline__ void frags_ready(bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform, one of {0, 4, 5, 6, 8, 10, 12}
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 8: wait_vm<8>(); break;
    case 10: wait_vm<10>(); break;
    default: wait_vm<12>(); break;
  }
}
}  // namespace ring

template <bool A_KC, bool B_KC>
constexpr int ring_lds_bytes() {
  return ring::NS * ((A_KC ? ring::KC_BYTES : ring::KS_BYTES) + (B_KC ? ring::KC_BYTES : ring::KS_BYTES));
}
// blk / nblk: this problem's block id and block count (== blockIdx.x / gridDim.x unless several problems share a launch)
template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_ring_body(const GemmParams& p, unsigned char* smem, int blk, int nblk) {
  using namespace ring;
  constexpr int A_BYTES = A_KC ? KC_BYTES : KS_BYTES, B_BYTES = B_KC ? KC_BYTES : KS_BYTES, STAGE = A_BYTES + B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int bid = p.xcd_group ? ib_xcd_remap(blk, nblk) : blk;
  const int tiles_total = p.tiles_m * p.tiles_n;
  const int split_id = bid / tiles_total, tile_id = bid % tiles_total;
  const int ti = tile_id / p.tiles_n, tj = tile_id % p.tiles_n;
  const int i0 = ti * BM, j0 = tj * BN;
  const int kb = split_id * p.k_chunk;
  const int ke = min(p.K, kb + p.k_chunk);
  const int nk = (ke - kb) / BK;                            // exact by the launch conditions
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  f32x4_t acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // weight gradient: the bias gradient rides along.  sum_k A(i,k) = one more MFMA per row tile against an all-ones
  // fragment (+25 % matrix work in the blocks of column tile 0, left wave column only): it replaces a column-sum launch
  // over the whole dz matrix per Linear layer.
  [[maybe_unused]] f32x4_t accb[4];
  [[maybe_unused]] bool do_bias = false;
  [[maybe_unused]] bf16x8_t ones;
  if constexpr (EPI == EPI_WGRAD) {
    do_bias = p.dbias_part != nullptr && tj == 0 && wj == 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) accb[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  }

#define RING_STAMP(k) do { if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
  RING_STAMP(0);
  int64_t oa[3], ob[3];
  offsets<A_KC>(p.lda, i0, p.M, wave, lane, oa);
  offsets<B_KC>(p.ldb, j0, p.N, wave, lane, ob);
  const int per = per_wave<A_KC>(wave) + per_wave<B_KC>(wave);   // this wave's LDS-DMA instructions per stage
  auto stage = [&](int kt) {
    unsigned char* tA = smem + (kt % NS) * STAGE;
    const int k0 = kb + kt * BK;
    issue<A_KC>(A, oa, A_KC ? (int64_t)k0 : (int64_t)k0 * p.lda, tA, wave);
    issue<B_KC>(B, ob, B_KC ? (int64_t)k0 : (int64_t)k0 * p.ldb, tA + A_BYTES, wave);
  };
  for (int s = 0; s < NS - 1 && s < nk; ++s) stage(s);
  RING_STAMP(1);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt(min(nk - 1 - kt, NS - 2) * per);             // my pieces of stage kt have landed
    __builtin_amdgcn_s_barrier();                            // everyone's have; everyone finished reading stage kt-1
    if (kt == 0) RING_STAMP(2);
    if (kt + NS - 1 < nk) stage(kt + NS - 1);
    const unsigned char* tA = smem + (kt % NS) * STAGE;
    const unsigned char* tB = tA + A_BYTES;
    bf16x8_t fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[t] = read_frag_asm<A_KC>(tA, wi * 64 + 16 * t, lane);
#pragma unroll
    for (int u = 0; u < 4; ++u) fb[u] = read_frag_asm<B_KC>(tB, wj * 64 + 16 * u, lane);
    frags_ready(fa, fb);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[t], acc[t][u], 0, 0, 0);
    if constexpr (EPI == EPI_WGRAD) {
      if (do_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[t], accb[t], 0, 0, 0);
      }
    }
  }
  RING_STAMP(3);
  if constexpr (EPI == EPI_WGRAD) {
    if (do_bias && (lane >> 4) == 0) {         // every output column holds the same sum: column 0 of each row tile
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = i0 + wi * 64 + 16 * t + (lane & 15);
        if (i < p.M) p.dbias_part[(int64_t)split_id * p.M + i] = accb[t][0];
      }
    }
    epilogue<bf16_t, EPI, IB_ACT_NONE>(p, acc, i0, j0, wi, wj, lane, split_id);
    RING_STAMP(4);
    return;
  }
  switch (p.act) {
    case IB_ACT_RELU: epilogue<bf16_t, EPI, IB_ACT_RELU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_TANH: epilogue<bf16_t, EPI, IB_ACT_TANH>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_SIGMOID: epilogue<bf16_t, EPI, IB_ACT_SIGMOID>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_SILU: epilogue<bf16_t, EPI, IB_ACT_SILU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    case IB_ACT_ELU: epilogue<bf16_t, EPI, IB_ACT_ELU>(p, acc, i0, j0, wi, wj, lane, split_id); break;
    default: epilogue<bf16_t, EPI, IB_ACT_NONE>(p, acc, i0, j0, wi, wj, lane, split_id); break;
  }
  RING_STAMP(4);
#undef RING_STAMP
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ring_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ring_lds_bytes<A_KC, B_KC>()];
  gemm_ring_body<A_KC, B_KC, EPI>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

// several weight-gradient problems in ONE launch (a training step's dW GEMMs are independent and each is short: one
// launch saves their kernel boundaries and the streams / joins that ran them side by side)
constexpr int WG_MAX = 6;
struct WgradMulti { GemmParams p[WG_MAX]; int blk0[WG_MAX + 1]; int n; };
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ring_wgrad_multi_kernel(WgradMulti m) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ring_lds_bytes<false, false>()];
  int e = 0;
  for (int j = 1; j < m.n; ++j)
    if ((int)blockIdx.x >= m.blk0[j]) e = j;
  gemm_ring_body<false, false, EPI_WGRAD>(m.p[e], smem, (int)blockIdx.x - m.blk0[e], m.blk0[e + 1] - m.blk0[e]);
}

// out[e] (+)= sum_s slab[s][e]   (fixed order -> bitwise reproducible).  float4 per thread, 4 slabs in flight.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_stride,
                                                          float* __restrict__ out, int64_t ldo, int rows, int cols,
                                                          int accumulate, int vec) {
  const int64_t n = (int64_t)rows * cols;
  if (vec) {   // cols % 4 == 0, slab_stride % 4 == 0, ldo % 4 == 0, 16-byte aligned bases
    const int64_t n4 = n >> 2;
    for (int64_t e4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e4 < n4; e4 += (int64_t)gridDim.x * blockDim.x) {
      const float4* p = reinterpret_cast<const float4*>(slabs) + e4;
      const int64_t st4 = slab_stride >> 2;
      float4 s = ib_slab_sum4(p, st4, nslab);
      const int64_t e = e4 << 2;
      const int r = (int)(e / cols), c0 = (int)(e % cols);
      float4* o = reinterpret_cast<float4*>(out + (int64_t)r * ldo + c0);
      if (accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
      *o = s;
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slabs[(int64_t)k * slab_stride + e];
    const int r = (int)(e / cols), c = (int)(e % cols);
    float* o = out + (int64_t)r * ldo + c;
    *o = accumulate ? (*o + s) : s;
  }
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename T> bool vec_load_ok(const void* p, int64_t ld) {
  return aligned(p, 4) && ((ld * (int64_t)sizeof(T)) % 4 == 0);
}
template <typename T> bool glds_ok(const void* p, int64_t ld) {
  return sizeof(T) == 2 && aligned(p, 16) && (ld % 8 == 0);
}
template <typename T> bool vec_store_ok(const void* p, int64_t ld) {
  return aligned(p, 4 * sizeof(T)) && (ld % 4 == 0);
}

// `group`: number of problems sharing the launch (ib_linear_wgrad_slabs_multi).  Each gets ~512 / group workgroups,
// i.e. the LAUNCH fills the chip twice over instead of every problem doing so: half the slab bytes written here and
// read back by the reduction (headline step, 4 problems: 0.2403 -> 0.2354 ms at 128 per problem; 96: 0.243, 64: 0.258)
int wgrad_split(int64_t M, int64_t N, int64_t K, int bk, int* chunk_out, int group = 1) {
  // reduction length is M; output tiles over [N, K]
  const int64_t tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  static const int env_target = ib_ab_int("IB_WGRAD_TARGET", 0);
  const int target = env_target ? env_target : (group <= 1 ? 256 : (group >= 4 ? 128 : 512 / group));
  // ~1 workgroup per CU: with the ring-pipelined main loop fewer, longer slices win (half the slab traffic; measured
  // step 0.310 -> 0.305 ms against 2 per CU).  IB_WGRAD_TARGET: tuning override.
  int64_t want = (target + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > 32) want = 32;
  int64_t chunk = (M + want - 1) / want;
  chunk = ((chunk + 63) / 64) * 64;  // multiple of both BKs
  if (chunk < 64) chunk = 64;
  const int64_t split = (M + chunk - 1) / chunk;
  (void)bk;
  *chunk_out = (int)chunk;
  return (int)split;
}

// ring kernel: bf16, every staged piece 16-byte aligned, exact reduction tiling, >= 2 K steps of 32
bool ring_ok(const GemmParams& p, int dtype, int red_len, int red_chunk) {
  static const int off = ib_ab_int("IB_NO_RING", 0);
  return !off && dtype == IB_BF16 && p.gldsA && p.gldsB && red_len % 32 == 0 && red_chunk % 32 == 0 && red_len >= 64 &&
         p.lda >= 8 && p.ldb >= 8;
}

// ---- small-M forward -------------------------------------------------------------------------------------------------
// y[M,N] = act(x[M,K] w[N,K]^T + bias) when the 128 x 128 tiling yields only a handful of workgroups (a batch of a few
// hundred rows: the reference's regression models, batch-1 analysis): 64 x 16 output tiles, and the four waves of a
// workgroup split the REDUCTION (k-blocks w, w+4, ...), so every wave walks a quarter of K.  Both operands are
// k-contiguous: fragments come straight from global memory (no LDS staging), the four partial accumulators are combined
// through LDS in a fixed order.  [256, 512, 1470]: 8 workgroups x 46 serial K steps -> 128 workgroups x 12.
namespace smallm {
constexpr int TM = 64, TN = 16, PD = 3;

__device__ __forceinline__ bf16x8_t ldfrag(const bf16_t* __restrict__ rowp, int ks, int K) {
  bf16x8_t v;
  const int nv = K - ks;
  if (nv >= 8) {
    __builtin_memcpy(&v, __builtin_assume_aligned(rowp + ks, 4), 16);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e < nv ? rowp[ks + e] : (bf16_t)0.f;
  }
  return v;
}

// NT = output-tile width / 16.  Only NT = 1 is instantiated: 64 x 64 tiles (NT = 4) were measured for the sampling loop's
// mid-sized problems (M = 3200, where 128 x 128 tiles leave most CUs idle) and lost to the ring kernel -- [3200, 1536,
// 512] 35.6 vs 17.6 us, [3200, 2048, 512] 48 vs 20 -- fragment loads straight from global memory are 16 rows x 64 bytes
// each and do not reach the rate of whole-row LDS-DMA staging once the tile count is in the hundreds.
template <int ACT, int NT>
__global__ __launch_bounds__(256) void fwd_smallm_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                         int64_t ldw, const float* __restrict__ bias,
                                                         const bf16_t* __restrict__ add_div, int64_t ld_ad,
                                                         const bf16_t* __restrict__ add_mod, int64_t ld_am, int seg,
                                                         bf16_t* __restrict__ y, int64_t ldy, int M, int N, int K) {
  constexpr int TNV = 16 * NT;
  __shared__ __attribute__((aligned(16))) float red[4][TM * TNV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TNV;
  const bf16_t* arow[4];
  const bf16_t* brow[NT];
#pragma unroll
  for (int t = 0; t < 4; ++t) arow[t] = x + (int64_t)min(i0 + 16 * t + r, M - 1) * ldx;
#pragma unroll
  for (int u = 0; u < NT; ++u) brow[u] = w + (int64_t)min(j0 + 16 * u + r, N - 1) * ldw;
  const int nkb = (K + 31) / 32;
  const int nmine = wave < nkb ? (nkb - wave + 3) / 4 : 0;
  f32x4_t acc[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[PD][4], fb[PD][NT];
  auto load = [&](int s, int it) {
    const int ks = (wave + 4 * it) * 32 + 8 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[s][t] = ldfrag(arow[t], ks, K);
#pragma unroll
    for (int u = 0; u < NT; ++u) fb[s][u] = ldfrag(brow[u], ks, K);
  };
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < nmine) load(s, s);
  for (int it0 = 0; it0 < nmine; it0 += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const int it = it0 + s;
      if (it < nmine) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < NT; ++u)
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s][u], fa[s][t], acc[t][u], 0, 0, 0);
        if (it + PD < nmine) load(s, it + PD);
      }
    }
  }
  // swapped operands: this lane holds columns 16*u + 4*kq .. +3 of row 16*t + r
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u)
      *reinterpret_cast<float4*>(&red[wave][(16 * t + r) * TNV + 16 * u + 4 * kq]) =
          make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
  __syncthreads();
  const bool vst = (ldy % 4) == 0 && (reinterpret_cast<uintptr_t>(y) % 8) == 0;
#pragma unroll
  for (int g = 0; g < NT; ++g) {                            // 256 threads x 4 outputs per pass
    const int o0 = (g * 256 + tid) * 4;
    const int row = o0 / TNV, c4 = o0 % TNV;
    const int gi = i0 + row, gj = j0 + c4;
    if (gi >= M || gj >= N) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = o0 + e;
      v[e] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
      if (gj + e < N) {
        if (bias) v[e] += bias[gj + e];
        if (add_div) v[e] += (float)add_div[(int64_t)(gi / seg) * ld_ad + gj + e];
        if (add_mod) v[e] += (float)add_mod[(int64_t)(gi % seg) * ld_am + gj + e];
      }
      v[e] = act_fwd_t<bf16_t, ACT>(v[e]);
    }
    bf16_t* dst = y + (int64_t)gi * ldy + gj;
    if (gj + 4 <= N && vst) {
      bf16x4_t o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x4_t*>(dst) = o4;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gj + e < N) dst[e] = (bf16_t)v[e];
    }
  }
}

// dx[M,K] = (dz[M,N] w[N,K]) * act'(aux): the same tiling for the backward.  w is k-STRIDED here (its rows are the
// reduction index), so the workgroup first transposes its slice w[:, 16 columns] into LDS (rows beyond N zero-filled);
// dz fragments come straight from global memory.
constexpr int RMAX = 1024, WS = RMAX + 8;
template <int ACT>
__global__ __launch_bounds__(256) void dgrad_smallm_kernel(const bf16_t* __restrict__ dz, int64_t lddz,
                                                           const bf16_t* __restrict__ w, int64_t ldw,
                                                           const bf16_t* __restrict__ aux, int64_t ldaux,
                                                           bf16_t* __restrict__ dx, int64_t lddx, int M, int N) {
  __shared__ __attribute__((aligned(16))) bf16_t wimg[16 * WS];
  __shared__ __attribute__((aligned(16))) float red[4][TM * TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
  const int nkb = (N + 31) / 32;
  for (int pce = tid; pce < nkb * 64; pce += 256) {        // one 16-byte piece = 8 columns of one reduction row
    const int k = pce >> 1, half = pce & 1;
    bf16x8_t v;
    if (k < N) __builtin_memcpy(&v, __builtin_assume_aligned(w + (int64_t)k * ldw + j0 + 8 * half, 16), 16);
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) wimg[(8 * half + e) * WS + k] = v[e];
  }
  __syncthreads();
  const bf16_t* arow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) arow[t] = dz + (int64_t)min(i0 + 16 * t + r, M - 1) * lddz;
  const bf16_t* brow = wimg + r * WS;
  const int nmine = wave < nkb ? (nkb - wave + 3) / 4 : 0;
  f32x4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[PD][4];
  auto load = [&](int s, int it) {
    const int ks = (wave + 4 * it) * 32 + 8 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[s][t] = ldfrag(arow[t], ks, N);
  };
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < nmine) load(s, s);
  for (int it0 = 0; it0 < nmine; it0 += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const int it = it0 + s;
      if (it < nmine) {
        bf16x8_t b;
        __builtin_memcpy(&b, __builtin_assume_aligned(brow + (wave + 4 * it) * 32 + 8 * kq, 16), 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa[s][t], acc[t], 0, 0, 0);
        if (it + PD < nmine) load(s, it + PD);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    *reinterpret_cast<float4*>(&red[wave][(16 * t + r) * TN + 4 * kq]) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  __syncthreads();
  const int row = tid >> 2, c4 = (tid & 3) * 4;
  const int gi = i0 + row, gj = j0 + c4;
  if (gi < M) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = row * TN + c4 + e;
      v[e] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    }
    if constexpr (ACT != IB_ACT_NONE) {
      bf16x4_t a4;
      __builtin_memcpy(&a4, __builtin_assume_aligned(aux + (int64_t)gi * ldaux + gj, 8), 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_bwd_t<bf16_t, ACT>((float)a4[e]);
    }
    bf16x4_t o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = (bf16_t)v[e];
    __builtin_memcpy(__builtin_assume_aligned(dx + (int64_t)gi * lddx + gj, 8), &o4, 8);
  }
}

// up to this many 128 x 128 tiles the small tiles win (Groundlink F=10 step: cap 32 0.528 ms, 64 0.489, 128 0.491,
// 256 0.502; the DDIM step at M = 3200, 100 tiles: 492 us of kernels at 32 / 64, 515 at 128)
inline int tile_cap() {
  static const int cap = ib_ab_int("IB_SMALLM_TILES", 64);
  return cap;
}
// GemmParams of a dgrad: A = dz [M, red], B = w [red, cols], C = dx [M, cols]; p.N = output columns, p.K = reduction length
inline bool dgrad_ok(const GemmParams& p) {
  static const int off = ib_ab_int("IB_NO_SMALLM", 0);
  const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  return !off && tiles <= tile_cap() && !p.addend && p.K >= 64 && p.K <= RMAX && p.N % 16 == 0 && p.lda % 2 == 0 && aligned(p.A, 4) &&
         p.ldb % 8 == 0 && aligned(p.B, 16) && p.ldc % 4 == 0 && aligned(p.C, 8) &&
         (p.act == IB_ACT_NONE || (p.aux && p.ldaux % 4 == 0 && aligned(p.aux, 8)));
}
int launch_dgrad(const GemmParams& p, hipStream_t s) {
  const dim3 grid((unsigned)(p.N / TN), (unsigned)((p.M + TM - 1) / TM)), block(256);
#define IB_SMALLM(ACT)                                                                                                    \
  hipLaunchKernelGGL((dgrad_smallm_kernel<ACT>), grid, block, 0, s, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.B, p.ldb,   \
                     (const bf16_t*)p.aux, p.ldaux, (bf16_t*)p.C, p.ldc, p.M, p.K)
  switch (p.act) {
    case IB_ACT_RELU: IB_SMALLM(IB_ACT_RELU); break;
    case IB_ACT_TANH: IB_SMALLM(IB_ACT_TANH); break;
    case IB_ACT_SIGMOID: IB_SMALLM(IB_ACT_SIGMOID); break;
    case IB_ACT_SILU: IB_SMALLM(IB_ACT_SILU); break;
    case IB_ACT_ELU: IB_SMALLM(IB_ACT_ELU); break;
    default: IB_SMALLM(IB_ACT_NONE); break;
  }
#undef IB_SMALLM
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// few 128 x 128 tiles (tile_cap()), plain epilogue (bias, the two row-broadcast addends, activation), rows 4-byte aligned
inline bool ok(const GemmParams& p) {
  static const int off = ib_ab_int("IB_NO_SMALLM", 0);
  const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  return !off && tiles <= tile_cap() && !p.Z && p.lda % 2 == 0 && p.ldb % 2 == 0 && aligned(p.A, 4) && aligned(p.B, 4) &&
         p.K >= 64;
}
template <int NT>
int launch(const GemmParams& p, hipStream_t s) {
  const dim3 grid((unsigned)((p.N + 16 * NT - 1) / (16 * NT)), (unsigned)((p.M + TM - 1) / TM)), block(256);
#define IB_SMALLM(ACT)                                                                                                      \
  hipLaunchKernelGGL((fwd_smallm_kernel<ACT, NT>), grid, block, 0, s, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.B, p.ldb,   \
                     p.bias, (const bf16_t*)p.add_div, p.ld_add_div, (const bf16_t*)p.add_mod, p.ld_add_mod, p.seg,         \
                     (bf16_t*)p.C, p.ldc, p.M, p.N, p.K)
  switch (p.act) {
    case IB_ACT_RELU: IB_SMALLM(IB_ACT_RELU); break;
    case IB_ACT_TANH: IB_SMALLM(IB_ACT_TANH); break;
    case IB_ACT_SIGMOID: IB_SMALLM(IB_ACT_SIGMOID); break;
    case IB_ACT_SILU: IB_SMALLM(IB_ACT_SILU); break;
    case IB_ACT_ELU: IB_SMALLM(IB_ACT_ELU); break;
    default: IB_SMALLM(IB_ACT_NONE); break;
  }
#undef IB_SMALLM
  IB_CHECK_LAUNCH();
  return IB_OK;
}
}  // namespace smallm

template <typename T>
int launch_fwd(GemmParams& p, hipStream_t s) {
  p.ablate = IB_ABLATE; p.prof = IB_AB_PROF(g_gemm_prof);
  p.vecA = vec_load_ok<T>(p.A, p.lda);
  p.vecB = vec_load_ok<T>(p.B, p.ldb);
  p.gldsA = glds_ok<T>(p.A, p.lda); p.gldsB = glds_ok<T>(p.B, p.ldb);
  p.vecC = vec_store_ok<T>(p.C, p.ldc) && (!p.Z || vec_store_ok<T>(p.Z, p.ldz));
  p.vecBias = !p.bias || aligned(p.bias, 16);
  p.vecAdd = (!p.add_div || vec_store_ok<T>(p.add_div, p.ld_add_div)) &&
             (!p.add_mod || vec_store_ok<T>(p.add_mod, p.ld_add_mod));
  p.tiles_n = (p.N + BN - 1) / BN; p.tiles_m = (p.M + BM - 1) / BM;
  p.k_chunk = p.K; p.slab_stride = 0; p.xcd_group = 1;
  const int tiles = p.tiles_m * p.tiles_n;
  if (sizeof(T) == 2 && smallm::ok(p)) { IB_PATH(IB_PATH_SMALLM); return smallm::launch<1>(p, s); }
  if (sizeof(T) == 2 && ring_ok(p, IB_BF16, p.K, p.K)) {
    IB_PATH(IB_PATH_RING);
    hipLaunchKernelGGL((gemm_ring_kernel<true, true, EPI_FWD>), dim3(tiles, 1), dim3(NTHREADS), 0, s, p);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  IB_PATH(IB_PATH_GENERIC);
  hipLaunchKernelGGL((gemm_kernel<T, true, true, EPI_FWD>), dim3(tiles, 1), dim3(NTHREADS), 0, s, p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

template <typename T>
int launch_dgrad(GemmParams& p, hipStream_t s) {
  p.ablate = IB_ABLATE; p.prof = IB_AB_PROF(g_gemm_prof);
  p.vecA = vec_load_ok<T>(p.A, p.lda);
  p.vecB = vec_load_ok<T>(p.B, p.ldb);
  p.gldsA = glds_ok<T>(p.A, p.lda); p.gldsB = glds_ok<T>(p.B, p.ldb);
  p.vecC = vec_store_ok<T>(p.C, p.ldc);
  p.vecAux = !p.aux || vec_store_ok<T>(p.aux, p.ldaux);
  p.vecAdd = !p.addend || vec_store_ok<T>(p.addend, p.ldadd);
  p.tiles_n = (p.N + BN - 1) / BN; p.tiles_m = (p.M + BM - 1) / BM;
  p.k_chunk = p.K; p.slab_stride = 0; p.xcd_group = 1;
  const int tiles = p.tiles_m * p.tiles_n;
  if (sizeof(T) == 2 && smallm::dgrad_ok(p)) { IB_PATH(IB_PATH_SMALLM); return smallm::launch_dgrad(p, s); }
  if (sizeof(T) == 2 && ring_ok(p, IB_BF16, p.K, p.K)) {
    IB_PATH(IB_PATH_RING);
    hipLaunchKernelGGL((gemm_ring_kernel<true, false, EPI_DGRAD>), dim3(tiles, 1), dim3(NTHREADS), 0, s, p);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  IB_PATH(IB_PATH_GENERIC);
  hipLaunchKernelGGL((gemm_kernel<T, true, false, EPI_DGRAD>), dim3(tiles, 1), dim3(NTHREADS), 0, s, p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

}  // namespace

#ifdef IB_AB
extern "C" int ib_debug_set_ablate(int mask) { g_ablate = mask; return IB_OK; }
extern "C" int ib_debug_set_gemm_prof(void* buf) { g_gemm_prof = reinterpret_cast<long long*>(buf); return IB_OK; }
#else       // measurement builds only (lib/ab/libib_hip_ab.so)
extern "C" int ib_debug_set_ablate(int) { return IB_E_UNSUPPORTED; }
extern "C" int ib_debug_set_gemm_prof(void*) { return IB_E_UNSUPPORTED; }
#endif

extern "C" int ib_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                             const void* add_div, int64_t ld_add_div, const void* add_mod,
                             int64_t ld_add_mod, int64_t seg, int act, void* y, int64_t ldy, void* z,
                             int64_t ldz, int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || ldy < N) return IB_E_ARG;
  if ((add_div || add_mod) && seg <= 0) return IB_E_ARG;
  if (z && ldz < N) return IB_E_ARG;
  if (act < IB_ACT_NONE || act > IB_ACT_ELU) return IB_E_ARG;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.B = w; p.ldb = ldw; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.C = y; p.ldc = ldy; p.Z = z; p.ldz = ldz; p.bias = bias;
  p.add_div = add_div; p.ld_add_div = ld_add_div; p.add_mod = add_mod; p.ld_add_mod = ld_add_mod;
  p.seg = (int)(seg > 0 ? seg : 1); p.act = act;
  if (dtype == IB_BF16 && !add_div && !add_mod && !z) {       // large-M training shapes: the 256 x 128 NT kernel
    const int rc = ib_gemm_nt_try(x, ldx, w, ldw, y, ldy, bias, act, nullptr, 0, IB_ACT_NONE, nullptr, 0, M, N, K, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) return rc;
  }
  if (dtype == IB_F32 && !add_div && !add_mod) {              // the reference's own batch sizes in fp32: reduction-split tiles
    const int rc = ib_f32_small_fwd_try((const float*)x, ldx, (const float*)w, ldw, bias, act, (float*)y, ldy, (float*)z, ldz,
                                        M, N, K, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) { IB_PATH(IB_PATH_SMALLM); return rc; }
  }
  if (dtype == IB_F32) return launch_fwd<float>(p, ib_s(stream));
  if (dtype == IB_BF16) return launch_fwd<bf16_t>(p, ib_s(stream));
  return IB_E_DTYPE;
}

extern "C" int ib_linear_dgrad(const void* dz, int64_t lddz, const void* w, int64_t ldw, int act_below,
                               const void* aux, int64_t ldaux, const void* addend, int64_t ldadd, void* dx,
                               int64_t lddx, int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream) {
  if (!dz || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldw < K || lddx < K) return IB_E_ARG;
  if (addend && ldadd < K) return IB_E_ARG;
  if (act_below != IB_ACT_NONE && (!aux || ldaux < K)) return IB_E_ARG;
  if (act_below < IB_ACT_NONE || act_below > IB_ACT_ELU) return IB_E_ARG;
  // C[M,K] = sum_n dz[m][n] * w[n][k]:  reduction length N, B(j=k, kk=n) = w[n*ldw + k]  (k-strided)
  GemmParams p{};
  p.A = dz; p.lda = lddz; p.B = w; p.ldb = ldw; p.M = (int)M; p.N = (int)K; p.K = (int)N;
  p.C = dx; p.ldc = lddx; p.aux = aux; p.ldaux = ldaux; p.act = act_below; p.seg = 1;
  p.addend = addend; p.ldadd = ldadd;
  if (dtype == IB_F32) {
    const int rc = ib_f32_small_dgrad_try((const float*)dz, lddz, (const float*)w, ldw, act_below, (const float*)aux, ldaux,
                                          (const float*)addend, ldadd, (float*)dx, lddx, M, N, K, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) { IB_PATH(IB_PATH_SMALLM); return rc; }
  }
  if (dtype == IB_F32) return launch_dgrad<float>(p, ib_s(stream));
  if (dtype == IB_BF16) return launch_dgrad<bf16_t>(p, ib_s(stream));
  return IB_E_DTYPE;
}

// The same product with the weight handed over TRANSPOSED (wt[K,N], k-contiguous along the reduction): both operands are
// then "NT" and large-M problems take the 256 x 128 kernel of gemm_nt.hip.  IB_E_UNSUPPORTED (nothing launched) when the
// problem does not qualify -- callers fall back to ib_linear_dgrad with the untransposed weight.
extern "C" int ib_linear_dgrad_wt(const void* dz, int64_t lddz, const void* wt, int64_t ldwt, int act_below, const void* aux,
                                  int64_t ldaux, const void* addend, int64_t ldadd, void* dx, int64_t lddx, int64_t M,
                                  int64_t N, int64_t K, int dtype, ib_stream_t stream) {
  if (!dz || !wt || !dx || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldwt < N || lddx < K) return IB_E_ARG;
  if (addend && ldadd < K) return IB_E_ARG;
  if (act_below != IB_ACT_NONE && (!aux || ldaux < K)) return IB_E_ARG;
  if (dtype != IB_BF16) return IB_E_UNSUPPORTED;
  // C[M,K] = sum_n dz[m][n] * wt[k][n]
  return ib_gemm_nt_try(dz, lddz, wt, ldwt, dx, lddx, nullptr, IB_ACT_NONE, aux, ldaux, act_below, addend, ldadd, M, K, N,
                        ib_s(stream));
}

// ---- skinny dgrad: few rows (a batch of time embeddings), long reduction --------------------------------------------
// dx[M,K] = (dz[M,N] . w[N,K]) * act'(aux), M <= 256.  The tiled kernels give such a problem 8 workgroups that each walk
// the whole reduction alone (measured 40 us for [256, 1024] x [1024, 512], on the step's critical path).  Here a
// workgroup owns 16 OUTPUT COLUMNS and all rows: K/16 workgroups, the w slice [N,16] transposed into LDS once, the dz
// rows streamed straight into MFMA fragments through a 4-deep register ring (each wave its own rows: nothing to share),
// and -- because a workgroup sees every row of its columns -- the bias gradient (column sums of dx) comes out of the
// same launch in a fixed order.
namespace skinny {
constexpr int NMAX = 1024, WS = NMAX + 8, PD = 4;       // w image: 16 columns x (N + 8 pad) bf16, k-contiguous

template <int ACT>
__global__ __launch_bounds__(256) void dgrad_skinny_kernel(const bf16_t* __restrict__ dz, int64_t lddz,
                                                           const bf16_t* __restrict__ w, int64_t ldw,
                                                           const bf16_t* __restrict__ aux, int64_t ldaux,
                                                           bf16_t* __restrict__ dx, int64_t lddx, float* __restrict__ dbias,
                                                           int accumulate, int M, int N) {
  __shared__ __attribute__((aligned(16))) bf16_t wimg[16 * WS];
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 16;
  // w[:, c0 .. c0+16) -> wimg[col][k]: one 16-byte piece = 8 columns of one reduction row
  // (four pieces per thread requested before the first is scattered: a rolled loop was N/128 dependent round trips)
  for (int p0 = tid; p0 < N * 2; p0 += 256 * 4) {
    bf16x8_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pce = min(p0 + 256 * u, N * 2 - 1), k = pce >> 1, half = pce & 1;
      __builtin_memcpy(&v[u], __builtin_assume_aligned(w + (int64_t)k * ldw + c0 + 8 * half, 16), 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pce = p0 + 256 * u, k = pce >> 1, half = pce & 1;
      if (pce < N * 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) wimg[(8 * half + e) * WS + k] = v[u][e];
      }
    }
  }
  __syncthreads();
  const int mt_total = (M + 15) / 16;
  const int r = lane & 15, kq = lane >> 4;
  const bf16_t* arow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = min((wave + 4 * j) * 16 + r, M - 1);   // tiles past the last row read valid memory and store nothing
    arow[j] = dz + (int64_t)row * lddz + 8 * kq;
  }
  f32x4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t ar[PD][4];
  const int nkb = N / 32;                                  // a multiple of PD by the launch conditions
#pragma unroll
  for (int sl = 0; sl < PD; ++sl)
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_memcpy(&ar[sl][j], __builtin_assume_aligned(arow[j] + sl * 32, 16), 16);
  const bf16_t* brow = wimg + r * WS + 8 * kq;
  for (int kb0 = 0; kb0 < nkb; kb0 += PD) {
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) {
      const int kb = kb0 + sl;
      bf16x8_t b;
      __builtin_memcpy(&b, __builtin_assume_aligned(brow + kb * 32, 16), 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ar[sl][j], acc[j], 0, 0, 0);
      const int kn = min(kb + PD, nkb - 1);                // the last PD refills re-read the final block (never used)
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_memcpy(&ar[sl][j], __builtin_assume_aligned(arow[j] + kn * 32, 16), 16);
    }
  }
  // swapped operands: this lane holds columns c0 + 4*kq .. +3 of row (tile, r)
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave + 4 * j) * 16 + r;
    if (wave + 4 * j >= mt_total || row >= M) continue;
    float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
    if constexpr (ACT != IB_ACT_NONE) {
      bf16x4_t a4;
      __builtin_memcpy(&a4, __builtin_assume_aligned(aux + (int64_t)row * ldaux + c0 + 4 * kq, 8), 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_bwd_t<bf16_t, ACT>((float)a4[e]);
    }
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (bf16_t)v[e];
      cs[e] += (float)o[e];                                // the bias gradient is the column sum of the STORED tensor
    }
    __builtin_memcpy(__builtin_assume_aligned(dx + (int64_t)row * lddx + c0 + 4 * kq, 8), &o, 8);
  }
  if (dbias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) cs[e] += __shfl_xor(cs[e], o, 64);     // the 16 rows of a tile: fixed butterfly
      if (r == 0) red[wave][4 * kq + e] = cs[e];
    }
    __syncthreads();
    if (tid < 16) {
      const float sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      dbias[c0 + tid] = accumulate ? dbias[c0 + tid] + sum : sum;
    }
  }
}
}  // namespace skinny

extern "C" int ib_linear_dgrad_skinny(const void* dz, int64_t lddz, const void* w, int64_t ldw, int act_below,
                                      const void* aux, int64_t ldaux, void* dx, int64_t lddx, float* dbias, int accumulate,
                                      int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream) {
  if (!dz || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldw < K || lddx < K) return IB_E_ARG;
  if (act_below != IB_ACT_NONE && (!aux || ldaux < K)) return IB_E_ARG;
  if (act_below < IB_ACT_NONE || act_below > IB_ACT_ELU) return IB_E_ARG;
  if (dtype != IB_BF16 || M > 256 || N > skinny::NMAX || N % (32 * skinny::PD) != 0 || K % 16 != 0) return IB_E_UNSUPPORTED;
  if (!aligned(dz, 16) || lddz % 8 != 0 || !aligned(w, 16) || ldw % 8 != 0 || !aligned(dx, 8) || lddx % 4 != 0 ||
      (aux && (!aligned(aux, 8) || ldaux % 4 != 0)))
    return IB_E_UNSUPPORTED;
  const dim3 grid((unsigned)(K / 16)), block(256);
  hipStream_t s = ib_s(stream);
  IB_PATH(IB_PATH_SKINNY);
#define IB_SKINNY(ACT)                                                                                                  \
  hipLaunchKernelGGL((skinny::dgrad_skinny_kernel<ACT>), grid, block, 0, s, (const bf16_t*)dz, lddz, (const bf16_t*)w, ldw, \
                     (const bf16_t*)aux, ldaux, (bf16_t*)dx, lddx, dbias, accumulate, (int)M, (int)N)
  switch (act_below) {
    case IB_ACT_RELU: IB_SKINNY(IB_ACT_RELU); break;
    case IB_ACT_TANH: IB_SKINNY(IB_ACT_TANH); break;
    case IB_ACT_SIGMOID: IB_SKINNY(IB_ACT_SIGMOID); break;
    case IB_ACT_SILU: IB_SKINNY(IB_ACT_SILU); break;
    case IB_ACT_ELU: IB_SKINNY(IB_ACT_ELU); break;
    default: IB_SKINNY(IB_ACT_NONE); break;
  }
#undef IB_SKINNY
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" size_t ib_linear_wgrad_workspace(int64_t M, int64_t N, int64_t K) {
  int chunk;
  const int split = wgrad_split(M, N, K, 64, &chunk);
  return split > 1 ? (size_t)split * (size_t)N * (size_t)K * sizeof(float) : 0;
}

namespace {
// the split-M weight-gradient GEMM.  slabs_only: always write fp32 partial slabs [split][N][K] into the workspace
// (even for split == 1) and leave the reduction to ib_slab_reduce_multi (one launch for several gradients).
// fills the GemmParams of dW = dz^T x (split over M); returns the error code, *split_out / *tiles_out on success
int wgrad_params(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw, int64_t lddw, int accumulate,
                 void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int dtype, bool slabs_only,
                 GemmParams& p, int* split_out, int* tiles_out, int* chunk_out, int group = 1) {
  if (!dz || !x || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldx < K) return IB_E_ARG;
  if (!slabs_only && (!dw || lddw < K)) return IB_E_ARG;
  if (dtype != IB_F32 && dtype != IB_BF16) return IB_E_DTYPE;
  int chunk;
  const int split = wgrad_split(M, N, K, 64, &chunk, group);
  const bool to_ws = slabs_only || split > 1;
  const size_t need = to_ws ? (size_t)split * (size_t)N * (size_t)K * sizeof(float) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need)) return IB_E_WORKSPACE;
  // C[N,K] = sum_m dz[m][n] * x[m][k]: A(i=n, kk=m) = dz[m*lddz + n], B(j=k, kk=m) = x[m*ldx + k]
  p = GemmParams{};
  p.A = dz; p.lda = lddz; p.B = x; p.ldb = ldx; p.M = (int)N; p.N = (int)K; p.K = (int)M;
  p.seg = 1; p.act = IB_ACT_NONE; p.accumulate = accumulate;
  p.ablate = IB_ABLATE; p.prof = IB_AB_PROF(g_gemm_prof);
  p.tiles_n = (p.N + BN - 1) / BN; p.tiles_m = (p.M + BM - 1) / BM;
  p.k_chunk = chunk;
  // all tiles of one M-chunk on one XCD only while that chunk's dz + x slices fit comfortably in the XCD's 4 MiB
  // L2 (measured: [12800,512,300] 42 -> 36 us grouped, but [12800,1536,512] 48 -> 63 us: 5.2 MB per chunk thrashes)
  p.xcd_group = ((int64_t)chunk * (N + K) * (dtype == IB_BF16 ? 2 : 4) <= (2 << 20)) ? 1 : 0;
  if (to_ws) {
    p.C = workspace; p.ldc = K; p.slab_stride = (int64_t)N * K; p.accumulate = 0;
  } else {
    p.C = dw; p.ldc = lddw; p.slab_stride = 0;
  }
  p.vecC = aligned(p.C, 16) && (p.ldc % 4 == 0) && (p.slab_stride % 4 == 0);
  if (dtype == IB_F32) {
    p.vecA = vec_load_ok<float>(p.A, p.lda); p.vecB = vec_load_ok<float>(p.B, p.ldb);
  } else {
    p.vecA = vec_load_ok<bf16_t>(p.A, p.lda); p.vecB = vec_load_ok<bf16_t>(p.B, p.ldb);
    p.gldsA = glds_ok<bf16_t>(p.A, p.lda); p.gldsB = glds_ok<bf16_t>(p.B, p.ldb);
  }
  *split_out = split; *tiles_out = p.tiles_m * p.tiles_n; *chunk_out = chunk;
  return IB_OK;
}

// the split-M weight-gradient GEMM.  slabs_only: always write fp32 partial slabs [split][N][K] into the workspace
// (even for split == 1) and leave the reduction to ib_slab_reduce_multi (one launch for several gradients).
int wgrad_gemm(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw, int64_t lddw, int accumulate,
               void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int dtype, hipStream_t s,
               bool slabs_only, int* split_out) {
  GemmParams p;
  int split, tiles, chunk;
  const int rc = wgrad_params(dz, lddz, x, ldx, dw, lddw, accumulate, workspace, workspace_bytes, M, N, K, dtype,
                              slabs_only, p, &split, &tiles, &chunk);
  if (rc != IB_OK) return rc;
  if (dtype == IB_F32) {
    IB_PATH(IB_PATH_GENERIC);
    hipLaunchKernelGGL((gemm_kernel<float, false, false, EPI_WGRAD>), dim3(tiles * split), dim3(NTHREADS), 0, s, p);
  } else {
    IB_PATH(ring_ok(p, IB_BF16, p.K, chunk) ? IB_PATH_RING : IB_PATH_GENERIC);
    if (ring_ok(p, IB_BF16, p.K, chunk))
      hipLaunchKernelGGL((gemm_ring_kernel<false, false, EPI_WGRAD>), dim3(tiles * split), dim3(NTHREADS), 0, s, p);
    else
      hipLaunchKernelGGL((gemm_kernel<bf16_t, false, false, EPI_WGRAD>), dim3(tiles * split), dim3(NTHREADS), 0, s, p);
  }
  IB_CHECK_LAUNCH();
  if (split_out) *split_out = split;
  if (!slabs_only && split > 1) {
    const int64_t n = (int64_t)N * K;
    const int rvec = (K % 4 == 0) && (lddw % 4 == 0) && aligned(workspace, 16) && aligned(dw, 16);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ib_grid_1d(rvec ? n / 4 : n, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(workspace), split, (int64_t)N * K, dw, lddw, (int)N, (int)K,
                       accumulate, rvec);
    IB_CHECK_LAUNCH();
  }
  return IB_OK;
}

}  // namespace

// ---- weight gradient over a SHORT reduction (a batch of a few hundred rows) -----------------------------------------------
// dw[N,K] (+)= dz[M,N]^T x[M,K] with M <= 1024.  The split-M kernels above exist to find parallelism in a long reduction;
// here the OUTPUT is the large side ([512, 1470] from 256 rows), so 64 x 64 output tiles alone give hundreds of
// workgroups: no split, no slabs, no reduction launch.  Both operands are indexed [reduction][column]: a workgroup stages
// its two [rows][64] slices in LDS (256 rows at a time) and reads MFMA fragments with the transposing ds_read_b64_tr_b16.
// [256 rows; 512 x 1470]: 20.8 us (GEMM + slab reduction) -> one launch.
namespace wsmall {
constexpr int CH = 256, PITCH = 72;       // rows per LDS chunk; image row pitch in elements (64 + 8: 144 B)

__device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* img, int rbase, int lane) {       // img: first row of a 32-row K step
  using namespace ring;
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int kr = 8 * (lane >> 4) + q;
  const unsigned a = lds_addr(img) + (kr * PITCH + rbase + 4 * pp) * 2;
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * PITCH * 2));
  u32x4_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256) void wgrad_smallm_kernel(const bf16_t* __restrict__ dz, int64_t lddz,
                                                           const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ dw,
                                                           int64_t lddw, float* __restrict__ dbias, int accumulate, int M,
                                                           int N, int K) {
  __shared__ __attribute__((aligned(16))) bf16_t aimg[CH * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t bimg[CH * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  f32x4_t acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int m0 = 0; m0 < M; m0 += CH) {
    if (m0) __syncthreads();                               // the previous chunk's fragments have been read
    const int rows = min(CH, M - m0), rows32 = (rows + 31) / 32 * 32;
    for (int pce = tid; pce < rows32 * 8; pce += 256) {    // [row][8 pieces of 8 columns], rows past the end zero-filled
      const int m = pce >> 3, c = (pce & 7) * 8;
      bf16x8_t va, vb;
      if (m < rows) {
        va = smallm::ldfrag(dz + (int64_t)(m0 + m) * lddz, n0 + c, N);
        vb = smallm::ldfrag(x + (int64_t)(m0 + m) * ldx, k0 + c, K);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { va[e] = (bf16_t)0.f; vb[e] = (bf16_t)0.f; }
      }
      __builtin_memcpy(__builtin_assume_aligned(aimg + m * PITCH + c, 16), &va, 16);
      __builtin_memcpy(__builtin_assume_aligned(bimg + m * PITCH + c, 16), &vb, 16);
    }
    __syncthreads();
    // the bias gradient of the same layer = column sums of dz: the first workgroup of every row of tiles has the slice
    // in LDS anyway (one thread per column, rows added in order)
    if (dbias && blockIdx.x == 0 && tid < 64) {
      float sacc = 0.f;
      for (int m = 0; m < rows; ++m) sacc += (float)aimg[m * PITCH + tid];
      bsum += sacc;
    }
    for (int ks = 0; ks < rows32; ks += 32) {
      bf16x8_t fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) fa[t] = frag_tr(aimg + ks * PITCH, wi * 32 + 16 * t, lane);
#pragma unroll
      for (int u = 0; u < 2; ++u) fb[u] = frag_tr(bimg + ks * PITCH, wj * 32 + 16 * u, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[t], acc[t][u], 0, 0, 0);
    }
  }
  if (dbias && blockIdx.x == 0 && tid < 64 && n0 + tid < N) dbias[n0 + tid] = accumulate ? dbias[n0 + tid] + bsum : bsum;
  // swapped operands: this lane holds output columns (k) 16*u + 4*(lane >> 4) .. +3 of output row (n) 16*t + (lane & 15)
  const bool vst = (lddw % 4) == 0 && (reinterpret_cast<uintptr_t>(dw) % 16) == 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + wi * 32 + 16 * t + (lane & 15);
    if (n >= N) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + wj * 32 + 16 * u + 4 * (lane >> 4);
      if (k >= K) continue;
      float* o = dw + (int64_t)n * lddw + k;
      float v[4] = {acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]};
      if (k + 4 <= K && vst) {
        float4 w4 = make_float4(v[0], v[1], v[2], v[3]);
        if (accumulate) { const float4 p4 = *reinterpret_cast<const float4*>(o); w4.x += p4.x; w4.y += p4.y; w4.z += p4.z; w4.w += p4.w; }
        *reinterpret_cast<float4*>(o) = w4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < K) o[e] = accumulate ? o[e] + v[e] : v[e];
      }
    }
  }
}

inline bool ok(const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int dtype) {
  static const int off = ib_ab_int("IB_NO_SMALLM", 0);
  // M <= 1024: with 2560 rows (Groundlink, F = 10) the ten serial LDS chunks per workgroup lose to the split-M kernels
  // (step 0.496 -> 0.789 ms when the limit was raised to 4096)
  return !off && dtype == IB_BF16 && M <= 1024 && lddz % 2 == 0 && ldx % 2 == 0 && aligned(dz, 4) && aligned(x, 4);
}
}  // namespace wsmall

extern "C" int ib_linear_wgrad(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw,
                               int64_t lddw, int accumulate, void* workspace, size_t workspace_bytes,
                               int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream) {
  if (dz && x && dw && M > 0 && N > 0 && K > 0 && lddz >= N && ldx >= K && lddw >= K && wsmall::ok(dz, lddz, x, ldx, M, dtype)) {
    IB_PATH(IB_PATH_WGRAD_SMALL);
    hipLaunchKernelGGL(wsmall::wgrad_smallm_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0,
                       ib_s(stream), (const bf16_t*)dz, lddz, (const bf16_t*)x, ldx, dw, lddw, nullptr, accumulate, (int)M, (int)N, (int)K);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  return wgrad_gemm(dz, lddz, x, ldx, dw, lddw, accumulate, workspace, workspace_bytes, M, N, K, dtype, ib_s(stream),
                    false, nullptr);
}

extern "C" int ib_linear_wgrad_bias(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw, int64_t lddw,
                                    float* dbias, int accumulate, int64_t M, int64_t N, int64_t K, int dtype,
                                    ib_stream_t stream) {
  if (!dz || !x || !dw || !dbias || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldx < K || lddw < K) return IB_E_ARG;
  if (dtype == IB_F32) {                                    // the reference's batch sizes in fp32 (gemm_f32_small.hip)
    const int rc = ib_f32_small_wgrad_bias_try((const float*)dz, lddz, (const float*)x, ldx, dw, lddw, dbias, accumulate, M, N, K,
                                               ib_s(stream));
    if (rc == IB_OK) IB_PATH(IB_PATH_WGRAD_SMALL);
    return rc;
  }
  if (!wsmall::ok(dz, lddz, x, ldx, M, dtype)) return IB_E_UNSUPPORTED;
  IB_PATH(IB_PATH_WGRAD_SMALL);
  hipLaunchKernelGGL(wsmall::wgrad_smallm_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0,
                     ib_s(stream), (const bf16_t*)dz, lddz, (const bf16_t*)x, ldx, dw, lddw, dbias, accumulate, (int)M, (int)N,
                     (int)K);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" size_t ib_linear_wgrad_slabs_workspace(int64_t M, int64_t N, int64_t K) {
  int chunk;
  int split = wgrad_split(M, N, K, 64, &chunk);
  const int tn = ib_gemm_tn_splits(M, N, K, 1);              // the 256 x 128 kernel (gemm_tn.hip) splits finest when alone
  if (tn > split) split = tn;
  return (size_t)split * (size_t)N * (size_t)K * sizeof(float);
}

extern "C" int ib_linear_wgrad_slabs(const void* dz, int64_t lddz, const void* x, int64_t ldx, void* workspace,
                                     size_t workspace_bytes, int* nslab_out, int64_t M, int64_t N, int64_t K,
                                     int dtype, ib_stream_t stream) {
  if (!nslab_out) return IB_E_ARG;
  if (dz && x && workspace && M > 0 && N > 0 && K > 0 && lddz >= N && ldx >= K && K % 4 == 0 &&
      workspace_bytes >= (size_t)N * K * sizeof(float) && aligned(workspace, 16) && wsmall::ok(dz, lddz, x, ldx, M, dtype)) {
    // short reduction: the one-pass kernel writes the whole gradient as a single "slab"
    IB_PATH(IB_PATH_WGRAD_SMALL);
    hipLaunchKernelGGL(wsmall::wgrad_smallm_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0,
                       ib_s(stream), (const bf16_t*)dz, lddz, (const bf16_t*)x, ldx, reinterpret_cast<float*>(workspace), K,
                       nullptr, 0, (int)M, (int)N, (int)K);
    IB_CHECK_LAUNCH();
    *nslab_out = 1;
    return IB_OK;
  }
  if (dtype == IB_BF16 && dz && x && workspace && lddz >= N && ldx >= K) {      // long reduction: the 256 x 128 kernel
    const void* dzs[1] = {dz}; const void* xs[1] = {x}; void* wss[1] = {workspace};
    const int64_t la[1] = {lddz}, lx[1] = {ldx}, Ms[1] = {M}, Ns[1] = {N}, Ks[1] = {K};
    const size_t wb[1] = {workspace_bytes};
    int32_t ns = 0;
    const int rc = ib_gemm_tn_multi(1, dzs, la, xs, lx, wss, wb, nullptr, &ns, Ms, Ns, Ks, ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) { *nslab_out = ns; return rc; }
  }
  return wgrad_gemm(dz, lddz, x, ldx, nullptr, 0, 0, workspace, workspace_bytes, M, N, K, dtype, ib_s(stream), true,
                    nslab_out);
}

extern "C" int ib_linear_wgrad_slabs_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                           const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                           int32_t* nslab_out, const int64_t* M, const int64_t* N, const int64_t* K,
                                           int dtype, ib_stream_t stream) {
  return ib_linear_wgrad_slabs_multi_bias(n, dz, lddz, x, ldx, workspace, workspace_bytes, nullptr, nslab_out, M, N, K, dtype,
                                          stream);
}

extern "C" int ib_linear_wgrad_slabs_multi_bias(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                                const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                                float* const* dbias_part, int32_t* nslab_out, const int64_t* M,
                                                const int64_t* N, const int64_t* K, int dtype, ib_stream_t stream) {
  if (n <= 0 || n > WG_MAX || !dz || !lddz || !x || !ldx || !workspace || !workspace_bytes || !nslab_out || !M || !N || !K)
    return IB_E_ARG;
  if (dtype != IB_BF16) return IB_E_UNSUPPORTED;
  {   // long reductions: the 256 x 128 LDS-DMA kernel (half the operand re-reads of the 128 x 128 tiles)
    const int rc = ib_gemm_tn_multi(n, dz, lddz, x, ldx, workspace, workspace_bytes, dbias_part, nslab_out, M, N, K,
                                    ib_s(stream));
    if (rc != IB_E_UNSUPPORTED) return rc;
  }
  WgradMulti m{};
  m.n = n;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    int split, tiles, chunk;
    const int rc = wgrad_params(dz[j], lddz[j], x[j], ldx[j], nullptr, 0, 0, workspace[j], workspace_bytes[j], M[j], N[j],
                                K[j], dtype, true, m.p[j], &split, &tiles, &chunk, n);
    if (rc != IB_OK) return rc;
    if (!ring_ok(m.p[j], IB_BF16, m.p[j].K, chunk)) return IB_E_UNSUPPORTED;   // caller falls back to single launches
    m.p[j].dbias_part = dbias_part ? dbias_part[j] : nullptr;      // [IB_WGRAD_MAX_SPLIT][N] fp32, rows 0 .. split-1 written
    nslab_out[j] = split;
    m.blk0[j] = blocks;
    blocks += tiles * split;
  }
  m.blk0[n] = blocks;
  IB_PATH(IB_PATH_RING_MULTI);
  hipLaunchKernelGGL(gemm_ring_wgrad_multi_kernel, dim3(blocks), dim3(NTHREADS), 0, ib_s(stream), m);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// The grouped weight-gradient launch with the time-MLP's hidden-layer backward (ib_time_mlp_bwd) riding along as extra
// workgroups of the SAME launch (csrc/gemm_tn.hip).  IB_E_UNSUPPORTED = nothing launched: the caller issues
// ib_linear_wgrad_slabs_multi and ib_time_mlp_bwd separately.
extern "C" int ib_linear_wgrad_slabs_multi_tb(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                              const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                              int32_t* nslab_out, const int64_t* M, const int64_t* N, const int64_t* K,
                                              int dtype, const void* de, int64_t ld_de, const void* w2, int64_t ldw2,
                                              const void* zu, int64_t ldzu, const void* s_rows, int64_t lds, float* dw1_slabs,
                                              float* db1_slabs, int64_t B, int64_t temb, int64_t hidden, int64_t out,
                                              ib_stream_t stream) {
  if (n <= 0 || n > WG_MAX || !dz || !lddz || !x || !ldx || !workspace || !workspace_bytes || !nslab_out || !M || !N || !K)
    return IB_E_ARG;
  if (dtype != IB_BF16 || !ib_time_mlp_bwd_supported(temb, hidden, out)) return IB_E_UNSUPPORTED;
  if (!de || !w2 || !zu || !s_rows || !dw1_slabs || !db1_slabs || B <= 0) return IB_E_ARG;
  if (ld_de < out || ldw2 < hidden || ldzu < hidden || lds < temb || ld_de % 8 || ldw2 % 8 || lds % 8) return IB_E_ARG;
  if (!aligned(de, 16) || !aligned(w2, 16) || !aligned(zu, 2) || !aligned(s_rows, 16) || !aligned(dw1_slabs, 16) ||
      !aligned(db1_slabs, 16))
    return IB_E_ARG;
  TimeBwdParams tb{(const bf16_t*)de, ld_de, (const bf16_t*)w2, ldw2, (const bf16_t*)zu, ldzu, (const bf16_t*)s_rows, lds,
                   dw1_slabs, db1_slabs, (int)B, (int)out, (int)hidden};
  return ib_gemm_tn_multi(n, dz, lddz, x, ldx, workspace, workspace_bytes, nullptr, nslab_out, M, N, K, ib_s(stream), &tb,
                          (int)temb);
}

// ---- y = LayerNorm(res + x W^T + b) for small token counts (the DDIM sampler: M = B*T = 3200): the GEMM is split over
// K into fp32 slabs so that a [3200, 512, 2048] problem fills the chip (25 x 4 tiles x 4 slices instead of 100
// workgroups walking 64 K steps each), and the slab reduction IS the LayerNorm kernel (bias + residual + statistics +
// affine), so the separate LayerNorm launch and the bf16 round trip of the GEMM output disappear.
namespace {
// N = 4 * LPR * NCH columns; LPR lanes per row, a lane holds NCH float4 (columns (c * LPR + l) * 4 ..).  Wide rows use a
// whole wave per row (3200 rows = 12 waves per CU; with 16 lanes per row the launch was 3 waves per CU and each lane
// walked 32 dependent-ish loads: 17 us for 26 MB)
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void slab_ln_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_stride,
                                                      const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                      int64_t ldres, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t ldy,
                                                      bf16_t* __restrict__ a_out, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int M, float eps) {
  constexpr int N = 4 * LPR * NCH, RPB = 256 / LPR;
  const int l = threadIdx.x % LPR;
  const int row = blockIdx.x * RPB + threadIdx.x / LPR;
  if (row >= M) return;
  // Slabs in batches of up to SB requested together (predicated on the count), added strictly in slab order.  As a rolled
  // loop -- one slab per trip, the trip count a runtime value -- this was `nslab` memory round trips in sequence: at the
  // sampler's M = 200 (16 slabs of the FFN output projection) 10 of the Linear + LayerNorm pair's 17 us.
  constexpr int SB = NCH <= 2 ? 8 : 4;
  float4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* srow = slabs + (int64_t)row * N + l * 4;
  for (int k0 = 0; k0 < nslab; k0 += SB) {
    float4 w[SB][NCH];
#pragma unroll
    for (int e = 0; e < SB; ++e)
      if (k0 + e < nslab) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          w[e][c] = *reinterpret_cast<const float4*>(srow + (int64_t)(k0 + e) * slab_stride + c * LPR * 4);
      }
#pragma unroll
    for (int e = 0; e < SB; ++e)
      if (k0 + e < nslab) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (k0 + e == 0) v[c] = w[e][c];          // the first slab is taken as it is (0 + x would turn -0 into +0)
          else { v[c].x += w[e][c].x; v[c].y += w[e][c].y; v[c].z += w[e][c].z; v[c].w += w[e][c].w; }
        }
      }
  }
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * LPR + l) * 4;
    float4 t = v[c];
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + col); t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    if (a_out) {      // the GEMM output itself (what the per-op plan stores between the two launches), bf16
      bf16x4_t o; o[0] = (bf16_t)t.x; o[1] = (bf16_t)t.y; o[2] = (bf16_t)t.z; o[3] = (bf16_t)t.w;
      *reinterpret_cast<bf16x4_t*>(a_out + (int64_t)row * N + col) = o;
      t.x = (float)o[0]; t.y = (float)o[1]; t.z = (float)o[2]; t.w = (float)o[3];
    }
    if (res) {
      const bf16x4_t r = *reinterpret_cast<const bf16x4_t*>(res + (int64_t)row * ldres + col);
      t.x += (float)r[0]; t.y += (float)r[1]; t.z += (float)r[2]; t.w += (float)r[3];
    }
    v[c] = t;
    s1 += (t.x + t.y) + (t.z + t.w);
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
  const float mu = s1 * (1.f / N);
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float a = v[c].x - mu, b = v[c].y - mu, cc = v[c].z - mu, d = v[c].w - mu;
    s2 += (a * a + b * b) + (cc * cc + d * d);
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  const float rs = 1.f / sqrtf(s2 * (1.f / N) + eps);
  if (l == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * LPR + l) * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(beta + col);
    bf16x4_t o;
    o[0] = (bf16_t)((v[c].x - mu) * rs * g.x + b.x); o[1] = (bf16_t)((v[c].y - mu) * rs * g.y + b.y);
    o[2] = (bf16_t)((v[c].z - mu) * rs * g.z + b.z); o[3] = (bf16_t)((v[c].w - mu) * rs * g.w + b.w);
    *reinterpret_cast<bf16x4_t*>(y + (int64_t)row * ldy + col) = o;
  }
}

int linear_ln_split(int64_t M, int64_t N, int64_t K, int* chunk_out) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int64_t want = (384 + tiles - 1) / tiles;              // ~1.5 workgroups per CU
  if (want > K / 128) want = K / 128;                      // at least four 32-deep K steps per slice
  if (want < 1) want = 1;
  int64_t chunk = ((K + want - 1) / want + 31) / 32 * 32;
  *chunk_out = (int)chunk;
  return (int)((K + chunk - 1) / chunk);
}
}  // namespace

// the reduction / LayerNorm launch on its own, for N = 512: slabs of another producer (linln_panel.hip's feed-forward launch)
int ib_slab_ln512_launch(const float* slabs, int nslab, int64_t slab_stride, const float* bias, const void* res, int64_t ldres,
                         const float* gamma, const float* beta, void* y, int64_t ldy, int64_t M, float eps, hipStream_t s) {
  hipLaunchKernelGGL((slab_ln_kernel<64, 2>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, slabs, nslab, slab_stride, bias,
                     reinterpret_cast<const bf16_t*>(res), ldres, gamma, beta, reinterpret_cast<bf16_t*>(y), ldy,
                     (bf16_t*)nullptr, (float*)nullptr, (float*)nullptr, (int)M, eps);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" size_t ib_linear_ln_fwd_workspace(int64_t M, int64_t N, int64_t K) {
  int chunk;
  int split = linear_ln_split(M, N, K, &chunk);
  const int nt = ib_gemm_nt_splitk_splits(M, N, K);            // the 256 x 128 kernel's split-K form may use more slabs
  if (nt > split) split = nt;
  return (size_t)split * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int ib_linear_ln_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const void* res,
                                int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, void* a_out,
                                float* mean, float* rstd, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
                                int64_t K, float eps, int dtype, ib_stream_t stream) {
  if (!x || !w || !gamma || !beta || !y || !workspace || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || ldy < N)
    return IB_E_ARG;
  if (res && ldres < N) return IB_E_ARG;
  if (dtype != IB_BF16 || N % 64 != 0 || N > 1024 || K % 32 != 0) return IB_E_UNSUPPORTED;
  int chunk;
  int split = linear_ln_split(M, N, K, &chunk);
  if (!aligned(workspace, 16) || !aligned(gamma, 16) || !aligned(beta, 16) || (bias && !aligned(bias, 16)) ||
      !aligned(y, 8) || ldy % 4 != 0 || (res && (!aligned(res, 8) || ldres % 4 != 0)) || (a_out && !aligned(a_out, 8)))
    return IB_E_ARG;
  hipStream_t s = ib_s(stream);
  // a few thousand rows (the sampler at B = 16: [3200, 512, 2048]): the 256 x 128 LDS-DMA kernel in split-K form -- 52 tiles x
  // 4 splits = 208 work items of 8 K steps instead of 400 ring-kernel workgroups of 16 short steps
  bool gemm_done = false;
  const int nts = ib_gemm_nt_splitk_splits(M, N, K);
  if (nts >= 2 && dtype == IB_BF16 && workspace_bytes >= (size_t)nts * M * N * sizeof(float)) {
    const int rc = ib_gemm_nt_splitk(x, ldx, w, ldw, reinterpret_cast<float*>(workspace), nts, M, N, K, s);
    if (rc == IB_OK) { gemm_done = true; split = nts; }
    else if (rc != IB_E_UNSUPPORTED) return rc;
  }
  if (workspace_bytes < (size_t)split * M * N * sizeof(float)) return IB_E_WORKSPACE;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.B = w; p.ldb = ldw; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.seg = 1; p.act = IB_ACT_NONE; p.ablate = IB_ABLATE; p.prof = IB_AB_PROF(g_gemm_prof);
  p.vecA = vec_load_ok<bf16_t>(p.A, p.lda); p.vecB = vec_load_ok<bf16_t>(p.B, p.ldb);
  p.gldsA = glds_ok<bf16_t>(p.A, p.lda); p.gldsB = glds_ok<bf16_t>(p.B, p.ldb);
  p.tiles_n = (p.N + BN - 1) / BN; p.tiles_m = (p.M + BM - 1) / BM;
  p.k_chunk = chunk; p.xcd_group = 1;
  p.C = workspace; p.ldc = N; p.slab_stride = (int64_t)M * N; p.accumulate = 0; p.vecC = 1;
  if (!gemm_done) {
    if (!ring_ok(p, IB_BF16, p.K, chunk)) return IB_E_UNSUPPORTED;
    IB_PATH(IB_PATH_LINLN);
    hipLaunchKernelGGL((gemm_ring_kernel<true, true, EPI_WGRAD>), dim3(p.tiles_m * p.tiles_n * split), dim3(NTHREADS), 0, s, p);
    IB_CHECK_LAUNCH();
  }
#define IB_SLAB_LN(LPR, NCH)                                                                                            \
  hipLaunchKernelGGL((slab_ln_kernel<LPR, NCH>), dim3((unsigned)((M + 256 / LPR - 1) / (256 / LPR))), dim3(256), 0, s,     \
                     reinterpret_cast<const float*>(workspace), split, (int64_t)M * N, bias,                               \
                     reinterpret_cast<const bf16_t*>(res), ldres, gamma, beta, reinterpret_cast<bf16_t*>(y), ldy,           \
                     reinterpret_cast<bf16_t*>(a_out), mean, rstd, (int)M, eps)
  switch (N) {
    case 64: IB_SLAB_LN(16, 1); break; case 128: IB_SLAB_LN(16, 2); break; case 256: IB_SLAB_LN(64, 1); break;
    case 512: IB_SLAB_LN(64, 2); break; case 1024: IB_SLAB_LN(64, 4); break;
    default: return IB_E_UNSUPPORTED;
  }
#undef IB_SLAB_LN
  IB_CHECK_LAUNCH();
  return IB_OK;
}
