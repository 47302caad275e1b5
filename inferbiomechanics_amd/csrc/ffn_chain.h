// Shared device helpers of the fused token-local transformer kernels (ffn_chain.hip: forward + weight packing;
// ffn_chain_bwd.hip: backward).  Two translation units: with both directions' <OUT, QKV> instantiations in one module hipcc
// (ROCm 7.2) dies in its inliner's call-graph update.
#pragma once
#include "ib_common.h"
#include <initializer_list>
#include <utility>

#ifdef IB_AB
extern long long* g_ffn_prof;        // ffn_chain.hip
#endif

namespace {

constexpr int FF_ROWS = 64, FF_WAVES = 8, FF_THREADS = 512, FF_D = 512, FF_CHUNK = 512, FF_MAXCHUNK = 8;
constexpr int FF_NT = 4, FF_KB = 16;                      // n-tiles per wave, k-blocks of 32 per GEMM phase
constexpr int FF_RS = FF_D * 2 + 16;                      // LDS row stride (bytes): +16 -> conflict-free b128 reads
constexpr int FF_BUF = FF_ROWS * FF_RS;
constexpr int64_t FF_WELEMS = (int64_t)FF_CHUNK * FF_D;   // elements of one packed [512 x 512] weight image
#define FF_STAMP(k) do { if (p.prof && tid == 0) p.prof[blockIdx.x * 64 + (k)] = wall_clock64(); } while (0)

template <int CTRL>
__device__ __forceinline__ float ff_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float ff_dpp_bcast_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
// sum over the 64 lanes of a wave (= one 512-column row, 8 columns per lane); DPP only, the total comes back through a
// scalar register (chain.hip::group_sum<64>)
__device__ __forceinline__ float ff_row_sum(float a) {
  a += ff_dpp_mov<0xB1>(a);     // quad_perm [1,0,3,2]
  a += ff_dpp_mov<0x4E>(a);     // quad_perm [2,3,0,1]
  a += ff_dpp_mov<0x141>(a);    // row_half_mirror
  a += ff_dpp_mov<0x140>(a);    // row_mirror
  a = ff_dpp_bcast_add<0x142, 0xA>(a);     // row_bcast15
  a = ff_dpp_bcast_add<0x143, 0xC>(a);     // row_bcast31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
}
__device__ __forceinline__ int ff_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ bf16x4_t ff_pack4(float a, float b, float c, float d) {
  bf16x4_t o;
  o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
  return o;
}
__device__ __forceinline__ void ff_unpack8(const uint4& q, float (&x)[8]) {
  const bf16x8_t v = __builtin_bit_cast(bf16x8_t, q);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
}
__device__ __forceinline__ uint4 ff_pack8(const float (&x)[8]) {
  bf16x8_t o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (bf16_t)x[k];
  return __builtin_bit_cast(uint4, o);
}
__device__ __forceinline__ void ff_load8f(const float* p, float (&x)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

template <int V> struct FfIntC { static constexpr int value = V; };
// acc[mt][u] += W_eff[16 (nt0 + u) .. +15][:] . A[16 mt .. +15][:]^T over FF_KB k-blocks of 32 (chain.hip::chain_gemm:
// swapped MFMA roles, 3-deep register ring, issue point of every k-block's prefetch pinned by a scheduling barrier,
// `side(kb)` = one piece of a neighbouring phase's row traffic per k-block, BEHIND the weight loads)
// RING = depth of the register prefetch ring (RING - 1 k-blocks of 4 KiB per wave in flight): the GEMM phases are bound by
// the L2 -> VGPR weight stream, i.e. by the bytes a CU keeps in flight; the phase with one live accumulator set affords a
// deeper ring than the phase that holds both
#ifndef FF_RING_A
#define FF_RING_A 3
#endif
#ifndef FF_RING_B
#define FF_RING_B 3
#endif
#ifndef FF_RING_BA          // the backward kernel's two phases
#define FF_RING_BA FF_RING_A
#endif
#ifndef FF_RING_BB
#define FF_RING_BB FF_RING_B
#endif
// (the k-loop is a compile-time index sequence, not an unrolled runtime loop: `side` receives its k-block as a TYPE, so a side
// job that parks a loaded piece in `array[kb / 2]` indexes with a constant the optimiser sees before inlining -- with a
// runtime-parameter index the array stayed in scratch and every piece was waited for with vmcnt(0) right behind its load)
template <int RING, int KB_, class Side>
__device__ __forceinline__ void ff_gemm_step(const bf16x8_t* __restrict__ wl, const unsigned char* arow,
                                             bf16x8_t (&wr)[RING][FF_NT], f32x4_t (&acc)[4][FF_NT], Side&& side) {
  constexpr int PD = RING - 1, SK = FF_WAVES * FF_NT;
  if constexpr (KB_ + PD < FF_KB) {
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) wr[(KB_ + PD) % RING][u] = wl[(u + (KB_ + PD) * SK) * 64];
  }
  side(FfIntC<KB_>{}, KB_);
  __builtin_amdgcn_sched_barrier(0);
  bf16x8_t fa[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS + 64 * KB_);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u)
      acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[KB_ % RING][u], fa[mt], acc[mt][u], 0, 0, 0);
}
template <int RING, class Side, int... KBs>
__device__ __forceinline__ void ff_gemm_seq(const bf16x8_t* __restrict__ wl, const unsigned char* arow,
                                            bf16x8_t (&wr)[RING][FF_NT], f32x4_t (&acc)[4][FF_NT], Side&& side,
                                            std::integer_sequence<int, KBs...>) {
  (ff_gemm_step<RING, KBs>(wl, arow, wr, acc, side), ...);
}
template <int RING, class Side>
__device__ __forceinline__ void ff_gemm(const bf16_t* __restrict__ wp, int nt0, const unsigned char* abuf, int lane,
                                        f32x4_t (&acc)[4][FF_NT], Side&& side) {
  constexpr int PD = RING - 1, SK = FF_WAVES * FF_NT;
  const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(wp) + (int64_t)nt0 * 64 + lane;
  const unsigned char* arow = abuf + (lane & 15) * FF_RS + 16 * (lane >> 4);
  bf16x8_t wr[RING][FF_NT];
#pragma unroll
  for (int s = 0; s < PD; ++s)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) wr[s][u] = wl[(u + s * SK) * 64];
  ff_gemm_seq<RING>(wl, arow, wr, acc, side, std::make_integer_sequence<int, FF_KB>{});
}
__device__ __forceinline__ void ff_zero(f32x4_t (&acc)[4][FF_NT]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) acc[mt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}
// one 16-byte piece of a panel image -> its place in a row-major HBM matrix (row pitch ldg elements); branch-free: rows
// beyond the panel are clamped onto its last row (a duplicate store of identical bytes) -- a branch around a store inside
// the k-loop makes hipcc drain the weight prefetch ring at the join (chain.hip)
__device__ __forceinline__ void ff_out_piece(const unsigned char* img, bf16_t* g, int64_t ldg, int nrows, int idx) {
  const int row = min(idx >> 6, nrows - 1), pc = idx & 63;
  const uint4 v = *reinterpret_cast<const uint4*>(img + row * FF_RS + pc * 16);
  *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(g + (int64_t)row * ldg) + pc * 16) = v;
}

// ---- row-wise passes: wave w owns rows w, w + 8, ... of the panel, a lane 8 consecutive columns of the row (one row per
// wave-instruction); the row sums are DPP-only (ff_row_sum).  Two rows at a time, stage by stage.

// LayerNorm of the bf16 rows of `img` (the saved LayerNorm input): y rows -> HBM (`yg`, may be null) (and, if `back`, back
// into the image: the next GEMM's input), the input rows -> HBM (`sg`; null: neither they nor mean / rstd are stored),
// mean / rstd -> HBM.  Rows beyond the panel: computed on whatever finite values the image holds, never stored.
__device__ __forceinline__ void ff_ln_rows_fwd(unsigned char* img, bool back, int nrows, int wave_s, int lane,
                                               const float (&gm)[8], const float (&bt)[8], float eps, bf16_t* yg, bf16_t* sg,
                                               float* mean_g, float* rstd_g) {
  const float invH = 1.f / (float)FF_D;
  const int c8 = lane * 8;
  constexpr int G = 2;
#pragma unroll
  for (int j0 = 0; j0 < FF_ROWS / FF_WAVES; j0 += G) {
    if (!back && wave_s + FF_WAVES * j0 >= nrows) break;
    uint4 q[G];
    float v[G][8], s1[G], sq[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) q[gi] = *reinterpret_cast<const uint4*>(img + (wave_s + FF_WAVES * (j0 + gi)) * FF_RS + c8 * 2);
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      ff_unpack8(q[gi], v[gi]);
      float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        a0 += v[gi][k]; a1 += v[gi][k + 1];
        q0 += v[gi][k] * v[gi][k]; q1 += v[gi][k + 1] * v[gi][k + 1];
      }
      s1[gi] = a0 + a1; sq[gi] = q0 + q1;
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) { s1[gi] = ff_row_sum(s1[gi]); sq[gi] = ff_row_sum(sq[gi]); }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int r = wave_s + FF_WAVES * (j0 + gi);
      const float mean = s1[gi] * invH;
      const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq[gi] * invH - mean * mean, 0.f) + eps);
      float hv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) hv[k] = (v[gi][k] - mean) * rstd * gm[k] + bt[k];
      const uint4 hq = ff_pack8(hv);
      if (back) *reinterpret_cast<uint4*>(img + r * FF_RS + c8 * 2) = hq;
      if (r < nrows) {
        if (yg) *reinterpret_cast<uint4*>(yg + (int64_t)r * FF_D + c8) = hq;
        if (sg) {                      // (frozen-weight forward: nothing is saved for a backward)
          *reinterpret_cast<uint4*>(sg + (int64_t)r * FF_D + c8) = q[gi];
          if (lane == 0) { mean_g[r] = mean; rstd_g[r] = rstd; }
        }
      }
    }
  }
}

// LayerNorm backward of the panel's rows: dy rows from HBM (`dyg`) or, if `dyimg`, from an LDS image; the saved LayerNorm
// input rows and statistics from HBM; dz rows -> image `zimg` (rows beyond the panel: exact zeros) and -> HBM (`dzg`);
// this lane's dgamma | dbeta sums over the wave's rows come back in dgam / dbet (ff_colsum_put / ff_colsum_out add them up).
template <bool FROM_IMG>
__device__ __forceinline__ void ff_ln_rows_bwd(const bf16_t* dyg, const unsigned char* dyimg, const bf16_t* sg,
                                               const float* mean_g, const float* rstd_g, const float* gamma, int nrows,
                                               int wave_s, int lane, unsigned char* zimg, bf16_t* dzg, float (&dgam)[8],
                                               float (&dbet)[8]) {
  constexpr int NJ = FF_ROWS / FF_WAVES;
  const float invH = 1.f / (float)FF_D;
  const int c8 = lane * 8;
  uint4 qd[NJ], qs[NJ];
  float2 st[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {         // every row of the wave requested before the first is used
    const int r = wave_s + FF_WAVES * j, rc = min(r, nrows - 1);
    if constexpr (FROM_IMG) qd[j] = *reinterpret_cast<const uint4*>(dyimg + r * FF_RS + c8 * 2);
    else qd[j] = *reinterpret_cast<const uint4*>(dyg + (int64_t)rc * FF_D + c8);
    qs[j] = *reinterpret_cast<const uint4*>(sg + (int64_t)rc * FF_D + c8);
    st[j] = make_float2(mean_g[rc], rstd_g[rc]);
  }
  float gm[8];
  ff_load8f(gamma + c8, gm);
#pragma unroll
  for (int k = 0; k < 8; ++k) { dgam[k] = 0.f; dbet[k] = 0.f; }
  constexpr int G = 2;
#pragma unroll
  for (int j0 = 0; j0 < NJ; j0 += G) {
    float dxh[G][8], xh[G][8], sa[G], sb[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const bool ok = wave_s + FF_WAVES * (j0 + gi) < nrows;
      float d[8], sv[8];
      ff_unpack8(qd[j0 + gi], d);
      ff_unpack8(qs[j0 + gi], sv);
      sa[gi] = 0.f; sb[gi] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dk = ok ? d[k] : 0.f;
        xh[gi][k] = (sv[k] - st[j0 + gi].x) * st[j0 + gi].y;
        dgam[k] += dk * xh[gi][k];
        dbet[k] += dk;
        dxh[gi][k] = dk * gm[k];
        sa[gi] += dxh[gi][k];
        sb[gi] += dxh[gi][k] * xh[gi][k];
      }
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) { sa[gi] = ff_row_sum(sa[gi]); sb[gi] = ff_row_sum(sb[gi]); }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int r = wave_s + FF_WAVES * (j0 + gi);
      const float ma = sa[gi] * invH, mb = sb[gi] * invH;
      float dz[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = st[j0 + gi].y * (dxh[gi][k] - ma - xh[gi][k] * mb);
      const uint4 zq = ff_pack8(dz);                      // rows beyond the panel: dxh = 0 -> exact zeros
      *reinterpret_cast<uint4*>(zimg + r * FF_RS + c8 * 2) = zq;
      if (r < nrows) *reinterpret_cast<uint4*>(dzg + (int64_t)r * FF_D + c8) = zq;
    }
  }
}
// a wave's per-lane dgamma | dbeta sums -> its rows of the exchange area `cr` ([2][8 waves][512] floats of LDS)
__device__ __forceinline__ void ff_colsum_put(float* cr, int wave, int lane, const float (&dgam)[8], const float (&dbet)[8]) {
  const int c8 = lane * 8;
  float* c0 = cr + (0 * FF_WAVES + wave) * FF_D + c8;
  float* c1 = cr + (1 * FF_WAVES + wave) * FF_D + c8;
  *reinterpret_cast<float4*>(c0) = make_float4(dgam[0], dgam[1], dgam[2], dgam[3]);
  *reinterpret_cast<float4*>(c0 + 4) = make_float4(dgam[4], dgam[5], dgam[6], dgam[7]);
  *reinterpret_cast<float4*>(c1) = make_float4(dbet[0], dbet[1], dbet[2], dbet[3]);
  *reinterpret_cast<float4*>(c1 + 4) = make_float4(dbet[4], dbet[5], dbet[6], dbet[7]);
}
// the panel's dgamma | dbeta: the eight waves' rows of `cr` added in wave order -> partial[q0][wg], partial[q0 + 1][wg]
__device__ __forceinline__ void ff_colsum_out(const float* cr, float* partial, int q0, int tid) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int w = 0; w < FF_WAVES; ++w) {
    s0 += cr[(0 * FF_WAVES + w) * FF_D + tid];
    s1 += cr[(1 * FF_WAVES + w) * FF_D + tid];
  }
  partial[((int64_t)q0 * gridDim.x + blockIdx.x) * FF_D + tid] = s0;
  partial[((int64_t)(q0 + 1) * gridDim.x + blockIdx.x) * FF_D + tid] = s1;
}
// 64 rows x 64 pieces of 16 bytes from HBM rows into an image (rows beyond the panel = copies of its last row: finite)
__device__ __forceinline__ void ff_panel_in(const bf16_t* g, unsigned char* img, int nrows, int tid) {
  uint4 xr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + j * FF_THREADS;
    const int row = min(idx >> 6, nrows - 1), pc = idx & 63;
    xr[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(g + (int64_t)row * FF_D) + pc * 16);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + j * FF_THREADS;
    *reinterpret_cast<uint4*>(img + (idx >> 6) * FF_RS + (idx & 63) * 16) = xr[j];
  }
}
__device__ __forceinline__ void ff_panel_out(const unsigned char* img, bf16_t* g, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + j * FF_THREADS;
    const int row = idx >> 6, pc = idx & 63;
    if (row < nrows)
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(g + (int64_t)row * FF_D) + pc * 16) =
          *reinterpret_cast<const uint4*>(img + row * FF_RS + pc * 16);
  }
}

#define FF_TIDV ((wave_s << 6) | ff_lane())

// ---- temporal self-attention INSIDE the panel launches (round 5).  With a panel = one window (P == T <= 64 frames) and
// d = 512 = 8 heads x 64, wave w's 64 output columns of the in-projection GEMMs ARE head w: its Q, K, V (and, in the
// backward, the dattn columns the out-projection's dgrad leaves in its accumulators) never concern another wave.  The whole
// softmax(Q K^T / 8) V of a (window, head) and its backward are wave-private: operands come from the wave's own accumulators
// or from its own 64-column slice of the two LDS images, no workgroup barrier inside.
// Lane map of v_mfma_f32_16x16x32_bf16 (attention_mfma.hip): A[row = lane % 16][k = 8 (lane / 16) + j],
// B[k][col = lane % 16], D[row = 4 (lane / 16) + r][col = lane % 16].
constexpr int FF_HEADS = FF_D / 64;                       // = FF_WAVES: one head per wave
constexpr float FF_ATT_SCALE = 0.125f;                    // 1 / sqrt(64)
typedef __attribute__((address_space(3))) s16x4_t ff_lds_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short ff_s16x8_t;

// accumulator pair -> operand of the next product: elements 0..3 <- tile a, 4..7 <- tile b (the reduction index of the next
// product is permuted identically on both of its operands)
__device__ __forceinline__ bf16x8_t ff_acc_frag(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t f;
  f[0] = (bf16_t)a[0]; f[1] = (bf16_t)a[1]; f[2] = (bf16_t)a[2]; f[3] = (bf16_t)a[3];
  f[4] = (bf16_t)b[0]; f[5] = (bf16_t)b[1]; f[6] = (bf16_t)b[2]; f[7] = (bf16_t)b[3];
  return f;
}
__device__ __forceinline__ bf16x8_t ff_cat4(const bf16x4_t& a, const bf16x4_t& b) {
  bf16x8_t f;
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  return f;
}
__device__ __forceinline__ float ff_g4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float ff_g4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// `sl` = the wave's 64-column slice of an image (image + 128 * wave bytes, rows FF_RS bytes apart).
// 8 consecutive columns of one slice row: operand of a product that reduces over the head dimension
__device__ __forceinline__ bf16x8_t ff_sl_row(const unsigned char* sl, int row, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(sl + row * FF_RS + 64 * ks + 16 * (lane >> 4));
}
// transposed operand A[i = column 16 dt + lane % 16][k = slice rows row_lo + 4 g + (0..3) | row_hi + 4 g + (0..3)]
__device__ __forceinline__ bf16x8_t ff_sl_tr(const unsigned char* sl, int row_lo, int row_hi, int dt, int lane) {
  const int qq = (lane & 15) >> 2, pp = lane & 3, g = lane >> 4;
  const unsigned char* a0 = sl + (row_lo + 4 * g + qq) * FF_RS + dt * 32 + 8 * pp;
  const unsigned char* a1 = sl + (row_hi + 4 * g + qq) * FF_RS + dt * 32 + 8 * pp;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ff_lds_s16x4_t*)(a0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ff_lds_s16x4_t*)(a1));
  ff_s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
// LDS written by some lanes of this wave, read by others: order the wave's own accesses (no workgroup barrier)
__device__ __forceinline__ void ff_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// T_att > 0: panels of exactly one window (the launches with the attention inside); 0: the token-count rule
inline int ffn_geometry(int64_t M, int64_t d, int64_t ffn, int* P, int* nchunk, int64_t T_att = 0) {
  if (d != FF_D || ffn <= 0 || ffn % FF_CHUNK != 0 || ffn / FF_CHUNK > FF_MAXCHUNK || M <= 0) return 0;
  int64_t rows = (M + 255) / 256;                         // one workgroup per CU when the token count allows it
  if (rows > FF_ROWS) rows = FF_ROWS;
  if (rows < 16) rows = M < 16 ? M : 16;
  if (T_att > 0) {
    if (T_att < 16 || T_att > FF_ROWS || M % T_att != 0) return 0;
    rows = T_att;
  }
  *P = (int)rows;
  *nchunk = (int)(ffn / FF_CHUNK);
  return (int)((M + rows - 1) / rows);
}


inline bool ff_al16(std::initializer_list<const void*> ptrs) {
  for (const void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) % 16) return false;
  return true;
}

}  // namespace
