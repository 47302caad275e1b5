// Fused feed-forward sublayer of the reference's post-norm encoder layer for gfx950 (round 4):
//
//     x2 = LayerNorm2( x1 + W2 . ReLU(W1 . x1 + b1) + b2 )            src/models/TransformerBaseline.py:15-19,33-36
//
// and its backward, each as ONE launch over panels of token rows -- the design of the MLP denoiser's chain kernel
// (chain.hip) applied to the transformer denoiser's token-local half.  The per-op plan ran it as
// GEMM [M,512]x[512,2048] (+ReLU) -> GEMM [M,2048]x[2048,512] -> residual + LayerNorm: at M = 12800 both GEMMs are bound by
// the L2 -> LDS ingest of 256 x 128 tiles with 22 % tile quantisation on 256 CUs (47 + 40 + 13 us per layer forward), and
// the [M, 2048] hidden activation makes an HBM round trip between them.
//
// Here a workgroup (512 threads = 8 waves) owns a panel of <= 64 token rows (50 at the headline shape: 256 workgroups = 256
// CUs, one round, no tile quantisation) and walks the hidden width in CHUNKS of 512 columns:
//     forward   per chunk c:  z = x1 . W1[c]^T (GEMM, K = 512)  -> ReLU -> bf16 into LDS image H (+ 64 mask bits per lane)
//                             y += H . W2[:, c]^T (GEMM, K = 512, accumulators stay in registers across the chunks)
//               then          s2 = x1 + y + b2 -> LDS -> row-wise LayerNorm (wave owns rows, DPP row sums) -> x2, s2, stats
//     backward  first         row-wise LayerNorm backward of dy -> dz2 (LDS image Z + HBM), dgamma / dbeta partial sums
//               per chunk c:  dh = Z . W2[:, c] (GEMM) -> masked by the forward's ReLU bits -> bf16 into LDS image D
//                             dx += D . W1[c] (GEMM, accumulators across chunks);  D rows -> HBM (dz1, the weight gradients' operand)
//               then          dx1 = dx + dz2 (residual) -> rows -> HBM
// Weights stream L2 -> VGPR from fragment-major packed images exactly as in chain.hip (a wave owns 64 output columns of
// every GEMM: its weight slice is private, every wave-instruction reads one contiguous 1 KiB block, k-block-major so the
// 32 KiB the eight waves request per k-step are contiguous).  The hidden activation f1 and dz1 still go to HBM ONCE
// (the weight-gradient GEMMs read them) as coalesced row stores interleaved behind the weight prefetch; they are never
// read back by this path.  Bound: the matrix pipes (64-row panels: 8 GEMM phases of [64 x 512 x 512] per direction) and the
// L2 -> CU weight stream (4 MB per workgroup and direction).
#include "ib_common.h"
#include <initializer_list>

namespace {

constexpr int FF_ROWS = 64, FF_WAVES = 8, FF_THREADS = 512, FF_D = 512, FF_CHUNK = 512, FF_MAXCHUNK = 8;
constexpr int FF_NT = 4, FF_KB = 16;                      // n-tiles per wave, k-blocks of 32 per GEMM phase
constexpr int FF_RS = FF_D * 2 + 16;                      // LDS row stride (bytes): +16 -> conflict-free b128 reads
constexpr int FF_BUF = FF_ROWS * FF_RS;
constexpr int64_t FF_WELEMS = (int64_t)FF_CHUNK * FF_D;   // elements of one packed [512 x 512] weight image
#ifdef IB_AB
long long* g_ffn_prof = nullptr;     // TIMING-ONLY (tools/ffn_prof.py): [workgroups][64] wall-clock stamps, measurement builds
#endif
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
#ifndef FF_PIPE
#define FF_PIPE 0
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
#if FF_PIPE
  // the activation fragments of k-block kb + 1 are requested BEFORE the MFMAs of k-block kb (double-buffered: the LDS round
  // trip runs under 16 MFMAs instead of in front of them)
  bf16x8_t fa[2][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) fa[0][mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS);
#endif
#pragma unroll
  for (int kb = 0; kb < FF_KB; ++kb) {
    if (kb + PD < FF_KB) {
#pragma unroll
      for (int u = 0; u < FF_NT; ++u) wr[(kb + PD) % RING][u] = wl[(u + (kb + PD) * SK) * 64];
    }
    side(FfIntC<0>{}, kb);
    __builtin_amdgcn_sched_barrier(0);
#if FF_PIPE
    if (kb + 1 < FF_KB) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        fa[(kb + 1) & 1][mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS + 64 * (kb + 1));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int u = 0; u < FF_NT; ++u)
        acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[kb % RING][u], fa[kb & 1][mt], acc[mt][u], 0, 0, 0);
#else
    bf16x8_t fa[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS + 64 * kb);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int u = 0; u < FF_NT; ++u)
        acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[kb % RING][u], fa[mt], acc[mt][u], 0, 0, 0);
#endif
  }
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

// LayerNorm of the bf16 rows of `img` (the saved LayerNorm input): y rows -> HBM (and, if `back`, back into the image: the
// next GEMM's input), the input rows -> HBM (`sg`), mean / rstd -> HBM.  Rows beyond the panel: computed on whatever
// finite values the image holds, never stored.
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
        *reinterpret_cast<uint4*>(yg + (int64_t)r * FF_D + c8) = hq;
        *reinterpret_cast<uint4*>(sg + (int64_t)r * FF_D + c8) = q[gi];
        if (lane == 0) { mean_g[r] = mean; rstd_g[r] = rstd; }
      }
    }
  }
}

// LayerNorm backward of the panel's rows: dy rows from HBM (`dyg`) or, if `dyimg`, from an LDS image; the saved LayerNorm
// input rows and statistics from HBM; dz rows -> image `zimg` (rows beyond the panel: exact zeros) and -> HBM (`dzg`);
// this lane's dgamma | dbeta sums over the wave's rows come back in dgam / dbet (ff_colsum_put / ff_colsum_out add them up).
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
    if (dyimg) qd[j] = *reinterpret_cast<const uint4*>(dyimg + r * FF_RS + c8 * 2);
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

struct FfnFwdParams {
  const bf16_t* x1;                 // [M, 512] sublayer input (plain form) -- or, with the attention epilogue, the LAYER input
                                    // x (residual of the first LayerNorm)
  const bf16_t* w1p; const bf16_t* w2p;      // packed: chunk c at + c * FF_WELEMS
  const float* b1; const float* b2; const float* gamma; const float* beta;
  bf16_t* f1;                       // [M, FF]   ReLU output (the weight gradients' operand)
  bf16_t* s2;                       // [M, 512]  x1 + f2 + b2 (LayerNorm input, read by the backward)
  bf16_t* y;                        // [M, 512]  LayerNorm output
  float* mean; float* rstd;         // [M]
  uint2* mask;                      // [workgroups][nchunk][512] ReLU bits of every lane's 64 accumulator values
  // attention epilogue (OUT): x1 = LayerNorm1(x + attn . Wo^T + bo) computed here instead of read
  const bf16_t* attn; const bf16_t* wop; const float* bo; const float* gamma1; const float* beta1;
  bf16_t* s1; bf16_t* x1out; float* mean1; float* rstd1;
  int M, P, FF, nchunk;
  float ln_eps;
  long long* prof;
};

template <bool OUT>
__global__ __launch_bounds__(FF_THREADS) void ffn_chain_fwd_kernel(FfnFwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FF_BUF];
  unsigned char* imgX = smem;
  unsigned char* imgH = smem + FF_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#define FF_TIDV ((wave_s << 6) | ff_lane())
  const int g = lane >> 4, l16 = lane & 15;
  const int r0 = blockIdx.x * p.P;
  const int nrows = min(p.P, p.M - r0);
  const int colb = wave * 16 * FF_NT + 4 * g;             // this lane's first column inside a 512-wide GEMM output
  const int c8 = lane * 8;                                // row-wise passes: 8 consecutive columns of the wave's row
  FF_STAMP(0);
  // ---- the panel's input rows -> image X (16-byte pieces, rows beyond the panel = copies of the last row: finite)
  ff_panel_in(p.x1 + (int64_t)r0 * FF_D, imgX, nrows, tid);
  if constexpr (OUT) {
    // attention output rows -> image H; o = attn . Wo^T (GEMM); s1 = x + o + bo written over x in image X (each position is
    // read and written by the same lane); row-wise LayerNorm1 turns image X into x1, the feed-forward input
    ff_panel_in(p.attn + (int64_t)r0 * FF_D, imgH, nrows, tid);
    __syncthreads();
    f32x4_t acco[4][FF_NT];
    ff_zero(acco);
    float4 bo4[FF_NT];
    auto sideo = [&](auto, int kb) {
      if (kb == FF_KB - 1) {
#pragma unroll
        for (int u = 0; u < FF_NT; ++u) bo4[u] = *reinterpret_cast<const float4*>(p.bo + colb + 16 * u);
      }
    };
    ff_gemm<FF_RING_B>(p.wop, wave_s * FF_NT, imgH, ff_lane(), acco, sideo);
    float gm1[8], bt1[8];
    ff_load8f(p.gamma1 + c8, gm1);
    ff_load8f(p.beta1 + c8, bt1);
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4_t* slot = reinterpret_cast<bf16x4_t*>(imgX + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2);
        const bf16x4_t xv = *slot;
        *slot = ff_pack4((acco[mt][u][0] + bo4[u].x) + (float)xv[0], (acco[mt][u][1] + bo4[u].y) + (float)xv[1],
                         (acco[mt][u][2] + bo4[u].z) + (float)xv[2], (acco[mt][u][3] + bo4[u].w) + (float)xv[3]);
      }
    }
    __syncthreads();                     // image X = s1 complete
    ff_ln_rows_fwd(imgX, true, nrows, wave_s, lane, gm1, bt1, p.ln_eps, p.x1out + (int64_t)r0 * FF_D,
                   p.s1 + (int64_t)r0 * FF_D, p.mean1 + r0, p.rstd1 + r0);
  }
  __syncthreads();
  FF_STAMP(1);

  f32x4_t accy[4][FF_NT];
  ff_zero(accy);
  for (int c = 0; c < p.nchunk; ++c) {
    f32x4_t acc1[4][FF_NT];
    ff_zero(acc1);
    float4 b4[FF_NT];
    auto side1 = [&](auto, int kb) {
      if (kb == FF_KB - 1) {
#pragma unroll
        for (int u = 0; u < FF_NT; ++u) b4[u] = *reinterpret_cast<const float4*>(p.b1 + c * FF_CHUNK + colb + 16 * u);
      }
    };
    ff_gemm<FF_RING_A>(p.w1p + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgX, ff_lane(), acc1, side1);
    FF_STAMP(2 + 4 * c);
    __syncthreads();                     // every wave is past the previous chunk's second GEMM: image H may be rewritten
    FF_STAMP(3 + 4 * c);
    uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
      const float bb[4] = {b4[u].x, b4[u].y, b4[u].z, b4[u].w};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = acc1[mt][u][r] + bb[r];
          const bool on = z > 0.f;
          const int bit = (mt * FF_NT + u) * 4 + r;       // compile-time constant
          if (bit < 32) mlo |= on ? (1u << bit) : 0u; else mhi |= on ? (1u << (bit - 32)) : 0u;
          v[r] = on ? z : 0.f;
        }
        *reinterpret_cast<bf16x4_t*>(imgH + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) = ff_pack4(v[0], v[1], v[2], v[3]);
      }
    }
    if (p.mask) p.mask[((int64_t)blockIdx.x * p.nchunk + c) * FF_THREADS + tid] = make_uint2(mlo, mhi);
    __syncthreads();                     // image H = ReLU chunk complete
    FF_STAMP(4 + 4 * c);
    bf16_t* f1g = p.f1 + (int64_t)r0 * p.FF + c * FF_CHUNK;
    auto side2 = [&](auto, int kb) {     // the chunk's rows -> HBM: 8 pieces per thread, one per two k-blocks
      if ((kb & 1) == 0) ff_out_piece(imgH, f1g, p.FF, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
    };
    ff_gemm<FF_RING_B>(p.w2p + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgH, ff_lane(), accy, side2);
    FF_STAMP(5 + 4 * c);
  }

  // ---- s2 = x1 + y + b2 (column owners, fp32) -> bf16 into image H -> row-wise LayerNorm
  float gm[8], bt[8];
  ff_load8f(p.gamma + c8, gm);
  ff_load8f(p.beta + c8, bt);
  float4 b2v[FF_NT];
#pragma unroll
  for (int u = 0; u < FF_NT; ++u) b2v[u] = *reinterpret_cast<const float4*>(p.b2 + colb + 16 * u);
  __syncthreads();                       // the last chunk's second GEMM is done reading image H
#pragma unroll
  for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const bf16x4_t xv = *reinterpret_cast<const bf16x4_t*>(imgX + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2);
      *reinterpret_cast<bf16x4_t*>(imgH + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) =
          ff_pack4((accy[mt][u][0] + b2v[u].x) + (float)xv[0], (accy[mt][u][1] + b2v[u].y) + (float)xv[1],
                   (accy[mt][u][2] + b2v[u].z) + (float)xv[2], (accy[mt][u][3] + b2v[u].w) + (float)xv[3]);
    }
  }
  __syncthreads();
  ff_ln_rows_fwd(imgH, false, nrows, wave_s, lane, gm, bt, p.ln_eps, p.y + (int64_t)r0 * FF_D, p.s2 + (int64_t)r0 * FF_D,
                 p.mean + r0, p.rstd + r0);
  FF_STAMP(2 + 4 * p.nchunk);
}

struct FfnBwdParams {
  const bf16_t* dy;                 // [M, 512] gradient w.r.t. the LayerNorm output
  const bf16_t* s2;                 // [M, 512] LayerNorm input saved by the forward
  const float* mean; const float* rstd; const float* gamma;
  const bf16_t* w2tp; const bf16_t* w1tp;    // packed transposed weights, chunk c at + c * FF_WELEMS
  const uint2* mask;
  bf16_t* ds2;                      // [M, 512]  d(x1 + f2) = dz2: the feedforward.2 weight gradient's operand + residual addend
  bf16_t* dz1;                      // [M, FF]   gradient w.r.t. the hidden pre-activation
  bf16_t* dx1;                      // [M, 512]  gradient w.r.t. the sublayer input (plain form only)
  float* partial;                   // [2 or 4][workgroups][512]: dgamma2, dbeta2 (, dgamma1, dbeta1) of every panel
  // attention epilogue (OUT): LayerNorm1 backward of dx1 and the out-projection's dgrad
  const bf16_t* s1; const float* mean1; const float* rstd1; const float* gamma1; const bf16_t* wotp;
  bf16_t* ds1;                      // [M, 512]  d(x + o): the out-projection's weight-gradient operand + the layer input's addend
  bf16_t* dattn;                    // [M, 512]  ds1 . Wo
  int M, P, FF, nchunk;
  long long* prof;
};

template <bool OUT>
__global__ __launch_bounds__(FF_THREADS) void ffn_chain_bwd_kernel(FfnBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FF_BUF];
  unsigned char* imgZ = smem;
  unsigned char* imgD = smem + FF_BUF;
  static_assert(2 * FF_WAVES * FF_D * 4 <= FF_BUF, "dgamma / dbeta exchange must fit an image");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int g = lane >> 4, l16 = lane & 15;
  const int r0 = blockIdx.x * p.P;
  const int nrows = min(p.P, p.M - r0);
  const int colb = wave * 16 * FF_NT + 4 * g;
  // ---- LayerNorm2 backward, row-wise: dz2 rows -> image Z (+ HBM); its dgamma | dbeta through image D's storage
  {
    float dgam[8], dbet[8];
    ff_ln_rows_bwd(p.dy + (int64_t)r0 * FF_D, nullptr, p.s2 + (int64_t)r0 * FF_D, p.mean + r0, p.rstd + r0, p.gamma, nrows,
                   wave_s, lane, imgZ, p.ds2 + (int64_t)r0 * FF_D, dgam, dbet);
    ff_colsum_put(reinterpret_cast<float*>(imgD), wave, lane, dgam, dbet);
  }
  __syncthreads();                       // image Z = dz2 complete; the exchange rows are written
  ff_colsum_out(reinterpret_cast<const float*>(imgD), p.partial, 0, tid);
  __syncthreads();                       // image D is free for the first chunk

  f32x4_t accx[4][FF_NT];
  ff_zero(accx);
  for (int c = 0; c < p.nchunk; ++c) {
    f32x4_t acca[4][FF_NT];
    ff_zero(acca);
    uint2 mk = make_uint2(0u, 0u);
    auto sideA = [&](auto, int kb) {
      if (kb == FF_KB - 1) mk = p.mask[((int64_t)blockIdx.x * p.nchunk + c) * FF_THREADS + FF_TIDV];
    };
    ff_gemm<FF_RING_BA>(p.w2tp + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgZ, ff_lane(), acca, sideA);
    __syncthreads();                     // every wave is past the previous chunk's second GEMM: image D may be rewritten
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bit = (mt * FF_NT + u) * 4 + r;
          const bool on = bit < 32 ? ((mk.x >> bit) & 1u) : ((mk.y >> (bit - 32)) & 1u);
          v[r] = on ? acca[mt][u][r] : 0.f;
        }
        *reinterpret_cast<bf16x4_t*>(imgD + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) = ff_pack4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();                     // image D = dz1 chunk complete
    bf16_t* dzg = p.dz1 + (int64_t)r0 * p.FF + c * FF_CHUNK;
    auto sideB = [&](auto, int kb) {
      if ((kb & 1) == 0) ff_out_piece(imgD, dzg, p.FF, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
    };
    ff_gemm<FF_RING_BB>(p.w1tp + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgD, ff_lane(), accx, sideB);
  }
  // ---- dx1 = dx + dz2 (the residual path) -> image D
  __syncthreads();
#pragma unroll
  for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const bf16x4_t zv = *reinterpret_cast<const bf16x4_t*>(imgZ + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2);
      *reinterpret_cast<bf16x4_t*>(imgD + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) =
          ff_pack4(accx[mt][u][0] + (float)zv[0], accx[mt][u][1] + (float)zv[1], accx[mt][u][2] + (float)zv[2],
                   accx[mt][u][3] + (float)zv[3]);
    }
  }
  __syncthreads();                       // image D = dx1; image Z is free
  if constexpr (!OUT) {
    ff_panel_out(imgD, p.dx1 + (int64_t)r0 * FF_D, nrows, tid);
  } else {
    // LayerNorm1 backward of dx1 (rows of image D) -> ds1 rows -> image Z (+ HBM); then dattn = ds1 . Wo
    float dgam[8], dbet[8];
    ff_ln_rows_bwd(nullptr, imgD, p.s1 + (int64_t)r0 * FF_D, p.mean1 + r0, p.rstd1 + r0, p.gamma1, nrows, wave_s, lane,
                   imgZ, p.ds1 + (int64_t)r0 * FF_D, dgam, dbet);
    __syncthreads();                     // image Z = ds1 complete; image D (dx1 rows) no longer read: it takes the exchange
    ff_colsum_put(reinterpret_cast<float*>(imgD), wave, lane, dgam, dbet);
    __syncthreads();
    ff_colsum_out(reinterpret_cast<const float*>(imgD), p.partial, 2, tid);
    __syncthreads();                     // ... and is free again for the out-projection's result
    f32x4_t acco[4][FF_NT];
    ff_zero(acco);
    auto sideo = [&](auto, int) {};
    ff_gemm<FF_RING_BB>(p.wotp, wave_s * FF_NT, imgZ, ff_lane(), acco, sideo);
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        *reinterpret_cast<bf16x4_t*>(imgD + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) =
            ff_pack4(acco[mt][u][0], acco[mt][u][1], acco[mt][u][2], acco[mt][u][3]);
    }
    __syncthreads();
    ff_panel_out(imgD, p.dattn + (int64_t)r0 * FF_D, nrows, tid);
  }
}
#undef FF_TIDV

// ---- weight packing: [512 x 512] sub-matrices of the row-major bf16 weights -> fragment-major 1-KiB blocks
// (block (nt, kb) at (kb * 32 + nt) * 1 KiB; W_eff[n][k] = src[n * ld + k], or src[k * ld + n] when transposed)
struct FfnPackDesc { const bf16_t* src; int64_t ld; int transpose; int pad; bf16_t* dst; };
constexpr int FF_MAXDESC = 96;                            // layers x ({fwd1, fwd2, bwdA, bwdB} x chunks + {Wo, Wo^T}): 4 layers of ffn 2048 = 72
struct FfnPackParams { FfnPackDesc d[FF_MAXDESC]; int count; };
constexpr int FF_BLOCKS_PER_DESC = 32 * FF_KB;            // 512 one-KiB blocks

__global__ __launch_bounds__(256) void ffn_pack_kernel(FfnPackParams p) {
  __shared__ bf16_t tile[4][32][17];                      // per wave: a [32 k][16 n] source tile, transposed through LDS
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int total = p.count * FF_BLOCKS_PER_DESC;
  for (int blk = blockIdx.x * 4 + w; blk < total; blk += gridDim.x * 4) {
    const FfnPackDesc& d = p.d[blk / FF_BLOCKS_PER_DESC];
    const int local = blk % FF_BLOCKS_PER_DESC;
    const int nt = local % 32, kb = local / 32;
    bf16x8_t v;
    if (!d.transpose) {
      const int n = 16 * nt + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
      v = *reinterpret_cast<const bf16x8_t*>(d.src + (int64_t)n * d.ld + k0);
    } else {
      // source rows are k: lane (kk = lane / 2, half = lane % 2) reads 8 consecutive n of row k0 + kk (16-byte loads), the
      // [32 k][16 n] tile is turned through LDS, lane (n, kq) then takes k = 8 kq .. 8 kq + 7 of column n
      const int kk = lane >> 1, hf = lane & 1;
      const bf16x8_t s = *reinterpret_cast<const bf16x8_t*>(d.src + (int64_t)(32 * kb + kk) * d.ld + 16 * nt + 8 * hf);
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[w][kk][8 * hf + j] = s[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[w][8 * kq + j][n];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    reinterpret_cast<bf16x8_t*>(d.dst)[(int64_t)local * 64 + lane] = v;
  }
}

int ffn_geometry(int64_t M, int64_t d, int64_t ffn, int* P, int* nchunk) {
  if (d != FF_D || ffn <= 0 || ffn % FF_CHUNK != 0 || ffn / FF_CHUNK > FF_MAXCHUNK || M <= 0) return 0;
  int64_t rows = (M + 255) / 256;                         // one workgroup per CU when the token count allows it
  if (rows > FF_ROWS) rows = FF_ROWS;
  if (rows < 16) rows = M < 16 ? M : 16;
  *P = (int)rows;
  *nchunk = (int)(ffn / FF_CHUNK);
  return (int)((M + rows - 1) / rows);
}

}  // namespace

#ifdef IB_AB
extern "C" int ib_debug_set_ffn_prof(void* buf) { g_ffn_prof = reinterpret_cast<long long*>(buf); return IB_OK; }
#else
extern "C" int ib_debug_set_ffn_prof(void*) { return IB_E_UNSUPPORTED; }       // measurement builds only
#endif

extern "C" int ib_ffn_chain_supported(int64_t d, int64_t ffn) {
  return (d == FF_D && ffn > 0 && ffn % FF_CHUNK == 0 && ffn / FF_CHUNK <= FF_MAXCHUNK) ? 1 : 0;
}
// elements of ONE layer's packed image: {W1 chunks | W2 chunks | W2^T chunks | W1^T chunks | Wo | Wo^T}, each 512 x 512
extern "C" size_t ib_ffn_chain_packed_elems(int64_t d, int64_t ffn) {
  return ib_ffn_chain_supported(d, ffn) ? (size_t)((4 * (ffn / FF_CHUNK) + 2) * FF_WELEMS) : 0;
}
extern "C" int ib_ffn_chain_workgroups(int64_t M, int64_t d, int64_t ffn, int* rows_per_wg) {
  int P = 0, nc = 0;
  const int n = ffn_geometry(M, d, ffn, &P, &nc);
  if (rows_per_wg) *rows_per_wg = P;
  return n;
}
extern "C" size_t ib_ffn_chain_mask_bytes(int64_t M, int64_t d, int64_t ffn) {
  int P = 0, nc = 0;
  const int n = ffn_geometry(M, d, ffn, &P, &nc);
  return (size_t)n * nc * FF_THREADS * sizeof(uint2);
}

extern "C" int ib_ffn_chain_pack(const void* const* w1, const int64_t* ld1, const void* const* w2, const int64_t* ld2,
                                 const void* const* wo, const int64_t* ldo, void* const* packed, int layers, int64_t d,
                                 int64_t ffn, ib_stream_t stream) {
  if (!w1 || !ld1 || !w2 || !ld2 || !packed || layers < 1 || !ib_ffn_chain_supported(d, ffn)) return IB_E_ARG;
  const int nc = (int)(ffn / FF_CHUNK);
  if (layers * (4 * nc + 2) > FF_MAXDESC) return IB_E_UNSUPPORTED;
  FfnPackParams pp{};
  int c = 0;
  for (int l = 0; l < layers; ++l) {
    if (!w1[l] || !w2[l] || !packed[l] || ld1[l] < d || ld2[l] < ffn || ld1[l] % 8 || ld2[l] % 8) return IB_E_ARG;
    if ((reinterpret_cast<uintptr_t>(w1[l]) | reinterpret_cast<uintptr_t>(w2[l]) | reinterpret_cast<uintptr_t>(packed[l])) % 16)
      return IB_E_ARG;
    const bf16_t* W1 = reinterpret_cast<const bf16_t*>(w1[l]);     // [ffn, d]
    const bf16_t* W2 = reinterpret_cast<const bf16_t*>(w2[l]);     // [d, ffn]
    bf16_t* dst = reinterpret_cast<bf16_t*>(packed[l]);
    for (int q = 0; q < 4; ++q)
      for (int k = 0; k < nc; ++k) {
        FfnPackDesc& e = pp.d[c++];
        e.dst = dst + (int64_t)(q * nc + k) * FF_WELEMS;
        // q = 0: W_eff[n][kk] = W1[512 k + n][kk]          (forward, hidden chunk k)
        // q = 1: W_eff[n][kk] = W2[n][512 k + kk]          (forward, output from hidden chunk k)
        // q = 2: W_eff[n][kk] = W2[kk][512 k + n]          (backward: dh chunk k from dz2)
        // q = 3: W_eff[n][kk] = W1[512 k + kk][n]          (backward: dx from dz1 chunk k)
        if (q == 0) { e.src = W1 + (int64_t)FF_CHUNK * k * ld1[l]; e.ld = ld1[l]; e.transpose = 0; }
        if (q == 1) { e.src = W2 + (int64_t)FF_CHUNK * k; e.ld = ld2[l]; e.transpose = 0; }
        if (q == 2) { e.src = W2 + (int64_t)FF_CHUNK * k; e.ld = ld2[l]; e.transpose = 1; }
        if (q == 3) { e.src = W1 + (int64_t)FF_CHUNK * k * ld1[l]; e.ld = ld1[l]; e.transpose = 1; }
      }
    if (wo && wo[l]) {                   // the attention out-projection [d, d]: forward image and its transpose
      if (!ldo || ldo[l] < d || ldo[l] % 8 || reinterpret_cast<uintptr_t>(wo[l]) % 16) return IB_E_ARG;
      for (int q = 0; q < 2; ++q) {
        FfnPackDesc& e = pp.d[c++];
        e.src = reinterpret_cast<const bf16_t*>(wo[l]); e.ld = ldo[l]; e.transpose = q;
        e.dst = dst + (int64_t)(4 * nc + q) * FF_WELEMS;
      }
    }
  }
  pp.count = c;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(ib_grid_1d((int64_t)c * FF_BLOCKS_PER_DESC, 4, 2048)), dim3(256), 0, ib_s(stream), pp);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

namespace {
bool ff_al16(std::initializer_list<const void*> ptrs) {
  for (const void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) % 16) return false;
  return true;
}
}  // namespace

extern "C" int ib_ffn_chain_fwd(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                                const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                                const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1,
                                void* x1_out, float* mean1, float* rstd1, int64_t M, int64_t d, int64_t ffn, float ln_eps,
                                ib_stream_t stream) {
  FfnFwdParams p{};
  int P = 0, nc = 0;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc);
  if (!nwg) return IB_E_UNSUPPORTED;
  if (!x1 || !packed || !b1 || !b2 || !gamma || !beta || !f1 || !s2 || !y || !mean || !rstd) return IB_E_ARG;
  if (!ff_al16({x1, packed, b1, b2, gamma, beta, f1, s2, y, mask})) return IB_E_ARG;
  const bool out = attn != nullptr;
  if (out && (!bo || !gamma1 || !beta1 || !s1 || !x1_out || !mean1 || !rstd1 || !ff_al16({attn, bo, gamma1, beta1, s1, x1_out})))
    return IB_E_ARG;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.x1 = (const bf16_t*)x1; p.w1p = pk; p.w2p = pk + (int64_t)nc * FF_WELEMS;
  p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta;
  p.f1 = (bf16_t*)f1; p.s2 = (bf16_t*)s2; p.y = (bf16_t*)y; p.mean = mean; p.rstd = rstd; p.mask = (uint2*)mask;
  p.attn = (const bf16_t*)attn; p.wop = pk + (int64_t)4 * nc * FF_WELEMS; p.bo = bo; p.gamma1 = gamma1; p.beta1 = beta1;
  p.s1 = (bf16_t*)s1; p.x1out = (bf16_t*)x1_out; p.mean1 = mean1; p.rstd1 = rstd1;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc; p.ln_eps = ln_eps;
  p.prof = IB_AB_PROF(g_ffn_prof);
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (out) hipLaunchKernelGGL(ffn_chain_fwd_kernel<true>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL(ffn_chain_fwd_kernel<false>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_ffn_chain_bwd(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                                const void* packed, const void* mask, void* ds2, void* dz1, void* dx1, float* partial,
                                const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1,
                                void* dattn, int64_t M, int64_t d, int64_t ffn, ib_stream_t stream) {
  FfnBwdParams p{};
  int P = 0, nc = 0;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc);
  if (!nwg) return IB_E_UNSUPPORTED;
  if (!dy || !s2 || !mean || !rstd || !gamma || !packed || !mask || !ds2 || !dz1 || !partial) return IB_E_ARG;
  if (!ff_al16({dy, s2, gamma, packed, mask, ds2, dz1, dx1, partial})) return IB_E_ARG;
  const bool out = s1 != nullptr;
  if (out ? (!mean1 || !rstd1 || !gamma1 || !ds1 || !dattn || !ff_al16({s1, gamma1, ds1, dattn})) : !dx1) return IB_E_ARG;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.dy = (const bf16_t*)dy; p.s2 = (const bf16_t*)s2; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.w2tp = pk + (int64_t)2 * nc * FF_WELEMS; p.w1tp = pk + (int64_t)3 * nc * FF_WELEMS;
  p.mask = (const uint2*)mask; p.ds2 = (bf16_t*)ds2; p.dz1 = (bf16_t*)dz1; p.dx1 = (bf16_t*)dx1;
  p.partial = partial;
  p.s1 = (const bf16_t*)s1; p.mean1 = mean1; p.rstd1 = rstd1; p.gamma1 = gamma1;
  p.wotp = pk + (int64_t)(4 * nc + 1) * FF_WELEMS; p.ds1 = (bf16_t*)ds1; p.dattn = (bf16_t*)dattn;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc;
  p.prof = nullptr;
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (out) hipLaunchKernelGGL(ffn_chain_bwd_kernel<true>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL(ffn_chain_bwd_kernel<false>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
