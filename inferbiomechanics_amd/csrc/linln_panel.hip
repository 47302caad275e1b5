// y = LayerNorm(res + x . W^T + bias) for d = 512 x 512 projections at FEW THOUSAND rows and below (the DDIM sampler's
// attention out-projection + residual + LayerNorm1, TransformerBaseline.py:12-13,29-31, at B = 1 ... 16 windows of 200 frames),
// bf16, gfx950 -- ONE launch over panels of rows, the design of the fused training launches (ffn_chain.hip) cut down to its
// first phase.
//
// What it replaces at these row counts: a split-K GEMM into fp32 slabs + a reduction / LayerNorm launch (ib_linear_ln_fwd:
// 4 slabs of [3200, 512] fp32 written and read back = 52 MB of traffic and a kernel boundary for a 1.7-GFLOP product: 19.3 us
// at M = 3200, 11.5 us at M = 200).  Here a 512-thread workgroup owns a panel of P = ceil(M / 256) rows (13 at M = 3200: 247
// workgroups, one round), keeps the GEMM's fp32 result in LDS and normalises it in place: the rows are read once and written
// once.  Bound: the 512 KB of packed weights every workgroup streams through its CU's L2 port (6-7 us at the ~80 GB/s a CU
// takes in, ffn_chain.h), i.e. the launch is as long as ONE GEMM phase of the training kernels plus its row traffic.
// Panels are at most 32 rows (two MFMA row tiles): more rows per workgroup would leave CUs idle long before the matrix pipes
// matter.  The weights come from the layer's packed image (ib_ffn_chain_pack: block (nt, kb) of 1 KiB at (kb * 32 + nt) KiB).
#include "ffn_chain.h"

namespace {

constexpr int LP_XS = FF_D * 4 + 16;                       // bytes per fp32 exchange row

struct LinLnPanelParams {
  const bf16_t* x; int64_t ldx;                            // [M, 512] GEMM input rows
  const bf16_t* wp;                                        // packed [512 x 512] weight image (W_eff[n][k] = W[n][k])
  const float* bias;                                       // [512] or NULL
  const bf16_t* res; int64_t ldres;                        // [M, 512] residual rows or NULL
  const float* gamma; const float* beta;
  bf16_t* y; int64_t ldy;
  int M, P;
  float eps;
};

// ff_gemm (ffn_chain.h) over MT row tiles instead of four: same packed-weight stream, same 3-deep register ring, the issue
// point of every k-block's prefetch pinned by a scheduling barrier
template <int MT, int KB_, class Side>
__device__ __forceinline__ void lp_gemm_step(const bf16x8_t* __restrict__ wl, const unsigned char* arow,
                                             bf16x8_t (&wr)[3][FF_NT], f32x4_t (&acc)[MT][FF_NT], Side&& side) {
  constexpr int RING = 3, PD = RING - 1, SK = FF_WAVES * FF_NT;
  if constexpr (KB_ + PD < FF_KB) {
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) wr[(KB_ + PD) % RING][u] = wl[(u + (KB_ + PD) * SK) * 64];
  }
  side(FfIntC<KB_>{});
  __builtin_amdgcn_sched_barrier(0);
  bf16x8_t fa[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS + 64 * KB_);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u)
      acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[KB_ % RING][u], fa[mt], acc[mt][u], 0, 0, 0);
}
template <int MT, class Side, int... KBs>
__device__ __forceinline__ void lp_gemm_seq(const bf16x8_t* __restrict__ wl, const unsigned char* arow, bf16x8_t (&wr)[3][FF_NT],
                                            f32x4_t (&acc)[MT][FF_NT], Side&& side, std::integer_sequence<int, KBs...>) {
  (lp_gemm_step<MT, KBs>(wl, arow, wr, acc, side), ...);
}

template <int MT>
__global__ __launch_bounds__(FF_THREADS) void linln_panel_kernel(LinLnPanelParams p) {
  constexpr int ROWS = 16 * MT, PIECES = ROWS * 64 / FF_THREADS;          // 16-byte pieces of the input rows per thread
  __shared__ __attribute__((aligned(16))) unsigned char img[ROWS * FF_RS];
  __shared__ __attribute__((aligned(16))) unsigned char exch[ROWS * LP_XS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int g = lane >> 4, l16 = lane & 15;
  const int r0 = blockIdx.x * p.P;
  const int nrows = min(p.P, p.M - r0);
  const int colb = wave * 16 * FF_NT + 4 * g;
  // ---- the weight ring's first two k-blocks are requested ahead of the rows (nothing else is in flight yet)
  const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(p.wp) + (int64_t)(wave_s * FF_NT) * 64 + ff_lane();
  bf16x8_t wr[3][FF_NT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) wr[s][u] = wl[(u + s * (FF_WAVES * FF_NT)) * 64];
  // ---- the panel's input rows -> LDS image (rows beyond the panel: copies of its last row, finite and never stored)
  {
    uint4 v[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int idx = tid + j * FF_THREADS, r = min(idx >> 6, nrows - 1);
      v[j] = *reinterpret_cast<const uint4*>(p.x + (int64_t)(r0 + r) * p.ldx + (idx & 63) * 8);
    }
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int idx = tid + j * FF_THREADS;
      *reinterpret_cast<uint4*>(img + (idx >> 6) * FF_RS + (idx & 63) * 16) = v[j];
    }
  }
  __syncthreads();
  f32x4_t acc[MT][FF_NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) acc[mt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // side jobs behind the weight stream: the bias of this lane's columns and the residual values of its accumulator positions
  float4 b4[FF_NT];
  bf16x4_t rv[MT][FF_NT];
#pragma unroll
  for (int u = 0; u < FF_NT; ++u) b4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) rv[mt][u] = ff_pack4(0.f, 0.f, 0.f, 0.f);
  auto side = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    if constexpr (kb == FF_KB - 4) {
      if (p.bias) {
#pragma unroll
        for (int u = 0; u < FF_NT; ++u) b4[u] = *reinterpret_cast<const float4*>(p.bias + colb + 16 * u);
      }
    }
    if constexpr (kb >= FF_KB - 4 && kb - (FF_KB - 4) < MT) {
      constexpr int mt = kb - (FF_KB - 4);
      if (p.res) {
        const int64_t row = r0 + min(16 * mt + l16, nrows - 1);
#pragma unroll
        for (int u = 0; u < FF_NT; ++u) rv[mt][u] = *reinterpret_cast<const bf16x4_t*>(p.res + row * p.ldres + colb + 16 * u);
      }
    }
  };
  const unsigned char* arow = img + l16 * FF_RS + 16 * g;
  lp_gemm_seq<MT>(wl, arow, wr, acc, side, std::make_integer_sequence<int, FF_KB>{});
  // ---- LayerNorm input rows, fp32, into the exchange (a GEMM lane owns 4 columns of 16 MT rows; a LayerNorm wave owns rows)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
      float4 t;
      t.x = (acc[mt][u][0] + b4[u].x) + (float)rv[mt][u][0];
      t.y = (acc[mt][u][1] + b4[u].y) + (float)rv[mt][u][1];
      t.z = (acc[mt][u][2] + b4[u].z) + (float)rv[mt][u][2];
      t.w = (acc[mt][u][3] + b4[u].w) + (float)rv[mt][u][3];
      *reinterpret_cast<float4*>(exch + (16 * mt + l16) * LP_XS + (colb + 16 * u) * 4) = t;
    }
  float gm[8], bt[8];
  ff_load8f(p.gamma + lane * 8, gm);
  ff_load8f(p.beta + lane * 8, bt);
  __syncthreads();
  const float invH = 1.f / (float)FF_D;
  for (int row = wave_s; row < nrows; row += FF_WAVES) {
    const float4 a = *reinterpret_cast<const float4*>(exch + row * LP_XS + lane * 32);
    const float4 b = *reinterpret_cast<const float4*>(exch + row * LP_XS + lane * 32 + 16);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const float mu = ff_row_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * invH;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] -= mu; q += v[k] * v[k]; }
    const float rs = 1.f / sqrtf(ff_row_sum(q) * invH + p.eps);
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = v[k] * rs * gm[k] + bt[k];
    *reinterpret_cast<uint4*>(p.y + (int64_t)(r0 + row) * p.ldy + lane * 8) = ff_pack8(o);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// y[:, 512 c ..] = x . W[512 c ..]^T + bias for a [3 x 512, 512] weight (the in-projection of a frozen-weight layer,
// TransformerBaseline.py:12-13) at a few thousand rows and below: workgroup (panel, chunk c) streams ONE packed [512 x 512]
// image for its <= 64 rows; with few panels (a few hundred rows) a chunk is cut into four blocks of 128 output columns
// (NTW = 1: one column tile per wave) so that the launch still covers most of the chip.  bf16 rows straight from the
// accumulators (8-byte stores: a lane owns 4 consecutive columns of a row).
struct LinPanelParams {
  const bf16_t* x; int64_t ldx;                            // [M, 512]
  const bf16_t* wp;                                        // packed images, chunk c at + c * FF_WELEMS
  const float* bias;                                       // [chunks * 512] or NULL
  bf16_t* y; int64_t ldy;                                  // [M, chunks * 512]
  int M, P, chunks;
};

template <int MT, int NTW>
__global__ __launch_bounds__(FF_THREADS) void linear_panel_kernel(LinPanelParams p) {
  constexpr int ROWS = 16 * MT, PIECES = ROWS * 64 / FF_THREADS, SUBS = FF_NT / NTW;
  __shared__ __attribute__((aligned(16))) unsigned char img[ROWS * FF_RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int g = lane >> 4, l16 = lane & 15;
  const int per_panel = p.chunks * SUBS;
  const int panel = (int)blockIdx.x / per_panel, cs = (int)blockIdx.x % per_panel;
  const int c = cs / SUBS, sb = cs % SUBS;
  const int r0 = panel * p.P;
  const int nrows = min(p.P, p.M - r0);
  // column tiles of this wave inside chunk c: NTW consecutive ones from nt0; its first output column inside the chunk
  const int nt0 = (32 / SUBS) * sb + NTW * wave_s;
  const int col0 = 16 * ((32 / SUBS) * sb + NTW * wave) + 4 * g;
  const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(p.wp + (int64_t)c * FF_WELEMS) + (int64_t)nt0 * 64 + ff_lane();
  bf16x8_t wr[3][NTW];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int u = 0; u < NTW; ++u) wr[s][u] = wl[(u + s * 32) * 64];
  {
    uint4 v[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int idx = tid + j * FF_THREADS, r = min(idx >> 6, nrows - 1);
      v[j] = *reinterpret_cast<const uint4*>(p.x + (int64_t)(r0 + r) * p.ldx + (idx & 63) * 8);
    }
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int idx = tid + j * FF_THREADS;
      *reinterpret_cast<uint4*>(img + (idx >> 6) * FF_RS + (idx & 63) * 16) = v[j];
    }
  }
  __syncthreads();
  f32x4_t acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int u = 0; u < NTW; ++u) acc[mt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float4 b4[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) b4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned char* arow = img + l16 * FF_RS + 16 * g;
#pragma unroll
  for (int kb = 0; kb < FF_KB; ++kb) {
    if (kb + 2 < FF_KB) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) wr[(kb + 2) % 3][u] = wl[(u + (kb + 2) * 32) * 64];
    }
    if (kb == FF_KB - 2 && p.bias) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) b4[u] = *reinterpret_cast<const float4*>(p.bias + c * FF_CHUNK + col0 + 16 * u);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t fa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * FF_RS + 64 * kb);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < NTW; ++u)
        acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[kb % 3][u], fa[mt], acc[mt][u], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = 16 * mt + l16;
    if (row < nrows) {
      bf16_t* yr = p.y + (int64_t)(r0 + row) * p.ldy + c * FF_CHUNK + col0;
#pragma unroll
      for (int u = 0; u < NTW; ++u)
        *reinterpret_cast<bf16x4_t*>(yr + 16 * u) = ff_pack4(acc[mt][u][0] + b4[u].x, acc[mt][u][1] + b4[u].y,
                                                             acc[mt][u][2] + b4[u].z, acc[mt][u][3] + b4[u].w);
    }
  }
}

// rows per panel / column blocks per chunk of the launch above: panels x chunks workgroups ~ one round of the chip, at least
// 16 rows; 16-row panels that leave most CUs idle take four 128-column blocks per chunk
inline int lin_panel_geometry(int64_t M, int64_t chunks, int* P, int* subs) {
  if (M <= 0 || chunks < 1 || chunks > 8 || M > 8192) return 0;
  const int64_t panels_max = 256 / chunks;
  int64_t rows = (M + panels_max - 1) / panels_max;
  if (rows < 16) rows = 16;
  if (rows > FF_ROWS) rows = FF_ROWS;
  const int64_t panels = (M + rows - 1) / rows;
  if (P) *P = (int)rows;
  if (subs) *subs = (rows == 16 && panels * chunks * 4 <= 320) ? 4 : 1;
  return (int)panels;
}

// ---------------------------------------------------------------------------------------------------------------------
// The feed-forward sublayer of a frozen-weight forward at a few thousand rows and below:
//     y = LayerNorm2( x1 + W2 . ReLU(W1 . x1 + b1) + b2 )                       TransformerBaseline.py:15-19,33-36
// It replaces Linear + ReLU ([M, 2048] hidden activation through HBM) and the split-K Linear + LayerNorm pair (12.7 + 23.7 us
// at M = 3200, 7.6 + 15.1 us at M = 200).  The training launch (ffn_chain.hip) gives every 50-row panel to ONE workgroup, which
// streams all 4 MB of W1 / W2: right when 256 panels fill the chip, 76 us when there are 64.  Here a panel is shared by the
// `nchunk` workgroups of its hidden chunks: workgroup (panel, c) computes h_c = ReLU(x1 . W1[c]^T + b1[c]) into LDS and the
// partial product y_c = h_c . W2[:, c]^T (1 MB of weights: two GEMM phases) and leaves y_c as an fp32 slab [c][M][512]; the
// slab reduction that finishes it IS the LayerNorm launch of gemm.hip (bias + residual + statistics + affine, slabs added in
// chunk order): two launches, no hidden activation in HBM.
// (Built first with the reduction inside the launch -- a ticket per panel, the last arriver normalises, agent-scope fences
// around the ticket: the fences write back and invalidate the XCD's WHOLE L2, every workgroup's weight stream then missed:
// 106 us instead of 36.  Per-access sc1 loads would need hand-scheduled waits; the kernel boundary is the cheap fence.)
// Workgroups of one panel have consecutive ids = different XCDs; an XCD sees only chunks id % 8 % nchunk: 1-2 MB of weights
// in its L2 instead of all of them.
struct FfnCoopParams {
  const bf16_t* x1;                 // [M, 512]
  const bf16_t* w1p; const bf16_t* w2p;      // packed images, chunk c at + c * FF_WELEMS
  const float* b1;
  float* part;                      // [nchunk][M][512] fp32 partial products (the slabs of the LayerNorm launch)
  int M, P, nchunk;
};

__global__ __launch_bounds__(FF_THREADS) void ffn_coop_kernel(FfnCoopParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FF_BUF];
  unsigned char* imgX = smem;
  unsigned char* imgH = smem + FF_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int g = lane >> 4, l16 = lane & 15;
  const int panel = (int)blockIdx.x / p.nchunk, c = (int)blockIdx.x % p.nchunk;
  const int r0 = panel * p.P;
  const int nrows = min(p.P, p.M - r0);
  const int colb = wave * 16 * FF_NT + 4 * g;
  ff_panel_in(p.x1 + (int64_t)r0 * FF_D, imgX, nrows, tid);
  __syncthreads();
  {
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
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
      const float bb[4] = {b4[u].x, b4[u].y, b4[u].z, b4[u].w};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc1[mt][u][r] + bb[r], 0.f);
        *reinterpret_cast<bf16x4_t*>(imgH + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) = ff_pack4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  __syncthreads();                       // image H = this chunk's hidden activation
  {
    f32x4_t accy[4][FF_NT];
    ff_zero(accy);
    auto side2 = [&](auto, int) {};
    ff_gemm<FF_RING_B>(p.w2p + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgH, ff_lane(), accy, side2);
    float* pc = p.part + ((int64_t)c * p.M + r0) * FF_D;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = 16 * mt + l16;
      if (row < nrows) {
#pragma unroll
        for (int u = 0; u < FF_NT; ++u)
          *reinterpret_cast<f32x4_t*>(pc + (int64_t)row * FF_D + colb + 16 * u) = accy[mt][u];
      }
    }
  }
}

// ---- the same for a FEW HUNDRED rows (the sampler at B = 1, 2: M = 200, 400): 16-row panels x nchunk workgroups are 52 / 100
// workgroups streaming 1 MB each.  Here a hidden chunk is cut into SUB sub-chunks of 512 / SUB columns: workgroup
// (panel, c, s) computes h = ReLU(x1 . W1[cols]^T + b1[cols]) for its 128 or 256 hidden columns (every wave 16 / SUB ... one
// or two MFMA column tiles over the 16 k-blocks) and the partial product over THOSE reduction columns (4 or 8 k-blocks, all
// 512 output columns): 256 / 512 KB of weights per workgroup, nchunk x SUB slabs for the LayerNorm launch.
template <int NT, int KB0, int KBN, class Side>
__device__ __forceinline__ void sm_gemm(const bf16x8_t* __restrict__ wl, int kb_first, const unsigned char* arow,
                                        f32x4_t (&acc)[1][NT], Side&& side) {
  // block (u, j) of this phase = wl[(u + (kb_first + j) * 32) * 64]; A fragment j at arow + 64 * (KB0 + j)
  constexpr int RING = 3, PD = RING - 1;
  bf16x8_t wr[RING][NT];
#pragma unroll
  for (int s = 0; s < PD && s < KBN; ++s)
#pragma unroll
    for (int u = 0; u < NT; ++u) wr[s][u] = wl[(u + (kb_first + s) * 32) * 64];
#pragma unroll
  for (int j = 0; j < KBN; ++j) {
    if (j + PD < KBN) {
#pragma unroll
      for (int u = 0; u < NT; ++u) wr[(j + PD) % RING][u] = wl[(u + (kb_first + j + PD) * 32) * 64];
    }
    side(j);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(arow + 64 * (KB0 + j));
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j % RING][u], fa, acc[0][u], 0, 0, 0);
  }
}

template <int SUB>
__global__ __launch_bounds__(FF_THREADS) void ffn_coop_small_kernel(FfnCoopParams p) {
  constexpr int HC = FF_CHUNK / SUB;                        // hidden columns of this workgroup
  constexpr int NT1 = HC / (16 * FF_WAVES);                 // column tiles per wave in the first GEMM (1 or 2)
  constexpr int KB2 = HC / 32;                              // k-blocks of the second GEMM (4 or 8)
  static_assert(NT1 >= 1 && NT1 * 16 * FF_WAVES == HC, "sub-chunk = whole column tiles per wave");
  __shared__ __attribute__((aligned(16))) unsigned char imgX[16 * FF_RS];
  __shared__ __attribute__((aligned(16))) unsigned char imgH[16 * FF_RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int g = lane >> 4, l16 = lane & 15;
  const int per_panel = p.nchunk * SUB;
  const int panel = (int)blockIdx.x / per_panel, cs = (int)blockIdx.x % per_panel;
  const int c = cs / SUB, sb = cs % SUB;
  const int r0 = panel * 16;
  const int nrows = min(16, p.M - r0);
  {
    uint4 v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + j * FF_THREADS, r = min(idx >> 6, nrows - 1);
      v[j] = *reinterpret_cast<const uint4*>(p.x1 + (int64_t)(r0 + r) * FF_D + (idx & 63) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + j * FF_THREADS;
      *reinterpret_cast<uint4*>(imgX + (idx >> 6) * FF_RS + (idx & 63) * 16) = v[j];
    }
  }
  __syncthreads();
  const int ln = ff_lane();
  {
    // hidden columns HC * sb + 16 * NT1 * wave ... of chunk c: column tiles nt = (HC / 16) * sb + NT1 * wave + u
    f32x4_t acc1[1][NT1];
#pragma unroll
    for (int u = 0; u < NT1; ++u) acc1[0][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nt0 = (HC / 16) * sb + NT1 * wave_s;
    const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(p.w1p + (int64_t)c * FF_WELEMS) + (int64_t)nt0 * 64 + ln;
    float4 b4[NT1];
    const int hcol = HC * sb + 16 * NT1 * wave + 4 * g;     // this lane's first hidden column inside the chunk
    auto side1 = [&](int j) {
      if (j == FF_KB - 1) {
#pragma unroll
        for (int u = 0; u < NT1; ++u) b4[u] = *reinterpret_cast<const float4*>(p.b1 + c * FF_CHUNK + hcol + 16 * u);
      }
    };
    sm_gemm<NT1, 0, FF_KB>(wl, 0, imgX + l16 * FF_RS + 16 * g, acc1, side1);
#pragma unroll
    for (int u = 0; u < NT1; ++u) {
      // image H holds the sub-chunk's HC columns from column 0
      const int col = 16 * NT1 * wave + 4 * g + 16 * u;
      *reinterpret_cast<bf16x4_t*>(imgH + l16 * FF_RS + col * 2) =
          ff_pack4(fmaxf(acc1[0][u][0] + b4[u].x, 0.f), fmaxf(acc1[0][u][1] + b4[u].y, 0.f),
                   fmaxf(acc1[0][u][2] + b4[u].z, 0.f), fmaxf(acc1[0][u][3] + b4[u].w, 0.f));
    }
  }
  __syncthreads();
  {
    f32x4_t accy[1][FF_NT];
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) accy[0][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(p.w2p + (int64_t)c * FF_WELEMS) + (int64_t)(wave_s * FF_NT) * 64 + ln;
    auto side2 = [&](int) {};
    sm_gemm<FF_NT, 0, KB2>(wl, KB2 * sb, imgH + l16 * FF_RS + 16 * g, accy, side2);
    const int colb = wave * 16 * FF_NT + 4 * g;
    float* pc = p.part + ((int64_t)(c * SUB + sb) * p.M + r0) * FF_D;
    if (l16 < nrows) {
#pragma unroll
      for (int u = 0; u < FF_NT; ++u) *reinterpret_cast<f32x4_t*>(pc + (int64_t)l16 * FF_D + colb + 16 * u) = accy[0][u];
    }
  }
}

// rows per panel: panels x nchunk workgroups should fill the chip once (256 / nchunk panels), at least 16 rows (one MFMA
// row tile), at most 64 (the LDS images; beyond 64 x 256 / nchunk rows the launch takes several rounds)
// `sub` (hidden sub-chunks per chunk): 1 unless 16-row panels x nchunk leave most of the chip idle -- then 2 or 4, whichever
// brings the workgroup count nearest to (and not far beyond) 256
inline int ffn_coop_geometry(int64_t M, int64_t d, int64_t ffn, int* P, int* nchunk, int* sub = nullptr) {
  if (M <= 0 || d != FF_D || ffn <= 0 || ffn % FF_CHUNK != 0 || ffn / FF_CHUNK > FF_MAXCHUNK) return 0;
  const int nc = (int)(ffn / FF_CHUNK);
  const int64_t panels_max = 256 / nc;
  int64_t rows = (M + panels_max - 1) / panels_max;
  if (rows < 16) rows = 16;
  if (rows > FF_ROWS) rows = FF_ROWS;                       // more than one round of workgroups
  if (M > 32768) return 0;
  const int64_t panels = (M + rows - 1) / rows;
  int sb = 1;
  if (rows == 16) {
    if (panels * nc * 4 <= 320) sb = 4;
    else if (panels * nc * 2 <= 320) sb = 2;
  }
  if (P) *P = (int)rows;
  if (nchunk) *nchunk = nc;
  if (sub) *sub = sb;
  return (int)panels;
}

}  // namespace

// rows per workgroup and workgroup count: ceil(M / 256) rows while that is at most 32, else unsupported (large batches take
// the tile GEMM + LayerNorm: they fill the chip without streaming the whole weight per 32 rows)
extern "C" int ib_linear_ln_panel_workgroups(int64_t M, int64_t N, int64_t K, int32_t* rows_out) {
  if (M <= 0 || N != FF_D || K != FF_D) return 0;
  const int64_t P = (M + 255) / 256;
  if (P > 32) return 0;
  if (rows_out) *rows_out = (int32_t)P;
  return (int)((M + P - 1) / P);
}

extern "C" int ib_linear_ln_panel_fwd(const void* x, int64_t ldx, const void* w_packed, const float* bias, const void* res,
                                      int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, int64_t M,
                                      int64_t N, int64_t K, float eps, ib_stream_t stream) {
  if (!x || !w_packed || !gamma || !beta || !y || M <= 0) return IB_E_ARG;
  int32_t P = 0;
  const int nwg = ib_linear_ln_panel_workgroups(M, N, K, &P);
  if (!nwg) return IB_E_UNSUPPORTED;
  if (ldx < K || ldy < N || (res && ldres < N) || ldx % 8 != 0 || ldy % 8 != 0 || (res && ldres % 4 != 0)) return IB_E_ARG;
  if (!ff_al16({x, w_packed, gamma, beta, y}) || (bias && !ff_al16({bias})) ||
      (res && (reinterpret_cast<uintptr_t>(res) % 8) != 0))
    return IB_E_ARG;
  LinLnPanelParams p{};
  p.x = (const bf16_t*)x; p.ldx = ldx; p.wp = (const bf16_t*)w_packed; p.bias = bias;
  p.res = (const bf16_t*)res; p.ldres = ldres; p.gamma = gamma; p.beta = beta; p.y = (bf16_t*)y; p.ldy = ldy;
  p.M = (int)M; p.P = P; p.eps = eps;
  IB_PATH(IB_PATH_LINLN_PANEL);
  if (P <= 16) hipLaunchKernelGGL(linln_panel_kernel<1>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL(linln_panel_kernel<2>, dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" size_t ib_ffn_infer_workspace(int64_t M, int64_t d, int64_t ffn) {
  int nc = 0, sb = 1;
  if (!ffn_coop_geometry(M, d, ffn, nullptr, &nc, &sb)) return 0;
  return (size_t)nc * sb * (size_t)M * FF_D * sizeof(float);
}
extern "C" int ib_ffn_infer_workgroups(int64_t M, int64_t d, int64_t ffn, int32_t* rows_out, int32_t* panels_out) {
  int P = 0, nc = 0, sb = 1;
  const int panels = ffn_coop_geometry(M, d, ffn, &P, &nc, &sb);
  if (!panels) return 0;
  if (rows_out) *rows_out = P;
  if (panels_out) *panels_out = panels;
  return panels * nc * sb;
}
int ib_slab_ln512_launch(const float* slabs, int nslab, int64_t slab_stride, const float* bias, const void* res, int64_t ldres,
                         const float* gamma, const float* beta, void* y, int64_t ldy, int64_t M, float eps, hipStream_t s);   // gemm.hip

extern "C" int ib_ffn_infer_fwd(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                                const float* beta, void* y, void* workspace, size_t workspace_bytes, int64_t M, int64_t d,
                                int64_t ffn, float eps, ib_stream_t stream) {
  if (!x1 || !packed || !b1 || !b2 || !gamma || !beta || !y || !workspace) return IB_E_ARG;
  int P = 0, nc = 0, sb = 1;
  const int panels = ffn_coop_geometry(M, d, ffn, &P, &nc, &sb);
  if (!panels) return IB_E_UNSUPPORTED;
  if (workspace_bytes < ib_ffn_infer_workspace(M, d, ffn)) return IB_E_WORKSPACE;
  if (!ff_al16({x1, packed, b1, b2, gamma, beta, y, workspace})) return IB_E_ARG;
  FfnCoopParams p{};
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.x1 = (const bf16_t*)x1; p.w1p = pk; p.w2p = pk + (int64_t)nc * FF_WELEMS;
  p.b1 = b1; p.part = reinterpret_cast<float*>(workspace);
  p.M = (int)M; p.P = P; p.nchunk = nc;
  IB_PATH(IB_PATH_FFN_INFER);
  if (sb == 4) hipLaunchKernelGGL(ffn_coop_small_kernel<4>, dim3(panels * nc * 4), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (sb == 2) hipLaunchKernelGGL(ffn_coop_small_kernel<2>, dim3(panels * nc * 2), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL(ffn_coop_kernel, dim3(panels * nc), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return ib_slab_ln512_launch(p.part, nc * sb, (int64_t)M * FF_D, b2, x1, FF_D, gamma, beta, y, FF_D, M, eps, ib_s(stream));
}

extern "C" int ib_linear_panel_workgroups(int64_t M, int64_t N, int64_t K) {
  if (K != FF_D || N <= 0 || N % FF_CHUNK != 0) return 0;
  int P = 0, subs = 1;
  const int panels = lin_panel_geometry(M, N / FF_CHUNK, &P, &subs);
  return panels * (int)(N / FF_CHUNK) * subs;
}

extern "C" int ib_linear_panel_fwd(const void* x, int64_t ldx, const void* w_packed, const float* bias, void* y, int64_t ldy,
                                   int64_t M, int64_t N, int64_t K, ib_stream_t stream) {
  if (!x || !w_packed || !y || M <= 0) return IB_E_ARG;
  if (K != FF_D || N <= 0 || N % FF_CHUNK != 0) return IB_E_UNSUPPORTED;
  int P = 0, subs = 1;
  const int chunks = (int)(N / FF_CHUNK);
  const int panels = lin_panel_geometry(M, chunks, &P, &subs);
  if (!panels) return IB_E_UNSUPPORTED;
  if (ldx < K || ldy < N || ldx % 8 != 0 || ldy % 4 != 0) return IB_E_ARG;
  if (!ff_al16({x, w_packed}) || (bias && !ff_al16({bias})) || (reinterpret_cast<uintptr_t>(y) % 8) != 0) return IB_E_ARG;
  LinPanelParams p{};
  p.x = (const bf16_t*)x; p.ldx = ldx; p.wp = (const bf16_t*)w_packed; p.bias = bias; p.y = (bf16_t*)y; p.ldy = ldy;
  p.M = (int)M; p.P = P; p.chunks = chunks;
  IB_PATH(IB_PATH_LIN_PANEL);
  const dim3 grid((unsigned)(panels * chunks * subs));
  hipStream_t s = ib_s(stream);
  if (subs == 4) hipLaunchKernelGGL((linear_panel_kernel<1, 1>), grid, dim3(FF_THREADS), 0, s, p);
  else if (P <= 16) hipLaunchKernelGGL((linear_panel_kernel<1, 4>), grid, dim3(FF_THREADS), 0, s, p);
  else if (P <= 32) hipLaunchKernelGGL((linear_panel_kernel<2, 4>), grid, dim3(FF_THREADS), 0, s, p);
  else hipLaunchKernelGGL((linear_panel_kernel<4, 4>), grid, dim3(FF_THREADS), 0, s, p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
