// Fused training chain of the token-wise MLP denoiser (BASELINE.json configs[1]) for gfx950.
//
// One launch = q_sample + every forward block (GEMM + bias + time embedding + SiLU + LayerNorm) + head GEMM +
// MSE loss / dL/dpred + the whole dgrad chain (head^T GEMM, LayerNorm/SiLU backward, block^T GEMMs).  The weight
// gradients (reductions over ALL tokens) stay separate GEMMs; this kernel leaves their operands in HBM:
//   xt, h_i (block outputs), dz_i (gradients w.r.t. the block pre-activations), dpred.
//
// Why a chain kernel: the network is token-local (rows never mix) and its weights are ~1 MB in bf16, so a workgroup
// that owns a panel of <= 64 tokens can run the whole network with its activations resident in LDS and the weights
// streamed from L2.  The unfused plan moved every [M,512] activation through HBM 2-4 times and paid ~25 launches.
//
// Geometry: 512 threads = 8 waves; the panel's activations are the MFMA "b" operand (64 rows = 4 m-tiles), each
// wave owns a distinct slice of the layer's OUTPUT columns (NT n-tiles of 16) and therefore a distinct slice of the
// weight matrix -> weights go global -> VGPR directly (no LDS round trip, no sharing to exploit), from a
// fragment-major packed image (ib_mlp_chain_pack) so that every wave-instruction reads one contiguous 1 KiB block.
// MFMA is issued "swapped" (a = weights, b = activations): a lane ends up with 4 consecutive output columns of one
// row, so bias / embedding / gamma loads and every store are 8- or 16-byte pieces.
// LayerNorm row statistics: 16-lane-group shuffles + one cross-wave exchange through LDS, fixed order (deterministic).
#include "ib_common.h"
#include "time_bwd.h"
#include <stdlib.h>

namespace {

constexpr int CH_ROWS = 64, CH_WAVES = 8, CH_THREADS = 512, CH_MAXL = 4;
#ifdef IB_AB
long long* g_chain_prof = nullptr;     // TIMING-ONLY, measurement builds
#endif
#define CH_STAMP(k) do { if (p.prof && tid == 0) p.prof[blockIdx.x * 64 + (k)] = wall_clock64(); } while (0)

struct ChainParams {
  const bf16_t* x0; const bf16_t* eps; const int64_t* t;
  const void* const* slots;        // optional device array {x0, eps, t}: read at kernel start instead of the three above (a
                                   // captured graph then consumes each step's batch in place, no staging copy)
  const float* sqrt_ab; const float* sqrt_1mab; int table_rows;
  const bf16_t* e; int64_t ld_e;
  int M, T, D, L, P;             // P = tokens per workgroup (<= 64)
  const bf16_t* wf[CH_MAXL + 1];   // packed forward weights: blocks 0..L-1, then the head
  const bf16_t* wb[CH_MAXL + 1];   // packed TRANSPOSED weights: wb[i] = blocks.i.linear^T (1..L-1), wb[L] = head^T
  const float* bias[CH_MAXL + 1];
  const float* gamma[CH_MAXL]; const float* beta[CH_MAXL];
  bf16_t* xt; int64_t ld_xt;
  bf16_t* u[CH_MAXL]; bf16_t* h[CH_MAXL]; bf16_t* dz[CH_MAXL];
  bf16_t* dpred; int64_t ld_dpred;
  float* partial; int64_t ld_part; // [gridDim.x, W] per-workgroup column sums: per block dgamma | dbeta | dbias (3H each
                                   // block), then the head-bias sums (128*NTD columns), then the squared-error sum
  bf16_t* de_lp; int64_t ld_de;    // optional (panel == window): bf16 [B, L*H] time-embedding gradient rows
  float gscale, ln_eps;
  long long* prof;                 // TIMING-ONLY (tools/chain_prof.py): [gridDim.x][64] wall-clock stamps, else NULL
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= the 16 token rows a lane group holds); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// keeps per-phase index arithmetic (divisions by runtime row widths) from being computed once and spilled across phases
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// lane id recomputed at every use (2 VALU): a cached copy of threadIdx.x was spilled and its reload -- a VMEM
// operation, retired in order -- stalled on the weight prefetch issued just before it
__device__ __forceinline__ int lane_id_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each; __frcp_rn / 1.f/x expand to the 10-instruction IEEE division sequence)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ bf16x4_t pack4(float a, float b, float c, float d) {
  bf16x4_t o;
  o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
  return o;
}

// acc[mt][u] += W_eff[16*(nt0+u) .. +15][:] . A[16*mt .. +15][:]^T over KB k-blocks of 32.
// Packed weights: block (nt, kb) is 1 KiB at ((kb * NTOT + nt) * 64 + lane) * 16 bytes, NTOT = 8 * NT n-tiles: the 32 blocks
// the eight waves of a workgroup request at one k-step are 32 KiB CONTIGUOUS (k-block-major).  (n-tile-major, block
// (nt, kb) at nt * KB + kb, put those 32 requests 16 KiB apart -- a power-of-two stride that lands them on the same few
// L2 channels; IB_CHAIN_KBMAJOR=0 builds that layout for the A/B.)
#ifndef IB_CHAIN_KBMAJOR
#define IB_CHAIN_KBMAJOR 1
#endif
// IB_CHAIN_EPS_LATE (round 4): the head phase requests the panel's eps rows (the loss target) four k-blocks before its GEMM
// ends instead of ahead of it, and sums the loss with DPP adds; = 0 rebuilds the round-3 head for the A/B (59.9 -> 57.8-58.4 us)
#ifndef IB_CHAIN_EPS_LATE
#define IB_CHAIN_EPS_LATE 1
#endif
// Prefetch ring of 3 k-blocks (2 in flight while one is consumed); the loop is fully unrolled so ring slots are
// static registers.  `side(kb)` is called once per k-block: the row copies of the neighbouring phases (stores of the
// previous output image, loads of the next epilogue's operands) are issued a piece per k-block BEHIND the first weight
// loads -- vmcnt retires in order, so a burst of stores issued ahead of the weight stream delayed every GEMM phase by
// the stores' round trip (+5 us per phase).
template <int V> struct IntC { static constexpr int value = V; };
template <int NT, int KB, class Side>
__device__ __forceinline__ void chain_gemm(const bf16_t* __restrict__ wp, int nt0, const unsigned char* abuf, int rs,
                                           int lane, f32x4_t (&acc)[4][NT], Side&& side) {
#ifndef IB_CHAIN_RING
#define IB_CHAIN_RING 3
#endif
  constexpr int RING = IB_CHAIN_RING, PD = RING - 1;
#if IB_CHAIN_KBMAJOR
  constexpr int SU = 1, SK = CH_WAVES * NT;       // block strides of the n-tile / k-block index
  const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(wp) + (int64_t)nt0 * 64 + lane;
#else
  constexpr int SU = KB, SK = 1;
  const bf16x8_t* wl = reinterpret_cast<const bf16x8_t*>(wp) + ((int64_t)nt0 * KB) * 64 + lane;
#endif
  const unsigned char* arow = abuf + (lane & 15) * rs + 16 * (lane >> 4);
  bf16x8_t wr[RING][NT];
#pragma unroll
  for (int s = 0; s < PD && s < KB; ++s)
#pragma unroll
    for (int u = 0; u < NT; ++u) wr[s][u] = wl[(u * SU + s * SK) * 64];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + PD < KB) {
#pragma unroll
      for (int u = 0; u < NT; ++u) wr[(kb + PD) % RING][u] = wl[(u * SU + (kb + PD) * SK) * 64];
    }
    side(kb, IntC<KB>{});      // piece j of a side job runs at k-block j % KB
    // pin the issue point of this k-block's prefetch: at the register limit hipcc otherwise sinks every weight load
    // to just before its first MFMA (prefetch distance 0: a full L2 round trip per k-block)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t fa[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * rs + 64 * kb);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[kb % RING][u], fa[mt], acc[mt][u], 0, 0, 0);
  }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[4][NT]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[mt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// Cross-wave row reduction of TWO quantities in one exchange: each lane holds partial sums for NM of its rows (m-tiles
// mt0 .. mt0+NM-1) over this wave's columns.  red: [64 rows][8 waves] float2.  Returns the full-row totals (fixed order).
template <int NM>
__device__ __forceinline__ void row_reduce2(float (&a)[NM], float (&b)[NM], float2* red, int lane, int wave, int mt0) {
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    a[m] += __shfl_xor(a[m], 16, 64); b[m] += __shfl_xor(b[m], 16, 64);
    a[m] += __shfl_xor(a[m], 32, 64); b[m] += __shfl_xor(b[m], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int m = 0; m < NM; ++m) red[(16 * (mt0 + m) + lane) * CH_WAVES + wave] = make_float2(a[m], b[m]);
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const float4* r4 = reinterpret_cast<const float4*>(red + (16 * (mt0 + m) + (lane & 15)) * CH_WAVES);
    const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2], p3 = r4[3];
    a[m] = ((((((p0.x + p0.z) + p1.x) + p1.z) + p2.x) + p2.z) + p3.x) + p3.z;
    b[m] = ((((((p0.y + p0.w) + p1.y) + p1.w) + p2.y) + p2.w) + p3.y) + p3.w;
  }
}

// NTH: n-tiles per wave for the hidden width (H = 128 * NTH);  NTD / KBD: n-tiles per wave and k-blocks for the
// feature width (D <= 128 * NTD, D <= 32 * KBD).
// IB_CHAIN_NHF: the backward epilogue walks the panel's four m-tiles in NHF groups; the fewer m-tiles are live at once, the
// fewer registers spill -- and a spill reload is a vector-memory operation that retires in order behind the weight
// prefetch.  Measured (B = 256, T = 50): NHF = 1 736 B of scratch per lane; NHF = 2 208 B, 97-101 us; NHF = 4 72 B, 88 us.
#ifndef IB_CHAIN_NHF
#define IB_CHAIN_NHF 4
#endif
constexpr int CH_NWIN = 8;                                 // windows whose time embedding is staged per panel
template <int NTH, int NTD, int KBD>
struct ChainCfg {
  static constexpr int H = 128 * NTH, KBH = 4 * NTH, DP = 32 * KBD, DN = 128 * NTD;
  static constexpr int WMAX = (H > DP ? (H > DN ? H : DN) : (DP > DN ? DP : DN));
  static constexpr int RS = WMAX * 2 + 16;                 // LDS row stride (bytes): +16 -> conflict-free b128 reads
  static constexpr int BUF = CH_ROWS * RS;
  static constexpr int RED = CH_ROWS * CH_WAVES * 8;       // one cross-wave exchange array (float2)
  static constexpr int STATS = CH_MAXL * CH_ROWS * 2 * 4;
  static constexpr int EIMG = CH_NWIN * H * 2;
  static constexpr int LDS = 2 * BUF + 2 * RED + STATS + EIMG + 64;
  static constexpr int CSS = 16 * NTH + 4;                 // column-sum scratch row stride (floats)
  static_assert(CH_WAVES * 16 * CSS * 4 <= BUF, "column-sum scratch must fit one image buffer");
};

// ---- coalesced row movers between HBM and an LDS image (row stride rs bytes).  Per-lane epilogue accesses are
// 16 rows x 32 B per wave-instruction (store-issue-bound: ~9k cycles per 16 of them); whole rows move as 16-byte pieces,
// one piece per thread per call (piece j of thread tid is element tid + 512 j of the row-major piece grid).
// Branch-free: rows beyond the panel are CLAMPED to its last row (a duplicate store of identical bytes / a finite
// duplicate load whose consumers are masked).  A divergent branch around a load or store inside the GEMM loop makes
// hipcc fall back to s_waitcnt vmcnt(0) at the join, which drains the weight prefetch ring every k-block.
__device__ __forceinline__ void out_piece16(const unsigned char* img, int rs, bf16_t* g, int64_t ldg, int nrows, int ppr,
                                            int idx) {
  const int row = min(idx / ppr, nrows - 1), pc = idx % ppr;
  const uint4 v = *reinterpret_cast<const uint4*>(img + row * rs + pc * 16);
  *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(g + (int64_t)row * ldg) + pc * 16) = v;
}
__device__ __forceinline__ void out_piece8(const unsigned char* img, int rs, bf16_t* g, int64_t ldg, int nrows, int ppr,
                                           int idx) {
  const int row = min(idx / ppr, nrows - 1), pc = idx % ppr;
  const uint2 v = *reinterpret_cast<const uint2*>(img + row * rs + pc * 8);
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(g + (int64_t)row * ldg) + pc * 8) = v;
}
__device__ __forceinline__ uint4 in_piece16(const bf16_t* g, int64_t ldg, int nrows, int ppr, int idx) {
  const int row = min(idx / ppr, nrows - 1), pc = idx % ppr;
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(g + (int64_t)row * ldg) + pc * 16);
}
__device__ __forceinline__ uint2 in_piece8(const bf16_t* g, int64_t ldg, int nrows, int ppr, int idx) {
  const int row = min(idx / ppr, nrows - 1), pc = idx % ppr;
  return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(g + (int64_t)row * ldg) + pc * 8);
}
template <int NP>
__device__ __forceinline__ void store_rows16(unsigned char* img, int rs, int ppr, int tid, const uint4 (&r)[NP]) {
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int idx = tid + j * CH_THREADS;
    const int row = idx / ppr, pc = idx - row * ppr;
    if (row < CH_ROWS) *reinterpret_cast<uint4*>(img + row * rs + pc * 16) = r[j];
  }
}
template <int NP>
__device__ __forceinline__ void store_rows8(unsigned char* img, int rs, int ppr, int tid, const uint2 (&r)[NP]) {
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int idx = tid + j * CH_THREADS;
    const int row = idx / ppr, pc = idx - row * ppr;
    if (row < CH_ROWS) *reinterpret_cast<uint2*>(img + row * rs + pc * 8) = r[j];
  }
}

// column sums over the panel's 64 rows of a per-lane quantity q[u][r] (already summed over the lane's 4 m-tiles):
// through a wave-private LDS scratch [16 rows][16*NT columns] -- 4 float4 writes + 16 reads per lane instead of 64
// DPP steps.  Lane j < 16*NT returns the total of column (16*NT*wave + j); other lanes return 0.
// Lanes exchange data through LDS WITHOUT a workgroup barrier here (the scratch is wave-private and a wave's LDS
// operations execute in order), so the compiler must be told not to move the accesses across the exchange points.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int NT>
__device__ __forceinline__ float wave_colsum(const float (&q)[NT][4], float* scr, int css, int lane) {
  const int l16 = lane & 15, g = lane >> 4;
  wave_sync_lds();                 // earlier reads of the scratch (previous quantity) are done
#pragma unroll
  for (int u = 0; u < NT; ++u)
    *reinterpret_cast<float4*>(scr + l16 * css + 16 * u + 4 * g) = make_float4(q[u][0], q[u][1], q[u][2], q[u][3]);
  wave_sync_lds();
  float s = 0.f;
  if (lane < 16 * NT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s += scr[r * css + lane];
  }
  return s;
}

template <int NTH, int NTD, int KBD>
__global__ __launch_bounds__(CH_THREADS) void mlp_chain_kernel(ChainParams p) {
  using C = ChainCfg<NTH, NTD, KBD>;
  constexpr int H = C::H, RS = C::RS, PPR = H / 8, NPH = 2 * NTH;   // NPH: 16-byte pieces per thread of a [64, H] image
  constexpr int NHF = IB_CHAIN_NHF, MPH = 4 / NHF;                  // backward epilogue: m-tile groups processed one after another
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  unsigned char* buf0 = smem;
  unsigned char* buf1 = smem + C::BUF;
  float2* redA = reinterpret_cast<float2*>(smem + 2 * C::BUF);
  float2* redB = redA + CH_ROWS * CH_WAVES;
  float* stats = reinterpret_cast<float*>(redB + CH_ROWS * CH_WAVES);   // [L][64][2] mean, rstd
  unsigned char* eimg = reinterpret_cast<unsigned char*>(stats + CH_MAXL * CH_ROWS * 2);   // [CH_NWIN][H] bf16
  float* lossred = reinterpret_cast<float*>(eimg + C::EIMG);            // [8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);     // scalar copy: thread ids for the row movers are rebuilt
#define TIDV ((wave_s << 6) | lane_id_now())                    // from it, never kept in a (spillable) VGPR
  const int g = lane >> 4, l16 = lane & 15;
  const int r0 = blockIdx.x * p.P;
  const int D = p.D, M = p.M;
  const bf16_t* in_x0 = p.slots ? reinterpret_cast<const bf16_t*>(p.slots[0]) : p.x0;
  const bf16_t* in_eps = p.slots ? reinterpret_cast<const bf16_t*>(p.slots[1]) : p.eps;
  const int64_t* in_t = p.slots ? reinterpret_cast<const int64_t*>(p.slots[2]) : p.t;
  const int nrows = min(p.P, M - r0);                      // valid token rows of this panel
  const float invH = 1.f / (float)H;
  CH_STAMP(0);

  int rowg[4]; bool valid[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int lr = 16 * mt + l16;
    valid[mt] = lr < nrows;
    rowg[mt] = min(r0 + lr, M - 1);
  }
  // time-embedding rows of the panel's windows are staged in LDS (a panel of <= 64 tokens spans few windows; all 16
  // rows a lane group touches usually share ONE row of e: per-lane global loads of it were 16 x (16 rows x 32 B))
  const int w0 = r0 / p.T;
  const int nwin = (r0 + nrows - 1) / p.T - w0 + 1;
  const bool estage = nwin <= CH_NWIN;
  int ewl[4];                                              // local window of the lane's 4 rows
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ewl[mt] = min(rowg[mt] / p.T - w0, CH_NWIN - 1);

  // ---- q_sample: xt = sqrt_ab[t] x0 + sqrt_1mab[t] eps -> LDS image (zero-padded to DP columns / 64 rows) + HBM.
  // The per-row coefficients (two dependent loads) and the x0 / eps rows are fetched concurrently.
  {
    const int ppr = D >> 2;                                // 8-byte pieces per row
    uint2 rx[KBD], re[KBD];
#pragma unroll
    for (int j = 0; j < KBD; ++j) {
      rx[j] = in_piece8(in_x0 + (int64_t)r0 * D, D, nrows, ppr, TIDV + j * CH_THREADS);
      re[j] = in_piece8(in_eps + (int64_t)r0 * D, D, nrows, ppr, TIDV + j * CH_THREADS);
    }
    float2* coef = redA;
    if (tid < CH_ROWS) {
      float2 c = make_float2(0.f, 0.f);
      if (tid < nrows) {
        int64_t k = in_t[(r0 + tid) / p.T];
        k = k < 0 ? 0 : (k >= p.table_rows ? p.table_rows - 1 : k);
        c = make_float2(p.sqrt_ab[k], p.sqrt_1mab[k]);
      }
      coef[tid] = c;
    }
    // zero the whole image first (pad columns / rows), then the computed pieces overwrite their slots
    for (int idx = tid; idx < CH_ROWS * (C::DP / 8); idx += CH_THREADS) {
      const int row = idx / (C::DP / 8), pc = idx % (C::DP / 8);
      *reinterpret_cast<uint4*>(buf0 + row * RS + pc * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KBD; ++j) {
      const int idx = tid + j * CH_THREADS;
      const int row = idx / ppr, pc = idx - row * ppr;
      if (row < nrows) {
        const float2 c = coef[row];
        const bf16x4_t xv = __builtin_bit_cast(bf16x4_t, rx[j]), ev = __builtin_bit_cast(bf16x4_t, re[j]);
        const bf16x4_t o = pack4(c.x * (float)xv[0] + c.y * (float)ev[0], c.x * (float)xv[1] + c.y * (float)ev[1],
                                 c.x * (float)xv[2] + c.y * (float)ev[2], c.x * (float)xv[3] + c.y * (float)ev[3]);
        *reinterpret_cast<bf16x4_t*>(p.xt + (int64_t)(r0 + row) * p.ld_xt + pc * 4) = o;
        *reinterpret_cast<bf16x4_t*>(buf0 + row * RS + pc * 8) = o;
      }
    }
  }
  __syncthreads();
  CH_STAMP(1);

  unsigned char* cur = buf0;
  unsigned char* nxt = buf1;

  // ---- forward blocks
  for (int i = 0; i < p.L; ++i) {
    f32x4_t acc[4][NTH];
    zero_acc<NTH>(acc);
    // side jobs of the GEMM phase: the previous block's output rows go out (the `cur` image), this block's slice of
    // the time embedding comes in
    uint4 er = make_uint4(0u, 0u, 0u, 0u);
    const bf16_t* esrc = p.e + (int64_t)w0 * p.ld_e + (int64_t)i * H;
    if (i == 0) {
      auto side = [&](int kb, auto) {
        if (kb == 0) er = in_piece16(esrc, p.ld_e, min(nwin, CH_NWIN), PPR, TIDV);
      };
      chain_gemm<NTH, KBD>(p.wf[0], wave_s * NTH, cur, RS, lane_id_now(), acc, side);
    } else {
      bf16_t* hprev = p.h[i - 1] + (int64_t)r0 * H;
      auto side = [&](int kb, auto kbc) {
        constexpr int KB = decltype(kbc)::value;
        if (kb == 0) er = in_piece16(esrc, p.ld_e, min(nwin, CH_NWIN), PPR, TIDV);
#pragma unroll
        for (int j = 0; j < NPH; ++j)
          if (j % KB == kb) out_piece16(cur, RS, hprev, H, nrows, PPR, TIDV + j * CH_THREADS);
      };
      chain_gemm<NTH, C::KBH>(p.wf[i], wave_s * NTH, cur, RS, lane_id_now(), acc, side);
    }
    if (tid < CH_NWIN * PPR) *reinterpret_cast<uint4*>(eimg + tid * 16) = er;
    CH_STAMP(2 + 4 * i);
    __syncthreads();                       // every wave is done reading `cur`: it becomes the u image; e image complete
    const int colb = wave * 16 * NTH + 4 * g;
    // u = z + bias + e (bf16) ; v = silu(u) ; one-pass row statistics (sum, sum of squares)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 b4[NTH];
#pragma unroll
    for (int u = 0; u < NTH; ++u) b4[u] = *reinterpret_cast<const float4*>(p.bias[i] + colb + 16 * u);
    auto pass_u = [&](auto stagedc) {
      constexpr bool STAGED = decltype(stagedc)::value != 0;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16_t* erow = p.e + (int64_t)(rowg[mt] / p.T) * p.ld_e + (int64_t)i * H;
        const unsigned char* el = eimg + ewl[mt] * (H * 2);
#pragma unroll
        for (int u = 0; u < NTH; ++u) {
          const int col = colb + 16 * u;
          bf16x4_t e4;
          if constexpr (STAGED) e4 = *reinterpret_cast<const bf16x4_t*>(el + col * 2);
          else e4 = *reinterpret_cast<const bf16x4_t*>(erow + col);
          const float bb[4] = {b4[u].x, b4[u].y, b4[u].z, b4[u].w};
          bf16x4_t ub;
#pragma unroll
          for (int r = 0; r < 4; ++r) ub[r] = (bf16_t)(acc[mt][u][r] + bb[r] + (float)e4[r]);
          *reinterpret_cast<bf16x4_t*>(cur + (16 * mt + l16) * RS + col * 2) = ub;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (float)ub[r];
            const float v = x * fast_sigmoid(x);
            acc[mt][u][r] = v;
            s1[mt] += v;
            s2[mt] += v * v;
          }
        }
      }
    };
    if (estage) pass_u(IntC<1>{});
    else pass_u(IntC<0>{});
    CH_STAMP(3 + 4 * i);
    row_reduce2<4>(s1, s2, (i & 1) ? redB : redA, lane, wave, 0);
    CH_STAMP(4 + 4 * i);
#pragma unroll
    for (int j = 0; j < NPH; ++j)          // u image complete after the barrier inside the reduction
      out_piece16(cur, RS, p.u[i] + (int64_t)r0 * H, H, nrows, PPR, TIDV + j * CH_THREADS);
    float mean[4], rstd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      mean[mt] = s1[mt] * invH;
      rstd[mt] = __builtin_amdgcn_rsqf(fmaxf(s2[mt] * invH - mean[mt] * mean[mt], 0.f) + p.ln_eps);
    }
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        stats[(i * CH_ROWS + 16 * mt + lane) * 2 + 0] = mean[mt];
        stats[(i * CH_ROWS + 16 * mt + lane) * 2 + 1] = rstd[mt];
      }
    }
    // h = gamma * xhat + beta -> next A image (rows beyond the panel stay zero)
#pragma unroll
    for (int u = 0; u < NTH; ++u) {
      const int col = colb + 16 * u;
      const float4 g4 = *reinterpret_cast<const float4*>(p.gamma[i] + col);
      const float4 be4 = *reinterpret_cast<const float4*>(p.beta[i] + col);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float m = mean[mt], rsd = rstd[mt];
        bf16x4_t hb = pack4((acc[mt][u][0] - m) * rsd * g4.x + be4.x, (acc[mt][u][1] - m) * rsd * g4.y + be4.y,
                            (acc[mt][u][2] - m) * rsd * g4.z + be4.z, (acc[mt][u][3] - m) * rsd * g4.w + be4.w);
        if (!valid[mt]) hb = pack4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<bf16x4_t*>(nxt + (16 * mt + l16) * RS + col * 2) = hb;
      }
    }
    __syncthreads();
    CH_STAMP(5 + 4 * i);
    unsigned char* t = cur; cur = nxt; nxt = t;
  }

  // ---- head + loss + dL/dpred  (columns >= D come out as exact zeros: the packed head rows there are zero)
  {
    const int ppr = D >> 2;
    uint2 re[KBD];
    f32x4_t acc[4][NTD];
    zero_acc<NTD>(acc);
    bf16_t* hlast = p.h[p.L - 1] + (int64_t)r0 * H;
    const bf16_t* epsg = in_eps + (int64_t)r0 * D;
    auto side = [&](int kb, auto kbc) {
      constexpr int KB = decltype(kbc)::value;
#pragma unroll
      for (int j = 0; j < NPH; ++j)
        if (j % KB == kb) out_piece16(cur, RS, hlast, H, nrows, PPR, TIDV + j * CH_THREADS);
#pragma unroll
      for (int j = 0; j < KBD; ++j)
        if (j % KB == kb) re[j] = in_piece8(epsg, D, nrows, ppr, TIDV + j * CH_THREADS);
    };
    chain_gemm<NTD, C::KBH>(p.wf[p.L], wave_s * NTD, cur, RS, lane_id_now(), acc, side);
    CH_STAMP(2 + 4 * p.L);
    store_rows8<KBD>(nxt, RS, ppr, TIDV, re);     // eps image; dpred overwrites it in place
    __syncthreads();
    const int colb = wave * 16 * NTD + 4 * g;
    float lsum = 0.f;
    float* prow = p.partial + (int64_t)blockIdx.x * p.ld_part;
#pragma unroll
    for (int u = 0; u < NTD; ++u) {
      const int col = colb + 16 * u;
      const bool cin = col < D;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cin) b4 = *reinterpret_cast<const float4*>(p.bias[p.L] + col);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4_t* slot = reinterpret_cast<bf16x4_t*>(nxt + (16 * mt + l16) * RS + col * 2);
        bf16x4_t dp = pack4(0.f, 0.f, 0.f, 0.f);
        if (cin && valid[mt]) {
          const bf16x4_t e4 = *slot;
          float d[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = (float)(bf16_t)(acc[mt][u][r] + bb[r]);
            d[r] = pr - (float)e4[r];
            lsum += d[r] * d[r];
          }
          dp = pack4(d[0] * p.gscale, d[1] * p.gscale, d[2] * p.gscale, d[3] * p.gscale);
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[r] += (float)dp[r];
        }
        *slot = dp;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[r] = row16_sum(cs[r]);          // head-bias gradient: this panel's column sums
      if (l16 == 0) *reinterpret_cast<float4*>(prow + 3 * p.L * H + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
    lsum = ib_wave_sum(lsum);
    if (lane == 0) lossred[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < CH_WAVES; ++w) s += lossred[w];
      prow[3 * p.L * H + C::DN] = s;
    }
    CH_STAMP(3 + 4 * p.L);
    unsigned char* t = cur; cur = nxt; nxt = t;
  }

  // ---- backward: dh_i = d(out of block i) ; through LayerNorm and SiLU -> dz_i ; dh_{i-1} = dz_i W_i.
  // `cur` = the GEMM input image (dpred, then dz_{i+1}); after the GEMM it receives dz_i.  `nxt` = the u_i image.
  const bool dp16 = (p.ld_dpred & 7) == 0 && (reinterpret_cast<uintptr_t>(p.dpred) & 15) == 0;
  for (int i = p.L - 1; i >= 0; --i) {
    uint4 ru[NPH];
    f32x4_t acc[4][NTH];
    zero_acc<NTH>(acc);
    const bf16_t* usrc = p.u[i] + (int64_t)r0 * H;
    if (i == p.L - 1) {
      bf16_t* dpg = p.dpred + (int64_t)r0 * p.ld_dpred;
      const int ppd = dp16 ? (D + 7) >> 3 : D >> 2;      // 16-byte pieces (pad columns receive the image's zeros) or 8
      // [64, DP] image: DP/8 16-byte (or DP/4 8-byte) pieces per row cover every ld_dpred <= DP; piece columns beyond
      // the row pitch are clamped onto the row's last piece
      if (dp16) {
        auto side = [&](int kb, auto kbc) {
          constexpr int KB = decltype(kbc)::value;
#pragma unroll
          for (int j = 0; j < NPH; ++j)
            if (j % KB == kb) ru[j] = in_piece16(usrc, H, nrows, PPR, TIDV + j * CH_THREADS);
#pragma unroll
          for (int j = 0; j < (C::DP / 8 * CH_ROWS + CH_THREADS - 1) / CH_THREADS; ++j)
            if (j % KB == kb) {
              const int idx = TIDV + j * CH_THREADS;
              const int row = min(idx / (C::DP / 8), nrows - 1), pc = min(idx % (C::DP / 8), ppd - 1);
              const uint4 v = *reinterpret_cast<const uint4*>(cur + row * RS + pc * 16);
              *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(dpg + (int64_t)row * p.ld_dpred) + pc * 16) = v;
            }
        };
        chain_gemm<NTH, KBD>(p.wb[p.L], wave_s * NTH, cur, RS, lane_id_now(), acc, side);
      } else {
        auto side = [&](int kb, auto kbc) {
          constexpr int KB = decltype(kbc)::value;
#pragma unroll
          for (int j = 0; j < NPH; ++j)
            if (j % KB == kb) ru[j] = in_piece16(usrc, H, nrows, PPR, TIDV + j * CH_THREADS);
#pragma unroll
          for (int j = 0; j < (C::DP / 4 * CH_ROWS + CH_THREADS - 1) / CH_THREADS; ++j)
            if (j % KB == kb) {
              const int idx = TIDV + j * CH_THREADS;
              const int row = min(idx / (C::DP / 4), nrows - 1), pc = min(idx % (C::DP / 4), ppd - 1);
              const uint2 v = *reinterpret_cast<const uint2*>(cur + row * RS + pc * 8);
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(dpg + (int64_t)row * p.ld_dpred) + pc * 8) = v;
            }
        };
        chain_gemm<NTH, KBD>(p.wb[p.L], wave_s * NTH, cur, RS, lane_id_now(), acc, side);
      }
    } else {
      bf16_t* dzg = p.dz[i + 1] + (int64_t)r0 * H;
      auto side = [&](int kb, auto kbc) {
        constexpr int KB = decltype(kbc)::value;
#pragma unroll
        for (int j = 0; j < NPH; ++j)
          if (j % KB == kb) {
            ru[j] = in_piece16(usrc, H, nrows, PPR, TIDV + j * CH_THREADS);
            out_piece16(cur, RS, dzg, H, nrows, PPR, TIDV + j * CH_THREADS);
          }
      };
      chain_gemm<NTH, C::KBH>(p.wb[i + 1], wave_s * NTH, cur, RS, lane_id_now(), acc, side);
    }
    CH_STAMP(4 + 4 * p.L + 9 * (p.L - 1 - i));
    store_rows16<NPH>(nxt, RS, PPR, TIDV, ru);
    __syncthreads();
    CH_STAMP(5 + 4 * p.L + 9 * (p.L - 1 - i));
    const int colb = wave * 16 * NTH + 4 * g;
    float mean[4], rstd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float2 st = *reinterpret_cast<const float2*>(stats + (i * CH_ROWS + 16 * mt + l16) * 2);
      mean[mt] = st.x; rstd[mt] = st.y;
    }
    // Two m-tile halves (rows are independent; halving keeps xhat / silu' for only 32 elements per lane live):
    //   pass 1: xhat, silu'(u); dgamma / dbeta partials; row sums of dxhat and dxhat * xhat
    //   pass 2: dv = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) ; dz = dv * silu'(u)
    float dg[NTH][4], db[NTH][4];
#pragma unroll
    for (int u = 0; u < NTH; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { dg[u][r] = 0.f; db[u][r] = 0.f; }
#pragma unroll
    for (int hf = 0; hf < NHF; ++hf) {
      bf16x4_t xh[MPH][NTH];             // xhat, bf16 (its only later use multiplies it into the bf16 dz)
      bf16x4_t dsl[MPH][NTH];            // silu'(u), bf16
      float sa[MPH], sb[MPH];
#pragma unroll
      for (int m2 = 0; m2 < MPH; ++m2) { sa[m2] = 0.f; sb[m2] = 0.f; }
#pragma unroll
      for (int u = 0; u < NTH; ++u) {
        const int col = colb + 16 * u;
        const float4 g4 = *reinterpret_cast<const float4*>(p.gamma[i] + col);
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int m2 = 0; m2 < MPH; ++m2) {
          const int mt = MPH * hf + m2;
          const bf16x4_t ub = *reinterpret_cast<const bf16x4_t*>(nxt + (16 * mt + l16) * RS + col * 2);
          float ds[4], xq[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (float)ub[r];                  // rows beyond the panel hold a finite duplicate row
            const float sg = fast_sigmoid(x);
            const float v = x * sg;
            ds[r] = sg * (1.f + x * (1.f - sg));
            const float xhat = (v - mean[mt]) * rstd[mt];
            const float dh = valid[mt] ? acc[mt][u][r] : 0.f;
            dg[u][r] += dh * xhat;
            db[u][r] += dh;
            const float dxh = dh * gg[r];
            acc[mt][u][r] = dxh;
            xq[r] = xhat;
            sa[m2] += dxh;
            sb[m2] += dxh * xhat;
          }
          dsl[m2][u] = pack4(ds[0], ds[1], ds[2], ds[3]);
          xh[m2][u] = pack4(xq[0], xq[1], xq[2], xq[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);    // keep the dgamma / dbeta FMAs inside their pass (hipcc deferred them with their operands spilled)
      CH_STAMP(6 + 4 * p.L + 9 * (p.L - 1 - i) + 2 * hf);
      row_reduce2<MPH>(sa, sb, hf ? redA : redB, lane, wave, MPH * hf);
      CH_STAMP(7 + 4 * p.L + 9 * (p.L - 1 - i) + 2 * hf);
#pragma unroll
      for (int m2 = 0; m2 < MPH; ++m2) {
        const int mt = MPH * hf + m2;
        const float ma = sa[m2] * invH, mb = sb[m2] * invH, rsd = rstd[mt];
#pragma unroll
        for (int u = 0; u < NTH; ++u) {
          const int col = colb + 16 * u;
          float dzv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dzv[r] = rsd * (acc[mt][u][r] - ma - (float)xh[m2][u][r] * mb) * (float)dsl[m2][u][r];
          bf16x4_t o = pack4(dzv[0], dzv[1], dzv[2], dzv[3]);
          if (!valid[mt]) o = pack4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<bf16x4_t*>(cur + (16 * mt + l16) * RS + col * 2) = o;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    CH_STAMP(10 + 4 * p.L + 9 * (p.L - 1 - i));
    {
      // per-workgroup column sums dgamma | dbeta.  Scratch = the u image (`nxt`): every wave finished reading it before
      // the barrier inside the second reduction above.
      float* scr = reinterpret_cast<float*>(nxt) + wave * 16 * C::CSS;
      float* pg = p.partial + (int64_t)blockIdx.x * p.ld_part + 3 * i * H;
      const int col = wave * 16 * NTH + lane;
      const float sg_ = wave_colsum<NTH>(dg, scr, C::CSS, lane);
      const float sb_ = wave_colsum<NTH>(db, scr, C::CSS, lane);
      if (lane < 16 * NTH) {
        pg[col] = sg_;
        pg[H + col] = sb_;
      }
    }
    CH_STAMP(11 + 4 * p.L + 9 * (p.L - 1 - i));
    __syncthreads();
    {
      // dbias = column sums of the bf16 dz image (rows beyond the panel are zero), one column per thread, fixed order.
      // panel == one window: they ARE the time-embedding gradient row of that window.
      if (tid < H) {
        const unsigned char* cp = cur + tid * 2;
        float sz = 0.f;
#pragma unroll 16
        for (int r = 0; r < CH_ROWS; ++r) sz += (float)*reinterpret_cast<const bf16_t*>(cp + r * RS);
        p.partial[(int64_t)blockIdx.x * p.ld_part + 3 * i * H + 2 * H + tid] = sz;
        if (p.de_lp) p.de_lp[(int64_t)blockIdx.x * p.ld_de + i * H + tid] = (bf16_t)sz;
      }
    }
    if (i == 0) {
#pragma unroll
      for (int j = 0; j < NPH; ++j) out_piece16(cur, RS, p.dz[0] + (int64_t)r0 * H, H, nrows, PPR, TIDV + j * CH_THREADS);
    }
    CH_STAMP(12 + 4 * p.L + 9 * (p.L - 1 - i));
  }
}

#undef TIDV

// =====================================================================================================================
// v2 (round 3): the same chain with ROW-WISE epilogues.
//
// What the phase stamps of v1 said: the five GEMM phases take 24 of 74 us; the rest is epilogue time in which each wave
// owns 64 COLUMNS of all 64 rows -- every LayerNorm statistic is then a cross-wave exchange behind a workgroup barrier
// (the barrier waits for the SIMD's younger wave: 2-3 us each), every HBM row copy is a separate "piece mover" pass over
// the LDS image with runtime index arithmetic (divisions by the row width), rows 50..63 of a 50-token panel are computed
// and stored like real ones, and the pre-activation u makes an HBM round trip (26 MB written, 26 MB read per launch)
// only to be re-read by the workgroup that wrote it.
//
// v2 keeps the GEMM phases (a wave owns output columns, weights stream L2 -> VGPR) and turns the tile through LDS once per
// layer so that the epilogue owns whole ROWS: wave w takes rows w, w + 8, ... and a lane 8 consecutive columns of the row
// (H = 512: 64 lanes x 8 columns = one row per wave-instruction).
//   forward:  GEMM -> column owners write u = bf16(z + bias + e) into the second image -> barrier -> per row: SiLU, the two
//             row sums by a 64-lane butterfly (no LDS, no barrier), normalise, h row -> first image (next GEMM's input) and
//             -> HBM as one coalesced 1-KiB store from registers.  u stays IN REGISTERS (8 rows x 16 bytes per lane and
//             layer) until the backward of the same layer -- no HBM round trip, no LDS image (KEEP: up to 2 blocks).
//   backward: GEMM -> column owners write dh (fp32, two halves of 32 rows: 66 KB) -> barrier -> per row: LayerNorm / SiLU
//             backward entirely in registers, dz row -> image + HBM; dgamma / dbeta / dbias accumulate per lane over the
//             wave's rows and are summed across waves once per layer, in wave order (deterministic).
// Rows beyond the panel are skipped (wave-uniform test), not computed and clamped.  Same parameters, same outputs, same
// partial-sum layout as v1; u[i] may be NULL (then it is not stored at all) when the blocks fit the register plan.
// Sum of a per-lane value over the LPR lanes that hold one row (LPR = 16, 32, 64); every lane of the group gets the total.
// DPP only (ds_bpermute / __shfl_xor cross the LDS crossbar: ~100+ cycles of latency per step on the row's critical path):
// the 16 lanes of a DPP row first, then row_bcast15 (lane 15 of rows 0 / 2 added into rows 1 / 3), row_bcast31 (lane 31
// into rows 2 / 3); the total ends in the group's last lane and comes back through a scalar register.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_bcast_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float a, int lane) {
  a = row16_sum(a);
  if constexpr (LPR >= 32) a = dpp_bcast_add<0x142, 0xA>(a);
  if constexpr (LPR == 64) {
    a = dpp_bcast_add<0x143, 0xC>(a);
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
  } else if constexpr (LPR == 32) {
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    a = lane < 32 ? lo : hi;
  }
  return a;
}
__device__ __forceinline__ void unpack8(const uint4& q, float (&x)[8]) {
  const bf16x8_t v = __builtin_bit_cast(bf16x8_t, q);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
}
__device__ __forceinline__ uint4 pack8(const float (&x)[8]) {
  bf16x8_t o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (bf16_t)x[k];
  return __builtin_bit_cast(uint4, o);
}
__device__ __forceinline__ void load8f(const float* p, float (&x)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

template <int NTH, int NTD, int KBD, bool KEEP>
__global__ __launch_bounds__(CH_THREADS) void mlp_chain2_kernel(ChainParams p) {
  using C = ChainCfg<NTH, NTD, KBD>;
  constexpr int H = C::H, RS = C::RS, PPR = H / 8;
  constexpr int LPR = H / 8;                       // lanes per row in the row-wise passes (8 columns per lane)
  constexpr int RPW = 64 / LPR;                    // rows per wave-instruction (1 at H = 512)
  constexpr int RSTEP = CH_WAVES * RPW;            // row stride between a lane's successive rows
  constexpr int NJ = CH_ROWS / RSTEP, NJH = NJ / 2;
  constexpr int RSX = H * 4 + 16;                  // fp32 exchange image: 32 rows
  static_assert(32 * RSX <= C::BUF, "fp32 half-panel exchange must fit one image buffer");
  static_assert(3 * CH_WAVES * H * 4 <= C::BUF, "column-sum exchange must fit one image buffer");
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  unsigned char* imgA = smem;                       // GEMM input of the forward; fp32 exchange / column sums in the backward
  unsigned char* imgB = smem + C::BUF;              // u exchange (forward), eps / dpred, GEMM input of the backward
  float2* coef = reinterpret_cast<float2*>(smem + 2 * C::BUF);
  float* stats = reinterpret_cast<float*>(coef + 2 * CH_ROWS * CH_WAVES);   // [L][64][2] mean, rstd
  unsigned char* eimg = reinterpret_cast<unsigned char*>(stats + CH_MAXL * CH_ROWS * 2);   // [CH_NWIN][H] bf16
  float* lossred = reinterpret_cast<float*>(eimg + C::EIMG);            // [8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#define TIDV ((wave_s << 6) | lane_id_now())
  const int g = lane >> 4, l16 = lane & 15;
  const int r0 = blockIdx.x * p.P;
  const int D = p.D, M = p.M;
  const bf16_t* in_x0 = p.slots ? reinterpret_cast<const bf16_t*>(p.slots[0]) : p.x0;
  const bf16_t* in_eps = p.slots ? reinterpret_cast<const bf16_t*>(p.slots[1]) : p.eps;
  const int64_t* in_t = p.slots ? reinterpret_cast<const int64_t*>(p.slots[2]) : p.t;
  const int nrows = min(p.P, M - r0);
  const float invH = 1.f / (float)H;
  // row-wise ownership
  const int sub = lane / LPR, c8 = (lane % LPR) * 8;
  const int wrow = wave_s * RPW + sub;              // this lane's row of iteration 0 (wave-uniform at H = 512)
  CH_STAMP(0);

  int rowg[4]; bool valid[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int lr = 16 * mt + l16;
    valid[mt] = lr < nrows;
    rowg[mt] = min(r0 + lr, M - 1);
  }
  const int w0 = r0 / p.T;
  const int nwin = (r0 + nrows - 1) / p.T - w0 + 1;
  const bool estage = nwin <= CH_NWIN;
  int ewl[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ewl[mt] = min(rowg[mt] / p.T - w0, CH_NWIN - 1);

  // ---- q_sample, row-wise: wave w takes rows w, w + 8, ...; a row is D / 4 pieces of 8 bytes (<= 128: two per lane).
  // xt -> HBM and -> image A, whose columns D .. DP-1 are zeroed (the packed weights there are zero, but 0 x garbage is not)
  {
    const int ppr = D >> 2;
    uint2 rx[CH_ROWS / CH_WAVES][2], re[CH_ROWS / CH_WAVES][2];
#pragma unroll
    for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
      const int rc = min(wave_s + CH_WAVES * j, nrows - 1);
      const bf16_t* xr = in_x0 + (int64_t)(r0 + rc) * D;
      const bf16_t* er_ = in_eps + (int64_t)(r0 + rc) * D;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pc = min(lane + 64 * q, ppr - 1);
        rx[j][q] = *reinterpret_cast<const uint2*>(xr + pc * 4);
        re[j][q] = *reinterpret_cast<const uint2*>(er_ + pc * 4);
      }
    }
    // the per-row coefficients (three dependent loads: timestep, two table entries) BEHIND the row loads: the wave that
    // fetches them would otherwise stall on them before it has requested its rows
    if (tid < CH_ROWS) {
      float2 c = make_float2(0.f, 0.f);
      if (tid < nrows) {
        int64_t k = in_t[(r0 + tid) / p.T];
        k = k < 0 ? 0 : (k >= p.table_rows ? p.table_rows - 1 : k);
        c = make_float2(p.sqrt_ab[k], p.sqrt_1mab[k]);
      }
      coef[tid] = c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
      const int r = wave_s + CH_WAVES * j;
      if (r < nrows) {
        const float2 c = coef[r];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int pc = lane + 64 * q;
          const bf16x4_t xv = __builtin_bit_cast(bf16x4_t, rx[j][q]), ev = __builtin_bit_cast(bf16x4_t, re[j][q]);
          bf16x4_t o = pack4(c.x * (float)xv[0] + c.y * (float)ev[0], c.x * (float)xv[1] + c.y * (float)ev[1],
                             c.x * (float)xv[2] + c.y * (float)ev[2], c.x * (float)xv[3] + c.y * (float)ev[3]);
          if (pc < ppr) *reinterpret_cast<bf16x4_t*>(p.xt + (int64_t)(r0 + r) * p.ld_xt + pc * 4) = o;
          else o = pack4(0.f, 0.f, 0.f, 0.f);
          if (pc < C::DP / 4) *reinterpret_cast<bf16x4_t*>(imgA + r * RS + pc * 8) = o;
        }
      }
    }
  }
  __syncthreads();
  CH_STAMP(1);

  uint4 ukeep[KEEP ? 2 : 1][NJ];

  // ---- forward block i (I = i as a compile-time index of the kept registers)
  auto fwd_layer = [&](auto ic, const int i) {
    constexpr int I = decltype(ic)::value;
    f32x4_t acc[4][NTH];
    zero_acc<NTH>(acc);
    uint4 er = make_uint4(0u, 0u, 0u, 0u);
    const bf16_t* esrc = p.e + (int64_t)w0 * p.ld_e + (int64_t)i * H;
    const int colb = wave * 16 * NTH + 4 * g;
    // The panel's time-embedding rows.  One window per panel (the headline shape): every row adds the SAME e row, each
    // lane fetches its 4 x NTH columns of it straight into registers -- no LDS staging, no barrier.  Several windows: the
    // rows are staged in LDS (e image).  Both requests are issued unconditionally BEHIND the first weight loads (a branch
    // around a load makes hipcc drain the prefetch ring at the join); which one is used is a wave-uniform choice afterwards.
    bf16x4_t e1[NTH];
    float4 b4[NTH];
    float gm[8], bt[8];                    // LayerNorm gain / bias of this lane's 8 columns (row-wise pass)
    // (requested late in the k-loop: 44 registers that would otherwise be live through the whole GEMM phase, on top of the
    // accumulators, the prefetch ring and the kept pre-activations -- the first version of this spilled)
    auto side = [&](int kb, auto kbc) {
      constexpr int KB = decltype(kbc)::value;
      if (kb == KB - 1) {                  // the ring's other slots are dead by now
        load8f(p.gamma[i] + c8, gm);
        load8f(p.beta[i] + c8, bt);
      }
      if (kb == KB - 1) {
        er = in_piece16(esrc, p.ld_e, min(nwin, CH_NWIN), PPR, TIDV);
#pragma unroll
        for (int u = 0; u < NTH; ++u) {
          e1[u] = *reinterpret_cast<const bf16x4_t*>(esrc + colb + 16 * u);
          b4[u] = *reinterpret_cast<const float4*>(p.bias[i] + colb + 16 * u);
        }
      }
    };
    if (i == 0) chain_gemm<NTH, KBD>(p.wf[0], wave_s * NTH, imgA, RS, lane_id_now(), acc, side);
    else chain_gemm<NTH, C::KBH>(p.wf[i], wave_s * NTH, imgA, RS, lane_id_now(), acc, side);
    CH_STAMP(2 + 3 * i);
    if (nwin == 1) {
      // column owners: u = bf16(z + bias + e) -> image B
#pragma unroll
      for (int u = 0; u < NTH; ++u) {
        const float ee[4] = {(float)e1[u][0], (float)e1[u][1], (float)e1[u][2], (float)e1[u][3]};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)       // (z + bias) + e: the same order as the staged path
          *reinterpret_cast<bf16x4_t*>(imgB + (16 * mt + l16) * RS + (colb + 16 * u) * 2) =
              pack4(acc[mt][u][0] + b4[u].x + ee[0], acc[mt][u][1] + b4[u].y + ee[1], acc[mt][u][2] + b4[u].z + ee[2],
                    acc[mt][u][3] + b4[u].w + ee[3]);
      }
    } else {
      if (tid < CH_NWIN * PPR) *reinterpret_cast<uint4*>(eimg + tid * 16) = er;
      __syncthreads();                     // e image complete
      auto pre = [&](auto stagedc) {
        constexpr bool STAGED = decltype(stagedc)::value != 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const bf16_t* erow = p.e + (int64_t)(rowg[mt] / p.T) * p.ld_e + (int64_t)i * H;
          const unsigned char* el = eimg + ewl[mt] * (H * 2);
#pragma unroll
          for (int u = 0; u < NTH; ++u) {
            const int col = colb + 16 * u;
            bf16x4_t e4;
            if constexpr (STAGED) e4 = *reinterpret_cast<const bf16x4_t*>(el + col * 2);
            else e4 = *reinterpret_cast<const bf16x4_t*>(erow + col);
            *reinterpret_cast<bf16x4_t*>(imgB + (16 * mt + l16) * RS + col * 2) =
                pack4(acc[mt][u][0] + b4[u].x + (float)e4[0], acc[mt][u][1] + b4[u].y + (float)e4[1],
                      acc[mt][u][2] + b4[u].z + (float)e4[2], acc[mt][u][3] + b4[u].w + (float)e4[3]);
          }
        }
      };
      if (estage) pre(IntC<1>{});
      else pre(IntC<0>{});
    }
    __syncthreads();                       // u image complete; every wave is done reading image A
    CH_STAMP(3 + 3 * i);
    {
      bf16_t* ug = p.u[i];
      bf16_t* hg = p.h[i] + (int64_t)r0 * H;
      // G rows at a time, stage by stage (the rows of a group are independent: their latency chains -- LDS read, exp / rcp,
      // the row reductions -- overlap); a group whose first row lies beyond the panel ends the loop (wave-uniform at
      // H = 512; a lane's rows ascend).  Stores of single rows beyond the panel are predicated.
      constexpr int G = 2;
#pragma unroll
      for (int j0 = 0; j0 < NJ; j0 += G) {
        if (RPW == 1 && wave_s + RSTEP * j0 >= nrows) break;
        uint4 q[G];
        float v[G][8], s1[G], s2[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int r = wrow + RSTEP * (j0 + gi);
          q[gi] = *reinterpret_cast<const uint4*>(imgB + r * RS + c8 * 2);
          if constexpr (KEEP) ukeep[I][j0 + gi] = q[gi];
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          float x[8];
          unpack8(q[gi], x);
          // pairwise partial sums (even / odd columns): the compiler packs them into v_pk_add / v_pk_fma
          float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            v[gi][k] = x[k] * fast_sigmoid(x[k]);
            v[gi][k + 1] = x[k + 1] * fast_sigmoid(x[k + 1]);
            a0 += v[gi][k]; a1 += v[gi][k + 1];
            q0 += v[gi][k] * v[gi][k]; q1 += v[gi][k + 1] * v[gi][k + 1];
          }
          s1[gi] = a0 + a1; s2[gi] = q0 + q1;
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          s1[gi] = group_sum<LPR>(s1[gi], lane);
          s2[gi] = group_sum<LPR>(s2[gi], lane);
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int r = wrow + RSTEP * (j0 + gi);
          const bool ok = r < nrows;
          const float mean = s1[gi] * invH;
          const float rstd = __builtin_amdgcn_rsqf(fmaxf(s2[gi] * invH - mean * mean, 0.f) + p.ln_eps);
          float hv[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) hv[k] = (v[gi][k] - mean) * rstd * gm[k] + bt[k];
          const uint4 hq = pack8(hv);
          if (ok) {
            if (ug != nullptr) *reinterpret_cast<uint4*>(ug + (int64_t)(r0 + r) * H + c8) = q[gi];
            *reinterpret_cast<uint4*>(imgA + r * RS + c8 * 2) = hq;
            *reinterpret_cast<uint4*>(hg + (int64_t)r * H + c8) = hq;
            if (c8 == 0) *reinterpret_cast<float2*>(stats + (i * CH_ROWS + r) * 2) = make_float2(mean, rstd);
          }
        }
      }
    }
    __syncthreads();                       // image A = h_i complete
    CH_STAMP(4 + 3 * i);
  };
  if constexpr (KEEP) {
    fwd_layer(IntC<0>{}, 0);
    if (p.L > 1) fwd_layer(IntC<1>{}, 1);
  } else {
    for (int i = 0; i < p.L; ++i) fwd_layer(IntC<0>{}, i);
  }

  // ---- head + loss + dL/dpred (column owners; eps rows staged row-wise into image B, dpred written over them)
  const bool dp16 = (p.ld_dpred & 7) == 0 && (reinterpret_cast<uintptr_t>(p.dpred) & 15) == 0;
  {
    const int ppr = D >> 2;
    uint2 re[CH_ROWS / CH_WAVES][2];
    f32x4_t acc[4][NTD];
    zero_acc<NTD>(acc);
    const int colb = wave * 16 * NTD + 4 * g;
    // the head bias, requested during the GEMM's last k-block (requested in the epilogue it was a cold round trip with
    // nothing to hide behind -- every phase of this kernel starts where the previous one's last load returned)
    float4 hb4[NTD];
    auto side = [&](int kb, auto kbc) {
      constexpr int KB = decltype(kbc)::value;
#if IB_CHAIN_EPS_LATE
      // the panel's eps rows (the loss target), requested BEHIND the weight stream, four k-blocks before the GEMM ends:
      // ahead of it (vmcnt retires in order) the first weight k-blocks waited for these 16 row pieces' round trip
      if (kb == KB - 4) {
#pragma unroll
        for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
          const int rc = min(wave_s + CH_WAVES * j, nrows - 1);
          const bf16_t* er_ = in_eps + (int64_t)(r0 + rc) * D;
#pragma unroll
          for (int q = 0; q < 2; ++q) re[j][q] = *reinterpret_cast<const uint2*>(er_ + min(lane_id_now() + 64 * q, ppr - 1) * 4);
        }
      }
#endif
      if (kb == KB - 1) {
#pragma unroll
        for (int u = 0; u < NTD; ++u) {
          const int col = colb + 16 * u;
          hb4[u] = col < D ? *reinterpret_cast<const float4*>(p.bias[p.L] + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
#if !IB_CHAIN_EPS_LATE
#pragma unroll
    for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
      const int rc = min(wave_s + CH_WAVES * j, nrows - 1);
      const bf16_t* er_ = in_eps + (int64_t)(r0 + rc) * D;
#pragma unroll
      for (int q = 0; q < 2; ++q) re[j][q] = *reinterpret_cast<const uint2*>(er_ + min(lane + 64 * q, ppr - 1) * 4);
    }
#endif
    chain_gemm<NTD, C::KBH>(p.wf[p.L], wave_s * NTD, imgA, RS, lane_id_now(), acc, side);
    CH_STAMP(2 + 3 * p.L);
#pragma unroll
    for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
      const int r = wave_s + CH_WAVES * j;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (lane + 64 * q < ppr) *reinterpret_cast<uint2*>(imgB + r * RS + (lane + 64 * q) * 8) = re[j][q];
    }
    __syncthreads();
    float lsum = 0.f;
    float* prow = p.partial + (int64_t)blockIdx.x * p.ld_part;
#pragma unroll
    for (int u = 0; u < NTD; ++u) {
      const int col = colb + 16 * u;
      const bool cin = col < D;
      const float bb[4] = {hb4[u].x, hb4[u].y, hb4[u].z, hb4[u].w};
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4_t* slot = reinterpret_cast<bf16x4_t*>(imgB + (16 * mt + l16) * RS + col * 2);
        bf16x4_t dp = pack4(0.f, 0.f, 0.f, 0.f);
        if (cin && valid[mt]) {
          const bf16x4_t e4 = *slot;
          float d[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = (float)(bf16_t)(acc[mt][u][r] + bb[r]);
            d[r] = pr - (float)e4[r];
            lsum += d[r] * d[r];
          }
          dp = pack4(d[0] * p.gscale, d[1] * p.gscale, d[2] * p.gscale, d[3] * p.gscale);
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[r] += (float)dp[r];
        }
        *slot = dp;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[r] = row16_sum(cs[r]);
      if (l16 == 0) *reinterpret_cast<float4*>(prow + 3 * p.L * H + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
#if IB_CHAIN_EPS_LATE
    lsum = group_sum<64>(lsum, lane);      // DPP only (six ds_bpermute round trips through the LDS crossbar before)
#else
    lsum = ib_wave_sum(lsum);
#endif
    if (lane == 0) lossred[wave] = lsum;
    __syncthreads();                       // dpred image complete
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < CH_WAVES; ++w) s += lossred[w];
      prow[3 * p.L * H + C::DN] = s;
    }
    // dpred rows -> HBM, row-wise (pad columns of a 16-byte row pitch receive the image's zeros)
    bf16_t* dpg = p.dpred + (int64_t)r0 * p.ld_dpred;
#pragma unroll
    for (int j = 0; j < CH_ROWS / CH_WAVES; ++j) {
      const int r = wave_s + CH_WAVES * j;
      if (r < nrows) {
        if (dp16) {
          if (lane < ((D + 7) >> 3))
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(dpg + (int64_t)r * p.ld_dpred) + lane * 16) =
                *reinterpret_cast<const uint4*>(imgB + r * RS + lane * 16);
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (lane + 64 * q < ppr)
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(dpg + (int64_t)r * p.ld_dpred) + (lane + 64 * q) * 8) =
                  *reinterpret_cast<const uint2*>(imgB + r * RS + (lane + 64 * q) * 8);
        }
      }
    }
    CH_STAMP(3 + 3 * p.L);
  }

  // ---- backward of block i: dh = (dz_{i+1} or dpred) . W^T (GEMM input = image B) -> LayerNorm / SiLU backward -> dz_i
  auto bwd_layer = [&](auto ic, const int i) {
    constexpr int I = decltype(ic)::value;
    const int sb0 = 4 + 3 * p.L + 6 * (p.L - 1 - i);
    uint4 ur[NJ];
    if constexpr (KEEP) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ur[j] = ukeep[I][j];
    } else {
      const bf16_t* ug = p.u[i] + (int64_t)r0 * H;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        ur[j] = *reinterpret_cast<const uint4*>(ug + (int64_t)min(wrow + RSTEP * j, nrows - 1) * H + c8);
    }
    f32x4_t acc[4][NTH];
    zero_acc<NTH>(acc);
    float gm[8];
    auto side = [&](int kb, auto kbc) {
      constexpr int KB = decltype(kbc)::value;
      if (kb == KB - 1) load8f(p.gamma[i] + c8, gm);
    };
    if (i == p.L - 1) chain_gemm<NTH, KBD>(p.wb[p.L], wave_s * NTH, imgB, RS, lane_id_now(), acc, side);
    else chain_gemm<NTH, C::KBH>(p.wb[i + 1], wave_s * NTH, imgB, RS, lane_id_now(), acc, side);
    CH_STAMP(sb0);
    float dgam[8], dbet[8], dbs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { dgam[k] = 0.f; dbet[k] = 0.f; dbs[k] = 0.f; }
    bf16_t* dzg = p.dz[i] + (int64_t)r0 * H;
    const int colb = wave * 16 * NTH + 4 * g;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // column owners: dh of rows 32 half .. 32 half + 31 -> fp32 exchange image (image A's storage)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int u = 0; u < NTH; ++u)
          *reinterpret_cast<f32x4_t*>(imgA + (16 * m2 + l16) * RSX + (colb + 16 * u) * 4) = acc[2 * half + m2][u];
      __syncthreads();
      CH_STAMP(sb0 + 1 + 2 * half);
      constexpr int G = NJH >= 2 ? 2 : 1;
#pragma unroll
      for (int jj0 = 0; jj0 < NJH; jj0 += G) {
        if (RPW == 1 && wave_s + RSTEP * (half * NJH + jj0) >= nrows) break;
        float dxh[G][8], xh[G][8], ds[G][8], sa[G], sb[G];
        float2 st[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int j = half * NJH + jj0 + gi;
          const int r = wrow + RSTEP * j;
          const bool ok = r < nrows;
          float dh[8], x[8];
          load8f(reinterpret_cast<const float*>(imgA + (r - 32 * half) * RSX) + c8, dh);
          unpack8(ur[j], x);
          st[gi] = *reinterpret_cast<const float2*>(stats + (i * CH_ROWS + min(r, nrows - 1)) * 2);
          sa[gi] = 0.f; sb[gi] = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xx = ok ? x[k] : 0.f;          // rows beyond the panel hold garbage: keep the sums finite
            const float sg = fast_sigmoid(xx);
            const float v = xx * sg;
            ds[gi][k] = sg * (1.f + xx * (1.f - sg));
            xh[gi][k] = (v - st[gi].x) * st[gi].y;
            const float d = ok ? dh[k] : 0.f;
            dgam[k] += d * xh[gi][k];
            dbet[k] += d;
            dxh[gi][k] = d * gm[k];
            sa[gi] += dxh[gi][k];
            sb[gi] += dxh[gi][k] * xh[gi][k];
          }
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          sa[gi] = group_sum<LPR>(sa[gi], lane);
          sb[gi] = group_sum<LPR>(sb[gi], lane);
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int r = wrow + RSTEP * (half * NJH + jj0 + gi);
          const bool ok = r < nrows;
          const float ma = sa[gi] * invH, mb = sb[gi] * invH;
          float dzv[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) dzv[k] = st[gi].y * (dxh[gi][k] - ma - xh[gi][k] * mb) * ds[gi][k];
          const uint4 zq = pack8(dzv);
          if (ok) {
            float zr[8];
            unpack8(zq, zr);
#pragma unroll
            for (int k = 0; k < 8; ++k) dbs[k] += zr[k];
            *reinterpret_cast<uint4*>(imgB + r * RS + c8 * 2) = zq;
            *reinterpret_cast<uint4*>(dzg + (int64_t)r * H + c8) = zq;
          }
        }
      }
      __syncthreads();                     // exchange image free again; after the second half image B = dz_i
      CH_STAMP(sb0 + 2 + 2 * half);
    }
    {
      // dgamma | dbeta | dbias of this panel: the waves' per-lane sums over their rows, added in row-group order
      if constexpr (RPW > 1) {             // several row groups per wave: add them first (fixed order)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if constexpr (LPR == 16) {
            dgam[k] += __shfl_xor(dgam[k], 16, 64); dbet[k] += __shfl_xor(dbet[k], 16, 64); dbs[k] += __shfl_xor(dbs[k], 16, 64);
          }
          dgam[k] += __shfl_xor(dgam[k], 32, 64); dbet[k] += __shfl_xor(dbet[k], 32, 64); dbs[k] += __shfl_xor(dbs[k], 32, 64);
        }
      }
      float* cr = reinterpret_cast<float*>(imgA);
      if (sub == 0) {
        float* c0 = cr + ((0 * CH_WAVES + wave) * H + c8);
        float* c1 = cr + ((1 * CH_WAVES + wave) * H + c8);
        float* c2 = cr + ((2 * CH_WAVES + wave) * H + c8);
        *reinterpret_cast<float4*>(c0) = make_float4(dgam[0], dgam[1], dgam[2], dgam[3]);
        *reinterpret_cast<float4*>(c0 + 4) = make_float4(dgam[4], dgam[5], dgam[6], dgam[7]);
        *reinterpret_cast<float4*>(c1) = make_float4(dbet[0], dbet[1], dbet[2], dbet[3]);
        *reinterpret_cast<float4*>(c1 + 4) = make_float4(dbet[4], dbet[5], dbet[6], dbet[7]);
        *reinterpret_cast<float4*>(c2) = make_float4(dbs[0], dbs[1], dbs[2], dbs[3]);
        *reinterpret_cast<float4*>(c2 + 4) = make_float4(dbs[4], dbs[5], dbs[6], dbs[7]);
      }
      __syncthreads();
      if (tid < H) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < CH_WAVES; ++w) {
          s0 += cr[(0 * CH_WAVES + w) * H + tid];
          s1 += cr[(1 * CH_WAVES + w) * H + tid];
          s2 += cr[(2 * CH_WAVES + w) * H + tid];
        }
        float* pg = p.partial + (int64_t)blockIdx.x * p.ld_part + 3 * i * H;
        pg[tid] = s0;
        pg[H + tid] = s1;
        pg[2 * H + tid] = s2;
        if (p.de_lp) p.de_lp[(int64_t)blockIdx.x * p.ld_de + i * H + tid] = (bf16_t)s2;
      }
      __syncthreads();                     // the next block's exchange writes reuse this storage
    }
    CH_STAMP(sb0 + 5);
  };
  if constexpr (KEEP) {
    if (p.L > 1) bwd_layer(IntC<1>{}, 1);
    bwd_layer(IntC<0>{}, 0);
  } else {
    for (int i = p.L - 1; i >= 0; --i) bwd_layer(IntC<0>{}, i);
  }
}

#undef TIDV

// ---- weight packing: bf16 row-major [N_out, K_in] -> fragment-major blocks, zero padded
struct PackDesc {
  const bf16_t* src; int64_t ld;
  int N, K;            // logical W_eff dims (rows n, reduction k)
  int transpose;       // W_eff[n][k] = src[k * ld + n] instead of src[n * ld + k]
  int n_tiles, KB;
  int64_t dst_off;     // in elements
  int block0;
};
struct PackParams { PackDesc d[2 * CH_MAXL + 1]; int count; int total_blocks; bf16_t* dst; };

// pack blocks first, first + stride, ... (one wave per 1-KiB block)
__device__ __forceinline__ void pack_blocks(const PackParams& p, int first, int stride, int lane) {
  for (int blk = first; blk < p.total_blocks; blk += stride) {
    int di = 0;
    for (int j = 1; j < p.count; ++j)
      if (blk >= p.d[j].block0) di = j;
    const PackDesc& d = p.d[di];
    const int local = blk - d.block0;
#if IB_CHAIN_KBMAJOR
    const int nt = local % d.n_tiles, kb = local / d.n_tiles;
#else
    const int nt = local / d.KB, kb = local % d.KB;
#endif
    const int n = 16 * nt + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
    bf16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      bf16_t x = (bf16_t)0.f;
      if (n < d.N && k < d.K) x = d.transpose ? d.src[(int64_t)k * d.ld + n] : d.src[(int64_t)n * d.ld + k];
      v[j] = x;
    }
    reinterpret_cast<bf16x8_t*>(p.dst + d.dst_off)[(int64_t)local * 64 + lane] = v;
  }
}
__global__ __launch_bounds__(256) void mlp_chain_pack_kernel(PackParams p) {
  pack_blocks(p, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63);
}

struct ChainShape { int nth, ntd, kbd; };
bool chain_shape(int64_t D, int64_t H, ChainShape* s) {
  if (H != 128 && H != 256 && H != 512) return false;
  if (D <= 0 || D % 4 != 0 || D > 512) return false;
  s->nth = (int)(H / 128);
  // feature-width geometries (n-tiles per wave, k-blocks), smallest first; the instantiated set per hidden width
  static const int geo[6][2] = {{1, 2}, {1, 4}, {2, 8}, {3, 10}, {3, 12}, {4, 16}};
  const int kb_need = (int)((D + 31) / 32), nt_need = (int)(((D + 15) / 16 + 7) / 8);
  for (int j = 0; j < 6; ++j) {
    if (geo[j][0] < nt_need || geo[j][1] < kb_need) continue;
    const bool have = H == 512 || (H == 256 && j == 3) || (H == 128 && j < 2);
    if (!have) continue;
    s->ntd = geo[j][0]; s->kbd = geo[j][1];
    return true;
  }
  return false;
}

// packed image layout (elements): forward blocks 0..L-1, head, then transposed blocks 1..L-1, head^T
struct PackLayout { int64_t off_f[CH_MAXL + 1]; int64_t off_b[CH_MAXL + 1]; int64_t total; };
void chain_layout(int64_t D, int64_t H, int L, const ChainShape& s, PackLayout* o) {
  const int64_t kbh = H / 32, nth8 = H / 16, ntd8 = 8 * s.ntd;
  int64_t off = 0;
  for (int i = 0; i < L; ++i) { o->off_f[i] = off; off += nth8 * (i == 0 ? s.kbd : kbh) * 512; }
  o->off_f[L] = off; off += ntd8 * kbh * 512;
  o->off_b[0] = -1;
  for (int i = 1; i < L; ++i) { o->off_b[i] = off; off += nth8 * kbh * 512; }
  o->off_b[L] = off; off += nth8 * s.kbd * 512;
  o->total = off;
}

}  // namespace

// TIMING-ONLY: device buffer of [workgroups][16] int64 stamps filled by the next chain launches (NULL = off)
#ifdef IB_AB
extern "C" int ib_debug_set_chain_prof(void* buf) { g_chain_prof = reinterpret_cast<long long*>(buf); return IB_OK; }
#else
extern "C" int ib_debug_set_chain_prof(void*) { return IB_E_UNSUPPORTED; }     // measurement builds only
#endif

extern "C" int ib_mlp_chain_supported(int64_t D, int64_t H, int L) {
  ChainShape s;
  return (L >= 1 && L <= CH_MAXL && chain_shape(D, H, &s)) ? 1 : 0;
}

extern "C" size_t ib_mlp_chain_packed_elems(int64_t D, int64_t H, int L) {
  ChainShape s;
  if (L < 1 || L > CH_MAXL || !chain_shape(D, H, &s)) return 0;
  PackLayout lo;
  chain_layout(D, H, L, s, &lo);
  return (size_t)lo.total;
}

extern "C" int ib_mlp_chain_workgroups(int64_t M, int* rows_per_wg) {
  // one workgroup per CU when the token count allows it (256 CUs), never more than 64 tokens per workgroup
  int64_t P = (M + 255) / 256;
  if (P > CH_ROWS) P = CH_ROWS;
  if (P < 16) P = M < 16 ? M : 16;
  if (rows_per_wg) *rows_per_wg = (int)P;
  return (int)((M + P - 1) / P);
}

namespace {
int build_pack(const void* const* w, const int64_t* ldw, void* packed, int64_t D, int64_t H, int L, PackParams& pp) {
  ChainShape s;
  if (!w || !ldw || !packed || L < 1 || L > CH_MAXL || !chain_shape(D, H, &s)) return IB_E_ARG;
  PackLayout lo;
  chain_layout(D, H, L, s, &lo);
  pp = PackParams{};
  pp.dst = reinterpret_cast<bf16_t*>(packed);
  int blocks = 0, c = 0;
  auto add = [&](const void* src, int64_t ld, int N, int K, int tr, int n_tiles, int KB, int64_t off) {
    PackDesc& d = pp.d[c++];
    d.src = reinterpret_cast<const bf16_t*>(src); d.ld = ld; d.N = N; d.K = K; d.transpose = tr;
    d.n_tiles = n_tiles; d.KB = KB; d.dst_off = off; d.block0 = blocks;
    blocks += n_tiles * KB;
  };
  const int kbh = (int)(H / 32), nth8 = (int)(H / 16), ntd8 = 8 * s.ntd;
  for (int i = 0; i <= L; ++i)
    if (!w[i]) return IB_E_ARG;
  for (int i = 0; i < L; ++i) add(w[i], ldw[i], (int)H, i == 0 ? (int)D : (int)H, 0, nth8, i == 0 ? s.kbd : kbh, lo.off_f[i]);
  add(w[L], ldw[L], (int)D, (int)H, 0, ntd8, kbh, lo.off_f[L]);
  for (int i = 1; i < L; ++i) add(w[i], ldw[i], (int)H, (int)H, 1, nth8, kbh, lo.off_b[i]);
  add(w[L], ldw[L], (int)H, (int)D, 1, nth8, s.kbd, lo.off_b[L]);
  pp.count = c; pp.total_blocks = blocks;
  return IB_OK;
}
}  // namespace

extern "C" int ib_mlp_chain_pack(const void* const* w, const int64_t* ldw, void* packed, int64_t D, int64_t H, int L,
                                 ib_stream_t stream) {
  PackParams pp;
  const int rc = build_pack(w, ldw, packed, D, H, L, pp);
  if (rc != IB_OK) return rc;
  hipLaunchKernelGGL(mlp_chain_pack_kernel, dim3(ib_grid_1d(pp.total_blocks, 4)), dim3(256), 0, ib_s(stream), pp);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int64_t ib_mlp_chain_partial_width(int64_t D, int64_t H, int L) {
  ChainShape s;
  if (L < 1 || L > CH_MAXL || !chain_shape(D, H, &s)) return 0;
  return 3 * (int64_t)L * H + 128 * s.ntd + 4;
}

extern "C" int ib_mlp_chain_train(const void* x0, const void* eps, const int64_t* t, const float* sqrt_ab,
                                  const float* sqrt_1mab, int64_t table_rows, const void* e, int64_t ld_e,
                                  const void* packed, const float* const* bias, const float* const* gamma,
                                  const float* const* beta, void* xt, int64_t ld_xt, void* const* u, void* const* h,
                                  void* const* dz, void* dpred, int64_t ld_dpred, float* partial, int64_t ld_part,
                                  void* de_lp, int64_t ld_de, const void* const* in_slots, int64_t M, int64_t T,
                                  int64_t D, int64_t H, int L, float ln_eps, ib_stream_t stream) {
  ChainShape s;
  if (L < 1 || L > CH_MAXL || !chain_shape(D, H, &s)) return IB_E_UNSUPPORTED;
  if (!x0 || !eps || !t || !sqrt_ab || !sqrt_1mab || !e || !packed || !bias || !gamma || !beta || !xt || !u || !h ||
      !dz || !dpred || !partial || M <= 0 || T <= 0 || table_rows <= 0)
    return IB_E_ARG;
  if (ld_e < (int64_t)L * H || ld_e % 4 != 0 || ld_xt < D || ld_xt % 4 != 0 || ld_dpred < D || ld_dpred % 4 != 0) return IB_E_ARG;
  auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) % 8) == 0; };
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) % 16) == 0; };
  if (!al8(x0) || !al8(eps) || !al8(e) || !al8(xt) || !al8(dpred) || !al16(packed) || !al16(partial)) return IB_E_ARG;
  if (ld_part < ib_mlp_chain_partial_width(D, H, L) || ld_part % 4 != 0) return IB_E_ARG;
  PackLayout lo;
  chain_layout(D, H, L, s, &lo);
  ChainParams p{};
  p.x0 = (const bf16_t*)x0; p.eps = (const bf16_t*)eps; p.t = t; p.slots = in_slots;
  p.sqrt_ab = sqrt_ab; p.sqrt_1mab = sqrt_1mab;
  p.table_rows = (int)table_rows; p.e = (const bf16_t*)e; p.ld_e = ld_e;
  p.M = (int)M; p.T = (int)T; p.D = (int)D; p.L = L;
  int P;
  const int nwg = ib_mlp_chain_workgroups(M, &P);
  p.P = P;
  if (de_lp) {   // only meaningful when every workgroup's panel is exactly one window
    if (P != T || M % T != 0 || ld_de < (int64_t)L * H || ld_de % 4 != 0 || !al8(de_lp)) return IB_E_ARG;
  }
  p.de_lp = (bf16_t*)de_lp; p.ld_de = ld_de;
  const bf16_t* pk = (const bf16_t*)packed;
  for (int i = 0; i <= L; ++i) {
    p.wf[i] = pk + lo.off_f[i];
    p.wb[i] = lo.off_b[i] >= 0 ? pk + lo.off_b[i] : nullptr;
    if (!bias[i] || !al16(bias[i])) return IB_E_ARG;
    p.bias[i] = bias[i];
  }
  for (int i = 0; i < L; ++i) {
    if (!gamma[i] || !beta[i] || !h[i] || !dz[i]) return IB_E_ARG;
    if (!al16(gamma[i]) || !al16(beta[i]) || (u[i] && !al16(u[i])) || !al16(h[i]) || !al16(dz[i])) return IB_E_ARG;
    p.gamma[i] = gamma[i]; p.beta[i] = beta[i];
    p.u[i] = (bf16_t*)u[i]; p.h[i] = (bf16_t*)h[i]; p.dz[i] = (bf16_t*)dz[i];
  }
  p.xt = (bf16_t*)xt; p.ld_xt = ld_xt; p.dpred = (bf16_t*)dpred; p.ld_dpred = ld_dpred;
  p.partial = partial; p.ld_part = ld_part;
  p.gscale = 2.f / ((float)M * (float)D); p.ln_eps = ln_eps;
  p.prof = IB_AB_PROF(g_chain_prof);
  hipStream_t st = ib_s(stream);
  // v2 (row-wise epilogues) is the kernel; IB_CHAIN_V1=1 selects the round-2 kernel for A/B measurements.  v2 keeps u in
  // registers for L <= 2 (u[i] may then be NULL: nothing is stored); deeper stacks pass u through HBM as v1 does.
  static const bool use_v1 = ib_ab_set("IB_CHAIN_V1");
  const bool keep = L <= 2;
  for (int i = 0; i < L; ++i)
    if (!u[i] && (use_v1 || !keep)) return IB_E_ARG;
#define IB_CHAIN_LAUNCH(NTH, NTD, KBD)                                                                                  \
  do {                                                                                                                  \
    if (use_v1) hipLaunchKernelGGL((mlp_chain_kernel<NTH, NTD, KBD>), dim3(nwg), dim3(CH_THREADS), 0, st, p);           \
    else if (keep) hipLaunchKernelGGL((mlp_chain2_kernel<NTH, NTD, KBD, true>), dim3(nwg), dim3(CH_THREADS), 0, st, p); \
    else hipLaunchKernelGGL((mlp_chain2_kernel<NTH, NTD, KBD, false>), dim3(nwg), dim3(CH_THREADS), 0, st, p);          \
  } while (0)
  IB_PATH(use_v1 ? IB_PATH_CHAIN1 : IB_PATH_CHAIN2);
  if (s.nth == 4 && s.ntd == 3 && s.kbd == 10) IB_CHAIN_LAUNCH(4, 3, 10);
  else if (s.nth == 4 && s.ntd == 1 && s.kbd == 2) IB_CHAIN_LAUNCH(4, 1, 2);
  else if (s.nth == 4 && s.ntd == 1 && s.kbd == 4) IB_CHAIN_LAUNCH(4, 1, 4);
  else if (s.nth == 4 && s.ntd == 2 && s.kbd == 8) IB_CHAIN_LAUNCH(4, 2, 8);
  else if (s.nth == 4 && s.ntd == 3 && s.kbd == 12) IB_CHAIN_LAUNCH(4, 3, 12);
  else if (s.nth == 4 && s.ntd == 4 && s.kbd == 16) IB_CHAIN_LAUNCH(4, 4, 16);
  else if (s.nth == 2 && s.ntd == 3 && s.kbd == 10) IB_CHAIN_LAUNCH(2, 3, 10);
  else if (s.nth == 1 && s.ntd == 1 && s.kbd == 2) IB_CHAIN_LAUNCH(1, 1, 2);
  else if (s.nth == 1 && s.ntd == 1 && s.kbd == 4) IB_CHAIN_LAUNCH(1, 1, 4);
  else return IB_E_UNSUPPORTED;
#undef IB_CHAIN_LAUNCH
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// ---- fused time-embedding MLP forward: e = W2 silu(W1 sinus(t) + b1) + b2 for B windows in ONE launch (the per-op
// plan needs a gather and two M = B GEMMs whose 8-16 workgroups are pure latency: ~28 us of kernels + 3 boundaries on
// the step's critical path).  Workgroup = 64 windows x one group of 128 output columns; every column group recomputes
// the (tiny) hidden layer for its rows, so there is no inter-workgroup dependency.  Weights are read as they are
// (row-major bf16, fragment-shaped 16-byte loads): nothing to pack, so this runs beside the chain's weight packing.
namespace {
// all KB x NT weight fragments of a wave's slice, issued at once (they depend on nothing: one L2 round trip for the
// whole slice instead of one per k-block) ...
template <int NT, int KB>
__device__ __forceinline__ void rowmajor_load(const bf16_t* __restrict__ w, int64_t ldw, int n0, int lane,
                                              bf16x8_t (&wr)[KB][NT]) {
  const bf16_t* wl = w + (int64_t)(n0 + (lane & 15)) * ldw + 8 * (lane >> 4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int u = 0; u < NT; ++u) wr[kb][u] = *reinterpret_cast<const bf16x8_t*>(wl + (int64_t)16 * u * ldw + 32 * kb);
}
// ... and consumed later (MT m-tiles of 16 rows)
template <int NT, int KB, int MT>
__device__ __forceinline__ void reg_gemm(const bf16x8_t (&wr)[KB][NT], const unsigned char* abuf, int rs, int lane,
                                         f32x4_t (&acc)[MT][NT]) {
  const unsigned char* arow = abuf + (lane & 15) * rs + 16 * (lane >> 4);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    bf16x8_t fa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const bf16x8_t*>(arow + 16 * mt * rs + 64 * kb);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[mt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[kb][u], fa[mt], acc[mt][u], 0, 0, 0);
  }
}

struct TimeFwdParams {
  const float* table; int64_t table_rows; const int64_t* t;
  const void* const* slots;        // optional {.., .., t} device array (see ChainParams)
  const bf16_t* w1; int64_t ldw1; const float* b1;
  const bf16_t* w2; int64_t ldw2; const float* b2;
  bf16_t* s; bf16_t* zu; bf16_t* u; bf16_t* e; int64_t ld_e;
  int B, out, ncg, time_blocks;
  long long* prof;      // TIMING-ONLY: [time_blocks][8] stamps (ib_debug_set_chain_prof), else NULL
};

// NT1 = hidden / 128, KB1 = temb / 32 ; hidden = 128 * NT1 = 32 * KB2 ; output columns: 128 per workgroup (16 per wave);
// rows (windows) per workgroup: 16 * MT.  Every column group recomputes the hidden layer of its rows, and that epilogue is
// the launch's VALU cost (two quarter-rate transcendentals per element: 64 rows x 512 columns on one CU were 4.3 us of
// the 14 us a workgroup lived).  MT = 1 (16 windows per workgroup, four times the workgroups) is taken while the grid
// still fits the chip in one round; large batches keep MT = 4.
// Blocks >= p.time_blocks of the same launch pack the chain kernel's weights (ib_mlp_chain_prep): the two jobs are
// independent and each far too small to fill the chip; one launch saves a kernel boundary (~4.5 us in a captured step).
template <int NT1, int KB1, int MT>
__global__ __launch_bounds__(CH_THREADS) void time_mlp_fwd_kernel(TimeFwdParams p, PackParams pp) {
  constexpr int HID = 128 * NT1, KB2 = 4 * NT1, TE = 32 * KB1, ROWS = 16 * MT;
  constexpr int RS1 = TE * 2 + 16, RS2 = HID * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS * RS1 + ROWS * RS2];
  unsigned char* simg = smem;
  unsigned char* uimg = smem + ROWS * RS1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
  if ((int)blockIdx.x >= p.time_blocks) {
    pack_blocks(pp, ((int)blockIdx.x - p.time_blocks) * CH_WAVES + wave, ((int)gridDim.x - p.time_blocks) * CH_WAVES, lane);
    return;
  }
  const int r0 = ((int)blockIdx.x / p.ncg) * ROWS, cg = (int)blockIdx.x % p.ncg;
  const int nrows = min(ROWS, p.B - r0);
#define T_STAMP(k) do { if (p.prof && tid == 0) p.prof[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
  T_STAMP(0);
  bf16x8_t wr1[KB1][NT1], wr2[KB2][1];
  // gather: sinusoid rows (fp32 table) -> bf16 image; the first column group also writes them out (wgrad operand).
  // Thread (row group tid / (TE/4), column piece tid % (TE/4)) handles NG rows (threads beyond the rows idle).  Issue
  // order, pinned with scheduling barriers: timesteps (an L2 hit: 0.2 us) -> table rows (2.8 us: the launch's first touch
  // of that buffer) -> weight slices (32 KiB per wave: 3.6 us through one CU's L2 port).  Measured with stamps between
  // them: the vector memory pipe serves requests in order, so table rows requested BEHIND the weights start their 2.8 us
  // only when the last weight request has gone out (gather = 0.2 + 3.6 + 2.8 us); left alone the compiler put half of
  // them there.  Requested first they overlap the weight stream, which stage 1 waits for anyway.
  {
    constexpr int PPR = TE / 4, RPI = CH_THREADS / PPR, NG = (ROWS + RPI - 1) / RPI;
    const int64_t* tsrc = p.slots ? reinterpret_cast<const int64_t*>(p.slots[2]) : p.t;
    const int c = (tid % PPR) * 4, rb = tid / PPR;
    int64_t kk[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int row = rb + j * RPI;
      kk[j] = tsrc[r0 + min(row, nrows - 1)];
    }
    float4 v[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int64_t k = kk[j] < 0 ? 0 : (kk[j] >= p.table_rows ? p.table_rows - 1 : kk[j]);
      v[j] = *reinterpret_cast<const float4*>(p.table + k * TE + c);
    }
    __builtin_amdgcn_sched_barrier(0);
    rowmajor_load<NT1, KB1>(p.w1, p.ldw1, wave * 16 * NT1, lane, wr1);
    rowmajor_load<1, KB2>(p.w2, p.ldw2, cg * 128 + wave * 16, lane, wr2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int row = rb + j * RPI;
      if (row >= ROWS) continue;
      bf16x4_t o = pack4(v[j].x, v[j].y, v[j].z, v[j].w);
      if (row >= nrows) o = pack4(0.f, 0.f, 0.f, 0.f);
      else if (cg == 0) *reinterpret_cast<bf16x4_t*>(p.s + (int64_t)(r0 + row) * TE + c) = o;
      *reinterpret_cast<bf16x4_t*>(simg + row * RS1 + c * 2) = o;
    }
  }
  __syncthreads();
  T_STAMP(1);
  {
    f32x4_t acc[MT][NT1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int u = 0; u < NT1; ++u) acc[mt][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    reg_gemm<NT1, KB1, MT>(wr1, simg, RS1, lane, acc);
    const int colb = wave * 16 * NT1 + 4 * g;
#pragma unroll
    for (int u = 0; u < NT1; ++u) {
      const int col = colb + 16 * u;
      const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + col);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int lr = 16 * mt + l16;
        bf16x4_t zb, ub;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          zb[r] = (bf16_t)(acc[mt][u][r] + bb[r]);
          const float x = (float)zb[r];
          ub[r] = (bf16_t)(x * fast_sigmoid(x));
        }
        if (lr >= nrows) ub = pack4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<bf16x4_t*>(uimg + lr * RS2 + col * 2) = ub;
        if (cg == 0 && lr < nrows) {
          *reinterpret_cast<bf16x4_t*>(p.zu + (int64_t)(r0 + lr) * HID + col) = zb;
          *reinterpret_cast<bf16x4_t*>(p.u + (int64_t)(r0 + lr) * HID + col) = ub;
        }
      }
    }
  }
  __syncthreads();
  T_STAMP(2);
  {
    f32x4_t acc[MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int n0 = cg * 128 + wave * 16;
    reg_gemm<1, KB2, MT>(wr2, uimg, RS2, lane, acc);
    T_STAMP(3);
    const int col = n0 + 4 * g;
    const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + col);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int lr = 16 * mt + l16;
      if (lr < nrows)
        *reinterpret_cast<bf16x4_t*>(p.e + (int64_t)(r0 + lr) * p.ld_e + col) =
            pack4(acc[mt][0][0] + bb[0], acc[mt][0][1] + bb[1], acc[mt][0][2] + bb[2], acc[mt][0][3] + bb[3]);
    }
  }
  T_STAMP(4);
#undef T_STAMP
}
}  // namespace

extern "C" int ib_time_mlp_fwd_supported(int64_t temb, int64_t hidden, int64_t out) {
  const bool shape = (temb == 128 && hidden == 512) || (temb == 32 && hidden == 128);
  return (shape && out > 0 && out % 128 == 0) ? 1 : 0;
}

namespace {
int time_fwd_launch(const float* table, int64_t table_rows, const int64_t* t, const void* w1, int64_t ldw1,
                    const float* b1, const void* w2, int64_t ldw2, const float* b2, void* s, void* zu, void* u, void* e,
                    int64_t ld_e, int64_t B, int64_t temb, int64_t hidden, int64_t out, const PackParams& pp,
                    const void* const* in_slots, hipStream_t st) {
  if (!ib_time_mlp_fwd_supported(temb, hidden, out)) return IB_E_UNSUPPORTED;
  if (!table || !t || !w1 || !b1 || !w2 || !b2 || !s || !zu || !u || !e || B <= 0 || table_rows <= 0) return IB_E_ARG;
  if (ldw1 < temb || ldw2 < hidden || ld_e < out || ldw1 % 8 != 0 || ldw2 % 8 != 0 || ld_e % 4 != 0) return IB_E_ARG;
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
  if (!al(table, 16) || !al(w1, 16) || !al(w2, 16) || !al(b1, 16) || !al(b2, 16) || !al(s, 8) || !al(zu, 8) ||
      !al(u, 8) || !al(e, 8))
    return IB_E_ARG;
  TimeFwdParams p{};
  p.table = table; p.table_rows = table_rows; p.t = t; p.slots = in_slots;
  p.w1 = (const bf16_t*)w1; p.ldw1 = ldw1; p.b1 = b1; p.w2 = (const bf16_t*)w2; p.ldw2 = ldw2; p.b2 = b2;
  p.s = (bf16_t*)s; p.zu = (bf16_t*)zu; p.u = (bf16_t*)u; p.e = (bf16_t*)e; p.ld_e = ld_e;
  p.B = (int)B; p.out = (int)out;
  p.ncg = (int)(out / 128);
  // 16 windows per workgroup while that grid fits one round of the chip (beside the packing workgroups), else 64
  static const bool no_mt1 = ib_ab_set("IB_TIME_FWD_MT4");
  const bool mt1 = !no_mt1 && temb == 128 && (int64_t)p.ncg * ((B + 15) / 16) <= 256;
  const int rows = mt1 ? 16 : CH_ROWS;
  p.time_blocks = p.ncg * (int)((B + rows - 1) / rows);
  p.prof = IB_AB_PROF(g_chain_prof);
  const int pack_wgs = pp.total_blocks > 0 ? ib_grid_1d(pp.total_blocks, CH_WAVES, mt1 ? 128 : 224) : 0;
  const dim3 grid((unsigned)(p.time_blocks + pack_wgs));
  if (temb == 128 && hidden == 512) {
    if (mt1) hipLaunchKernelGGL((time_mlp_fwd_kernel<4, 4, 1>), grid, dim3(CH_THREADS), 0, st, p, pp);
    else hipLaunchKernelGGL((time_mlp_fwd_kernel<4, 4, 4>), grid, dim3(CH_THREADS), 0, st, p, pp);
  } else {
    hipLaunchKernelGGL((time_mlp_fwd_kernel<1, 1, 4>), grid, dim3(CH_THREADS), 0, st, p, pp);
  }
  IB_CHECK_LAUNCH();
  return IB_OK;
}
}  // namespace

extern "C" int ib_time_mlp_fwd(const float* table, int64_t table_rows, const int64_t* t, const void* w1, int64_t ldw1,
                               const float* b1, const void* w2, int64_t ldw2, const float* b2, void* s, void* zu,
                               void* u, void* e, int64_t ld_e, int64_t B, int64_t temb, int64_t hidden, int64_t out,
                               ib_stream_t stream) {
  PackParams pp{};
  return time_fwd_launch(table, table_rows, t, w1, ldw1, b1, w2, ldw2, b2, s, zu, u, e, ld_e, B, temb, hidden, out, pp,
                         nullptr, ib_s(stream));
}

// one launch for everything the chain kernel waits on: the time-embedding MLP forward AND the weight packing
extern "C" int ib_mlp_chain_prep(const float* table, int64_t table_rows, const int64_t* t, const void* w1, int64_t ldw1,
                                 const float* b1, const void* w2, int64_t ldw2, const float* b2, void* s, void* zu,
                                 void* u, void* e, int64_t ld_e, int64_t B, int64_t temb, int64_t hidden, int64_t out,
                                 const void* const* w, const int64_t* ldw, void* packed, int64_t D, int64_t H, int L,
                                 const void* const* in_slots, ib_stream_t stream) {
  PackParams pp;
  const int rc = build_pack(w, ldw, packed, D, H, L, pp);
  if (rc != IB_OK) return rc;
  return time_fwd_launch(table, table_rows, t, w1, ldw1, b1, w2, ldw2, b2, s, zu, u, e, ld_e, B, temb, hidden, out, pp,
                         in_slots, ib_s(stream));
}

// ---- fused time-embedding MLP backward (hidden layer):  dzu = (de W2) * silu'(zu),  dW1 = dzu^T s,  db1 = colsum(dzu)
// in ONE launch whose workgroups are independent.  The per-op form was a dependent pair on a forked stream -- the few-row
// dgrad (32 workgroups walking all of K = L*H: 19 us) and the weight gradient of its output (5.6 us) -- that cost the step
// 13.8 us even hidden behind the grouped weight gradients (fork / join of the captured graph + the CUs it takes), 28.6 us
// inline.  Here a workgroup owns 16 hidden columns x 64 windows: the W2 slice is transposed into LDS once, the four waves
// SPLIT THE REDUCTION over L*H (8 k-blocks each instead of 32 in sequence) and combine their accumulators through LDS in
// wave order, silu' is applied, and the same workgroup multiplies its [64 x 16] block of dzu into the gathered sinusoid rows:
// a partial dW1 [16, temb] per 64-window row group.  The row groups' partials are fp32 slabs the optimizer (or
// ib_step_reduce) sums in row-group order, like every split-M weight gradient -- deterministic, no atomics.
namespace {
constexpr int TB_THREADS = 256;
template <int TE>
__global__ __launch_bounds__(TB_THREADS) void time_mlp_bwd_kernel(TimeBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[time_bwd_lds<TE, 4>(TB_MAXOUT)];
  time_bwd_body<TE, 4>(p, (int)blockIdx.x, (int)blockIdx.y, 1, smem);
}
}  // namespace

extern "C" int ib_time_mlp_bwd_supported(int64_t temb, int64_t hidden, int64_t out) {
  return ((temb == 128 || temb == 32) && hidden > 0 && hidden % 16 == 0 && out >= 128 && out % 128 == 0 && out <= TB_MAXOUT) ? 1 : 0;
}
extern "C" int ib_time_mlp_bwd_slab_count(int64_t B) { return (int)((B + TB_ROWS - 1) / TB_ROWS); }

extern "C" int ib_time_mlp_bwd(const void* de, int64_t ld_de, const void* w2, int64_t ldw2, const void* zu, int64_t ldzu,
                               const void* s, int64_t lds, float* dw1_slabs, float* db1_slabs, int64_t B, int64_t temb,
                               int64_t hidden, int64_t out, ib_stream_t stream) {
  if (!ib_time_mlp_bwd_supported(temb, hidden, out)) return IB_E_UNSUPPORTED;
  if (!de || !w2 || !zu || !s || !dw1_slabs || !db1_slabs || B <= 0) return IB_E_ARG;
  if (ld_de < out || ldw2 < hidden || ldzu < hidden || lds < temb || ld_de % 8 || ldw2 % 8 || ldzu % 4 || lds % 8) return IB_E_ARG;
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
  if (!al(de, 16) || !al(w2, 16) || !al(zu, 8) || !al(s, 16) || !al(dw1_slabs, 16) || !al(db1_slabs, 16)) return IB_E_ARG;
  TimeBwdParams p{(const bf16_t*)de, ld_de, (const bf16_t*)w2, ldw2, (const bf16_t*)zu, ldzu, (const bf16_t*)s, lds,
                  dw1_slabs, db1_slabs, (int)B, (int)out, (int)hidden};
  const dim3 grid((unsigned)(hidden / 16), (unsigned)ib_time_mlp_bwd_slab_count(B));
  if (temb == 128) hipLaunchKernelGGL((time_mlp_bwd_kernel<128>), grid, dim3(TB_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL((time_mlp_bwd_kernel<32>), grid, dim3(TB_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// out = scale * sum(partial[0..parts))   (fixed order; the chain kernel's per-workgroup loss sums)
namespace {
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int parts, float scale,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < parts; i += 256) s += partial[i];
  s = ib_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (((red[0] + red[1]) + red[2]) + red[3]) * scale;
}
}  // namespace

extern "C" int ib_sum_partials(const float* partial, int64_t parts, float scale, float* out, ib_stream_t stream) {
  if (!partial || !out || parts <= 0) return IB_E_ARG;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ib_s(stream), partial, (int)parts, scale, out);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
