// Weight-gradient GEMM for long reductions, bf16, gfx950:  dW[N,K] = sum_m dz[m][n] * x[m][k]   (both operands are indexed
// [reduction][column]: "TN"), split over the reduction into fp32 partial slabs [split][N][K] (summed later in a fixed order).
//
// Why a second kernel: the 128 x 128 ring kernel of gemm.hip re-reads every operand slice once per output tile of the other
// dimension; at the transformer denoiser's shapes (M = 12800 token rows; [N,K] = [2048,512], [512,2048], [1536,512],
// [512,512]) rocprofv3 counts 699 MB of L2-miss traffic per grouped launch against 210 MB of operands -- the launch runs
// AT the HBM rate (5.3 TB/s), not at the MFMA rate.  This kernel uses the pipeline of gemm_nt.hip with a 256 (n) x 128 (k)
// output tile, which halves that traffic, and keeps a split's tiles on one XCD so they stream the same 64-row slices
// through one L2 together.
//
//   * LDS image of a stage = the 64 reduction rows as they lie in memory: [64][256 n] (512-byte rows) and [64][128 k]
//     (256-byte rows), filled by global_load_lds_dwordx4 (two / four rows per 1-KiB wave instruction);
//   * MFMA fragments (8 consecutive reduction indices of one column) come from ds_read_b64_tr_b16, the transposing LDS
//     read: 4 rows x 16 columns per 16-lane group.  The 32-byte column chunks of a row are XOR-swizzled with
//     s(m) = (m & 3) | ((m >> 3) & 1) << 2, so the eight row segments a 32-lane half reads in one instruction fall on
//     eight different 32-byte slots of the 256-byte bank row; as in gemm_nt.hip the permutation is applied on the per-lane
//     SOURCE address of the LDS-DMA;
//   * persistent workgroups, continuous stage stream across work items (problem, split, tile), one barrier per K step in
//     mid-step, fragments double-buffered, LDS-DMA issue between the MFMAs (see gemm_nt.hip);
//   * epilogue: fp32 accumulators straight to the slab (a lane holds 4 consecutive k of one n: 16-byte stores); the bias
//     gradient's per-split partial sums ride along as one more MFMA per row tile against an all-ones fragment.
#include <type_traits>

#include "ib_common.h"
#include "gemm_nt.h"
#include "time_bwd.h"

namespace {

constexpr int TM = 256, TK = 128, BK = 64, NS = 3, TN_THREADS = 512;
constexpr int A_ROW = TM * 2, B_ROW = TK * 2;                                  // bytes per LDS image row
constexpr int A_BYTES = BK * A_ROW, B_BYTES = BK * B_ROW, STAGE = A_BYTES + B_BYTES;      // 32768 + 16384
constexpr int LDS_BYTES = NS * STAGE;
constexpr int TN_MAX = 6;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// one MFMA fragment = rows m .. m+3 (lo) and m+4 .. m+7 (hi) of a 16-column chunk, transposed by the LDS
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ bf16x8_t lds_read_tr(unsigned addr) {
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF_LO));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF_HI));
  u32x4_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void frags_ready(bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
}
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct TnProblem {
  const bf16_t* A; const bf16_t* B; int lda, ldb;        // dz [M, N], x [M, K]
  int M, N, K;
  float* C;                                               // slabs [splits][N][K]
  float* dbias;                                           // optional [splits][N]
  int tiles_n, tiles_k, splits, chunk;                    // chunk = M / splits, a multiple of 64, >= 256
  int item0;                                              // first work item of this problem
};
// Rider: the time-embedding MLP's hidden-layer backward (time_bwd.h) as EXTRA workgroups of this launch.  The MLP denoiser's
// grouped launch has 204 work items for 256 CUs: 52 CUs idle for its whole 30 us, enough for the rider's 128 short workgroups
// (3 rounds of ~8 us) -- as a launch of its own it cost the step 11 us + a boundary.  Blocks [tn_grid, tn_grid + tb_blocks)
// are rider workgroups (one per 16 hidden columns, every row group); they share nothing with the GEMM's workgroups.
struct TnParams {
  TnProblem pr[TN_MAX]; int n, items; long long* prof;
  int tn_grid;                       // persistent GEMM workgroups (= the stride of the work-item walk)
  TimeBwdParams tb; int tb_blocks, tb_cgs, tb_te;
};
#ifdef IB_AB
long long* g_tn_prof = nullptr;
#endif

// BIAS: some problem of the launch wants its bias gradient.  The extra MFMAs then run unconditionally (every item, every
// wave): the K loop is bound by the LDS-DMA ingest (0.82 us per step against 0.49 us of MFMA), so +25 % matrix work is free,
// while a wave-uniform branch around them cost 50 spilled registers.
template <bool BIAS>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_kernel(TnParams P) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  if ((int)blockIdx.x >= P.tn_grid) {                          // rider workgroups (wave-uniform: whole workgroups)
    // one rider workgroup per 16 hidden columns walks ALL row groups (the launch has ~50 CUs to spare, a rider round is a
    // ~10-us latency chain whatever its size: 128 one-group riders took three rounds and stretched the launch 30 -> 37.7 us)
    const int b = (int)blockIdx.x - P.tn_grid;
    const int nrg = (P.tb.B + TB_ROWS - 1) / TB_ROWS;
    if (P.tb_te == 128) time_bwd_body<128, 8>(P.tb, b, 0, nrg, smem);
    else time_bwd_body<32, 8>(P.tb, b, 0, nrg, smem);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = P.tn_grid;
  const int first = ib_xcd_remap((int)blockIdx.x, nwg);
  if (first >= P.items) return;
  const unsigned smem0 = lds_off(smem);

  // fragment read bases: tile t of the n rows (A image) / tile u of the k columns (B image); the XOR with the row's
  // swizzle differs per lane and per tile, so four bases per operand
  // (kept as ONE base per operand: tile t only XORs bits 5-6 of the address -- (t ^ q) << 5 -- which no other term touches)
  unsigned ab0, bb0;
  {
    const int lane = lane_now();
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    ab0 = smem0 + (unsigned)((8 * g + q) * A_ROW + 32 * ((((wm ^ (g & 1)) << 2)) | q) + 8 * pq);
    bb0 = smem0 + A_BYTES + (unsigned)((8 * g + q) * B_ROW + 32 * ((((wn ^ (g & 1)) << 2)) | q) + 8 * pq);
  }
  auto read_frags = [&](unsigned so, auto ksc, bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
    constexpr int KS = decltype(ksc)::value;
    const unsigned a = ab0 + so, b = bb0 + so;
    fa[0] = lds_read_tr<KS * 32 * A_ROW, KS * 32 * A_ROW + 4 * A_ROW>(a);
    fa[1] = lds_read_tr<KS * 32 * A_ROW, KS * 32 * A_ROW + 4 * A_ROW>(a ^ 32u);
    fa[2] = lds_read_tr<KS * 32 * A_ROW, KS * 32 * A_ROW + 4 * A_ROW>(a ^ 64u);
    fa[3] = lds_read_tr<KS * 32 * A_ROW, KS * 32 * A_ROW + 4 * A_ROW>(a ^ 96u);
    fb[0] = lds_read_tr<KS * 32 * B_ROW, KS * 32 * B_ROW + 4 * B_ROW>(b);
    fb[1] = lds_read_tr<KS * 32 * B_ROW, KS * 32 * B_ROW + 4 * B_ROW>(b ^ 32u);
    fb[2] = lds_read_tr<KS * 32 * B_ROW, KS * 32 * B_ROW + 4 * B_ROW>(b ^ 64u);
    fb[3] = lds_read_tr<KS * 32 * B_ROW, KS * 32 * B_ROW + 4 * B_ROW>(b ^ 96u);
  };

  // work item -> (problem, split, tile); all values wave-uniform
  struct Item { int e, split, i0, j0, nk; };
  auto decode = [&](int item) {
    int e = 0;
    for (int j = 1; j < P.n; ++j)
      if (item >= P.pr[j].item0) e = j;
    const TnProblem& q = P.pr[e];
    const int local = item - q.item0;
    const int per_split = q.tiles_n * q.tiles_k;
    const int s = local / per_split, tl = local % per_split;
    return Item{e, s, (tl / q.tiles_k) * TM, (tl % q.tiles_k) * TK, q.chunk / BK};
  };
  // staging sources of an item: per-lane 32-bit element offsets of its six LDS-DMA pieces at reduction row 0 of the
  // chunk (A: rows 2 (w + 8 j) + lane / 32; B: rows 4 (w + 8 j) + lane / 16), the 16-byte piece = position ^ swizzle(row),
  // clamped onto the last valid piece of a ragged tile (duplicate columns only feed outputs that are never stored)
  auto item_ptrs = [&](const Item& it, unsigned& qa, unsigned& qb) {
    const TnProblem& q = P.pr[it.e];
    const int lane = lane_now();
    {
      const int r = (2 * wave + (lane >> 5));               // row within the stage, modulo 16 (j adds multiples of 16)
      const int sw = ((r & 3) | (((r >> 3) & 1) << 2)) << 1;
      const int pieces = (min(TM, q.N - it.i0) + 7) >> 3;
      const int src = min((lane & 31) ^ sw, pieces - 1);
      qa = (unsigned)(it.split * q.chunk + 2 * wave + (lane >> 5)) * (unsigned)q.lda + (unsigned)(it.i0 + 8 * src);
    }
    {
      const int r = (4 * wave + (lane >> 4));               // modulo 32 here; the swizzle only looks at bits 0, 1 and 3
      const int sw = ((r & 3) | (((r >> 3) & 1) << 2)) << 1;
      const int pieces = (min(TK, q.K - it.j0) + 7) >> 3;
      const int src = min((lane & 15) ^ sw, pieces - 1);
      qb = (unsigned)(it.split * q.chunk + 4 * wave + (lane >> 4)) * (unsigned)q.ldb + (unsigned)(it.j0 + 8 * src);
    }
  };
  // piece j of stage kt: A pieces are 16 rows apart (2 rows per 1-KiB instruction x 8 waves), B pieces 32 rows apart --
  // wave-uniform offsets on ONE per-lane offset per operand
  auto piece = [&](const TnProblem& q, unsigned qa, unsigned qb, int j, int kt, unsigned char* st) {
    if (j < 4)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(q.A + (size_t)qa + (size_t)((unsigned)(kt * BK + 16 * j) * (unsigned)q.lda)),
                                       (lds_void_t*)(st + (wave + 8 * j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_void_t*)(q.B + (size_t)qb + (size_t)((unsigned)(kt * BK + 32 * (j - 4)) * (unsigned)q.ldb)),
                                       (lds_void_t*)(st + A_BYTES + (wave + 8 * (j - 4)) * 1024), 16, 0, 0);
  };

  unsigned pa, pb;
  Item cur = decode(first);
  item_ptrs(cur, pa, pb);
  int slot = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < 6; ++j) piece(P.pr[cur.e], pa, pb, j, s, smem + s * STAGE);
  bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
  wait_vm<12>();
  __builtin_amdgcn_s_barrier();
  bool fresh = true;
  [[maybe_unused]] bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;

  for (int item = first; item < P.items; item += nwg) {
    const TnProblem& q = P.pr[cur.e];
    const bool has_next = item + nwg < P.items;
    const Item nxt = has_next ? decode(item + nwg) : cur;
    const int nk = cur.nk;
    const bool store_bias = BIAS && q.dbias != nullptr && cur.j0 == 0 && wn == 0;
    const bool ragged = (cur.i0 + TM > q.N) || (cur.j0 + TK > q.K) || (q.dbias != nullptr && cur.j0 == 0);

    f32x4_t acc[4][4];
    [[maybe_unused]] f32x4_t accb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (BIAS) accb[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    auto step = [&](auto issue_c, auto wait_c, auto read_c, int kt) {
      constexpr int ISSUE = decltype(issue_c)::value, WAIT = decltype(wait_c)::value;
      constexpr bool READ = decltype(read_c)::value != 0;
      [[maybe_unused]] unsigned pn_a, pn_b;
      const unsigned so = (unsigned)(slot * STAGE);
      read_frags(so, K1{}, fa1, fb1);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[u], fa0[t], acc[t][u], 0, 0, 0);
      if constexpr (BIAS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa0[t], accb[t], 0, 0, 0);
      }
      frags_ready(fa1, fb1);
      if constexpr (WAIT >= 0) wait_vm<WAIT>();
      __builtin_amdgcn_s_barrier();
      const int slot1 = slot == NS - 1 ? 0 : slot + 1;
      if constexpr (READ) read_frags((unsigned)(slot1 * STAGE), K0{}, fa0, fb0);
      unsigned char* st = smem + so;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[u], fa1[t], acc[t][u], 0, 0, 0);
        if constexpr (ISSUE == 1) {
          piece(q, pa, pb, t, kt + 3, st);
          if (t < 2) piece(q, pa, pb, 4 + t, kt + 3, st);
        } else if constexpr (ISSUE == 2) {
          if (t == 0) item_ptrs(nxt, pn_a, pn_b);
          piece(P.pr[nxt.e], pn_a, pn_b, t, kt + 3 - nk, st);
          if (t < 2) piece(P.pr[nxt.e], pn_a, pn_b, 4 + t, kt + 3 - nk, st);
        }
      }
      if constexpr (BIAS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa1[t], accb[t], 0, 0, 0);
      }
      if constexpr (READ) frags_ready(fa0, fb0);
      slot = slot1;
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
    using W6 = std::integral_constant<int, 6>; using W22 = std::integral_constant<int, 22>;
    using W0 = std::integral_constant<int, 0>; using WN = std::integral_constant<int, -1>;
    read_frags((unsigned)(slot * STAGE), K0{}, fa0, fb0);
    frags_ready(fa0, fb0);
    // first step: the sixteen slab stores of the previous item (uniform per thread unless that item was ragged or
    // carried bias stores) may stay in flight behind stage 1
    if (fresh) step(C1{}, W6{}, C1{}, 0); else step(C1{}, W22{}, C1{}, 0);
    for (int kt = 1; kt < nk - 3; ++kt) step(C1{}, W6{}, C1{}, kt);
    if (has_next) {
      step(C2{}, W6{}, C1{}, nk - 3); step(C2{}, W6{}, C1{}, nk - 2); step(C0{}, W6{}, C0{}, nk - 1);
    } else {
      step(C0{}, W6{}, C1{}, nk - 3); step(C0{}, W0{}, C1{}, nk - 2); step(C0{}, WN{}, C0{}, nk - 1);
    }

    // ---- epilogue: fp32 accumulators to the slab of this split (no LDS: the stage slots keep streaming)
    {
      const int lane = lane_now();
      float* slab = q.C + (size_t)cur.split * (size_t)q.N * (size_t)q.K;
      const int n0 = cur.i0 + wm * 64 + (lane & 15), k0 = cur.j0 + wn * 64 + 4 * (lane >> 4);
      if (!ragged) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            *reinterpret_cast<f32x4_t*>(slab + (size_t)(n0 + 16 * t) * q.K + k0 + 16 * u) = acc[t][u];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (n0 + 16 * t < q.N && k0 + 16 * u < q.K)
              *reinterpret_cast<f32x4_t*>(slab + (size_t)(n0 + 16 * t) * q.K + k0 + 16 * u) = acc[t][u];
        if (store_bias && (lane >> 4) == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (n0 + 16 * t < q.N) q.dbias[(size_t)cur.split * q.N + n0 + 16 * t] = accb[t][0];
        }
      }
    }
    fresh = ragged;
    if (has_next) {
      // the slot of the item's last stage is free (every wave left it at the barrier of the last step): the next item's
      // stage 2 goes there
      const int lslot = slot == 0 ? NS - 1 : slot - 1;
      item_ptrs(nxt, pa, pb);
#pragma unroll
      for (int j = 0; j < 6; ++j) piece(P.pr[nxt.e], pa, pb, j, 2, smem + lslot * STAGE);
      cur = nxt;
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// number of reduction splits the TN kernel uses for a problem that shares its launch with `group - 1` others (0: the
// problem does not qualify).  Splits divide M / 64 exactly (every work item has the same number of K steps, >= 4).
int ib_gemm_tn_splits(int64_t M, int64_t N, int64_t K, int group) {
  static const int off = ib_ab_int("IB_NO_TN", 0);
  // work items per launch the split selection aims at (256 = one per CU).  More items fill the tail rounds better but every
  // split is one more fp32 slab of the gradient for the step's final reduction to read (measured, B = 256, T = 50: MLP denoiser
  // step 0.227 / 0.214 / 0.227 / 0.227 ms and transformer step 2.81 / 2.50 / 2.69 / 2.55 ms at 192 / 256 / 384 / 512).
  // IB_TN_TARGET: tuning override.
  static const int total = ib_ab_int("IB_TN_TARGET", 256);
  // a lone problem must be long (short reductions have their own one-pass kernel); inside a group a short one rides along
  if (off || M % 64 != 0 || M < 256 || (group <= 1 && M < 4096) || N < 64 || K < 64 || K % 4 != 0) return 0;
  const int64_t steps = M / 64;
  const int64_t tiles = ((N + TM - 1) / TM) * ((K + TK - 1) / TK);
  const int64_t target = group <= 1 ? total : (total + group - 1) / group;
  int best = 0;
  int64_t best_d = 0;
  for (int64_t s = 1; s <= 32 && s <= steps / 4; ++s) {          // splits divide the K steps exactly, >= 4 steps each
    if (steps % s) continue;
    const int64_t d = tiles * s > target ? tiles * s - target : target - tiles * s;
    if (best == 0 || d < best_d) { best = (int)s; best_d = d; }
  }
  return best;
}

int ib_gemm_tn_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x, const int64_t* ldx,
                     void* const* workspace, const size_t* workspace_bytes, float* const* dbias_part, int32_t* nslab_out,
                     const int64_t* M, const int64_t* N, const int64_t* K, hipStream_t s, const void* rider, int rider_te) {
  if (n <= 0 || n > TN_MAX) return IB_E_UNSUPPORTED;
  if (!rider) {      // groups of 256-multiples (the transformer layers): 256 x 256 tiles, equal-length work items
    const int rc = ib_gemm_tn256_multi(n, dz, lddz, x, ldx, workspace, workspace_bytes, dbias_part, nslab_out, M, N, K, s);
    if (rc != IB_E_UNSUPPORTED) return rc;
  }
  TnParams P{};
  P.n = n;
  int items = 0;
  for (int j = 0; j < n; ++j) {
    const int sp = ib_gemm_tn_splits(M[j], N[j], K[j], n);
    if (sp <= 0) return IB_E_UNSUPPORTED;
    if (!al16(dz[j]) || !al16(x[j]) || !al16(workspace[j]) || lddz[j] % 8 || ldx[j] % 8) return IB_E_UNSUPPORTED;
    if (lddz[j] < (N[j] + 7) / 8 * 8 || ldx[j] < (K[j] + 7) / 8 * 8) return IB_E_UNSUPPORTED;
    if (M[j] * lddz[j] >= (int64_t(1) << 31) || M[j] * ldx[j] >= (int64_t(1) << 31)) return IB_E_UNSUPPORTED;
    if ((size_t)sp * (size_t)N[j] * (size_t)K[j] * sizeof(float) > workspace_bytes[j]) return IB_E_WORKSPACE;
    TnProblem& q = P.pr[j];
    q.A = (const bf16_t*)dz[j]; q.B = (const bf16_t*)x[j]; q.lda = (int)lddz[j]; q.ldb = (int)ldx[j];
    q.M = (int)M[j]; q.N = (int)N[j]; q.K = (int)K[j];
    q.C = (float*)workspace[j]; q.dbias = dbias_part ? dbias_part[j] : nullptr;
    q.tiles_n = (int)((N[j] + TM - 1) / TM); q.tiles_k = (int)((K[j] + TK - 1) / TK);
    q.splits = sp; q.chunk = (int)(M[j] / sp);
    q.item0 = items;
    items += q.tiles_n * q.tiles_k * sp;
    nslab_out[j] = sp;
  }
  P.items = items;
  P.prof = IB_AB_PROF(g_tn_prof);
  int grid = items < 256 ? items : 256;
  P.tn_grid = grid;
  if (rider) {
    const TimeBwdParams& tb = *reinterpret_cast<const TimeBwdParams*>(rider);
    if (tb.out % 256 != 0 || (rider_te != 128 && rider_te != 32) ||
        (rider_te == 128 ? time_bwd_lds<128, 8>(tb.out) : time_bwd_lds<32, 8>(tb.out)) > LDS_BYTES)
      return IB_E_UNSUPPORTED;
    P.tb = tb; P.tb_te = rider_te; P.tb_cgs = tb.hidden / 16;
    P.tb_blocks = P.tb_cgs;
    grid += P.tb_blocks;
  }
  bool any_bias = false;
  for (int j = 0; j < n; ++j) any_bias = any_bias || P.pr[j].dbias != nullptr;
  IB_PATH(IB_PATH_TN);
  if (any_bias) hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(grid), dim3(TN_THREADS), 0, s, P);
  else hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(grid), dim3(TN_THREADS), 0, s, P);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
