// Large-M bf16 "NT" GEMM for gfx950: C[M,N] = epilogue( sum_k A[m][k] * B[n][k] ), both operands k-contiguous.
//
// The training shapes of the transformer denoiser (BASELINE.json configs[2]/[3]: M = B*T = 12800 token rows,
// N, K in {512, 1536, 2048}) are what the 128 x 128 ring kernel of gemm.hip handles worst: with K = 512 a tile is 16
// K steps long, so its prologue (first-stage latency) and its epilogue (16 x 8-byte-per-lane stores per wave: store-ISSUE
// bound, ~as long as the K loop) weigh as much as the loop.  This kernel is built around those two ends:
//
//   * 256 x 128 output tile per 512-thread workgroup (8 waves as 4 x 2, 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles):
//     twice the flops per staged byte and per barrier of the 128^2 tile, K steps of 64 (32 MFMAs per wave per barrier);
//   * operands HBM/L2 -> LDS by global_load_lds_dwordx4 only (no VGPR round trip), three 48-KiB stages: two in flight
//     while one is consumed, counted s_waitcnt vmcnt, raw s_barrier, fragment reads as inline-asm ds_read_b128;
//   * LDS image = plain 128-byte rows (64 k) whose eight 16-byte pieces are XOR-swizzled by (row & 7): the fragment
//     reads of a 16-lane group then cover all sixteen 16-byte slots of the 256-byte bank row exactly once
//     (conflict-free), and because LDS-DMA writes lane-linearly the permutation is applied on the per-lane SOURCE
//     address (piece = slot ^ row) -- each row's 8 lanes still fetch one whole 128-byte line;
//   * PERSISTENT workgroups (one per CU) walk the tiles in an XCD-aware order (the column tiles of one row panel stay on
//     one XCD's L2), so consecutive tiles of a workgroup need no new launch and no LDS re-allocation;
//   * epilogue through LDS: every wave drops its 64 x 64 accumulators (bias / activation applied) as bf16 into a
//     [256][128] image, then all 512 threads move whole 16-byte pieces: the activation-derivative operand and the
//     residual addend are read, and C is written, in fully coalesced 256-byte row segments.
//
// Epilogues: forward  C = act(A B^T + bias)                  (nn.Linear + activation, src/models/TransformerBaseline.py:14-18)
//            backward C = (A B^T) * act'(aux) + addend       (autograd of the layer below + residual path; B = W^T here: the
//                                                            caller keeps a transposed bf16 copy of the weight, see
//                                                            ib_transpose_multi)
#include "ib_common.h"
#include "gemm_nt.h"
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 128, BK = 64, NS = 3, NT_THREADS = 512;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;      // 32768 + 16384
constexpr int LDS_BYTES = NS * STAGE;                                                       // 147456
constexpr int CS = BN * 2 + 16;                                                             // C staging row stride (bytes)
static_assert(BM * CS <= 2 * STAGE, "the C staging image overlays two stages");

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
template <int OFF>
__device__ __forceinline__ bf16x8_t lds_read16(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void frags_ready(bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
}
// lane id recomputed where it is needed (2 VALU): values derived from a cached threadIdx.x were spilled, and a spill
// reload is a VMEM operation -- its s_waitcnt vmcnt(0) drains the LDS-DMA stages in flight
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct NtParams {
  const bf16_t* A; const bf16_t* B; int64_t lda, ldb;
  int M, N, K;
  bf16_t* C; int64_t ldc;
  const float* bias;
  const bf16_t* aux; int64_t ldaux;
  const bf16_t* addend; int64_t ldadd;
  int tiles_m, tiles_n;
  // split-K form (SLAB kernels): work item = (split, tile); split s reduces k in [s * kchunk, (s + 1) * kchunk) and writes
  // its fp32 partial tile into slab[s][M][N].  The plain form has splits = 1, kchunk = K.
  float* slab; int splits, kchunk, tiles_mn;
  long long* prof;             // TIMING-ONLY (tools/nt_prof.py): [gridDim.x][16] wall-clock stamps, else NULL
};
#ifdef IB_AB
long long* g_nt_prof = nullptr;
#endif
#define NT_STAMP(k) do { if (p.prof && threadIdx.x == 0 && (k) < 16) p.prof[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

template <int ACT> __device__ __forceinline__ float nt_act(float v) {
  if constexpr (ACT == IB_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return tanhf(v);
  else if constexpr (ACT == IB_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  else if constexpr (ACT == IB_ACT_SILU) return v / (1.f + __expf(-v));
  else if constexpr (ACT == IB_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
  else return v;
}
template <int ACT> __device__ __forceinline__ float nt_act_bwd(float aux) {
  if constexpr (ACT == IB_ACT_RELU) return aux > 0.f ? 1.f : 0.f;
  else if constexpr (ACT == IB_ACT_TANH) return 1.f - aux * aux;
  else if constexpr (ACT == IB_ACT_SIGMOID) return aux * (1.f - aux);
  else if constexpr (ACT == IB_ACT_ELU) return aux > 0.f ? 1.f : aux + 1.f;
  else return 1.f;
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
template <int OFF>
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {          // n wave-uniform, one of {0, 6, 12, 14}
  if (n >= 14) wait_vm<14>();
  else if (n >= 12) wait_vm<12>();
  else if (n >= 6) wait_vm<6>();
  else wait_vm<0>();
}

// One K step = two groups of 16 MFMAs (k sub-steps of 32).  The fragments of a group are read from LDS while the
// previous group's MFMAs run (two register sets F0 / F1), the workgroup barrier sits in the MIDDLE of the step (after the
// wave's last read of the stage, before its first read of the next one), and the six LDS-DMA pieces of stage g + 3 are
// issued between the MFMAs of the second group.  The stage stream is CONTINUOUS across a workgroup's tiles: the next
// tile's stages 0 and 1 are in flight while the current tile's last steps and its epilogue run; only stage 2 of the
// next tile waits for the epilogue (the C staging image borrows that slot).
//
// FWD_ACT: activation of the forward epilogue (bias added first); BWD_ACT: derivative factor taken from `aux`
// (IB_ACT_NONE = none); HAS_ADD: a residual addend is added last.
template <int FWD_ACT, int BWD_ACT, bool HAS_ADD, bool HAS_BIAS, bool SLAB = false>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_nt_kernel(NtParams p) {
  // epilogue operands are requested one K step early (their latency runs beside the last step's MFMAs); EPI_LOADS = how
  // many vector-memory loads that puts behind the prefetched stages (the last step's counted wait leaves them in flight)
  constexpr int EPI_LOADS = (HAS_BIAS ? 4 : 0) + (BWD_ACT != IB_ACT_NONE ? 8 : 0) + (HAS_ADD ? 8 : 0);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.kchunk / BK;                              // >= 4 (host check)
  const int tiles = p.tiles_mn * p.splits;
  const int nwg = (int)gridDim.x;
  // XCD-aware walk: workgroups b and b + 8 share an XCD (round-robin dispatch) -> give each XCD a contiguous run of
  // the logical tile order (column tiles of one row panel are neighbours in it).  Speed only.
  const int first_tile = ib_xcd_remap((int)blockIdx.x, nwg);          // bijective for every grid size

  // per-lane constants of the fragment reads (the only lane-derived values kept live across the loops)
  const unsigned smem0 = lds_off(smem);
  unsigned a_base0, b_base0;
  {
    const int lane = lane_now();
    const unsigned fragx = (unsigned)((((lane >> 4) ^ (lane & 7)) << 4));
    a_base0 = smem0 + (unsigned)((wm * 64 + (lane & 15)) * 128) + fragx;
    b_base0 = smem0 + A_BYTES + (unsigned)((wn * 64 + (lane & 15)) * 128) + fragx;
  }

  // staging sources per lane as 32-bit ELEMENT offsets from A / B (the host checks they fit): this tile's, the next tile's
  unsigned pa[4], pb[2];
  auto tile_ptrs = [&](int tile, unsigned (&qa)[4], unsigned (&qb)[2]) {
    const int tmn = tile % p.tiles_mn;
    const int i0 = (tmn / p.tiles_n) * BM, j0 = (tmn % p.tiles_n) * BN;
    const unsigned kofs = (unsigned)((tile / p.tiles_mn) * p.kchunk);       // the split's first k
    const int lane = lane_now();
    const int srow = lane >> 3;                             // row inside an 8-row chunk
    const int spc = (lane & 7) ^ srow;                      // the 16-byte piece of that row this lane fetches
#pragma unroll
    for (int j = 0; j < 4; ++j)
      qa[j] = (unsigned)min(i0 + 8 * (wave + 8 * j) + srow, p.M - 1) * (unsigned)p.lda + 8u * spc + kofs;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      qb[j] = (unsigned)min(j0 + 8 * (wave + 8 * j) + srow, p.N - 1) * (unsigned)p.ldb + 8u * spc + kofs;
  };
  // LDS-DMA piece j (0..3: A chunks, 4..5: B chunks) of a stage: k offset k0 (elements), slot base `st`
  auto piece = [&](const unsigned (&qa)[4], const unsigned (&qb)[2], int j, int k0, unsigned char* st) {
    if (j < 4)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.A + (size_t)(qa[j] + (unsigned)k0)), (lds_void_t*)(st + (wave + 8 * j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.B + (size_t)(qb[j - 4] + (unsigned)k0)), (lds_void_t*)(st + A_BYTES + (wave + 8 * (j - 4)) * 1024), 16, 0, 0);
  };
  auto read_frags = [&](unsigned so, int ks, bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
    const unsigned aa = (a_base0 + so) ^ (ks ? 64u : 0u), bb = (b_base0 + so) ^ (ks ? 64u : 0u);
    fa[0] = lds_read16<0>(aa); fa[1] = lds_read16<2048>(aa); fa[2] = lds_read16<4096>(aa); fa[3] = lds_read16<6144>(aa);
    fb[0] = lds_read16<0>(bb); fb[1] = lds_read16<2048>(bb); fb[2] = lds_read16<4096>(bb); fb[3] = lds_read16<6144>(bb);
  };

  if (first_tile >= tiles) return;
  tile_ptrs(first_tile, pa, pb);
  int slot = 0;                                              // LDS slot of the stage the next K step consumes
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < 6; ++j) piece(pa, pb, j, s * BK, smem + s * STAGE);
  bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
  wait_vm<12>();                                             // stage 0 landed (mine)
  __builtin_amdgcn_s_barrier();
  bool fresh = true;                                         // no epilogue stores are outstanding (first tile)
  int round = 0;
  NT_STAMP(0);

  for (int tile = first_tile; tile < tiles; tile += nwg) {
    const int tmn = tile % p.tiles_mn;
    const int i0 = (tmn / p.tiles_n) * BM, j0 = (tmn % p.tiles_n) * BN;
    const bool has_next = tile + nwg < tiles;
    const bool ragged = (i0 + BM > p.M) || (j0 + BN > p.N);

    f32x4_t acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // one K step; ISSUE: 0 = nothing, 1 = stage kt + 3 of this tile, 2 = stage kt + 3 - nk of the next tile;
    // WAIT: operations that may stay in flight when stage g + 1 must have landed (-1: there is no stage g + 1)
    auto step = [&](auto issue_c, auto wait_c, auto read_c, int kt) {
      constexpr int ISSUE = decltype(issue_c)::value, WAIT = decltype(wait_c)::value;
      constexpr bool READ = decltype(read_c)::value != 0;     // prefetch the first fragments of stage g + 1
      [[maybe_unused]] unsigned pn_a[4], pn_b[2];
      const unsigned so = (unsigned)(slot * STAGE);
      read_frags(so, 1, fa1, fb1);                           // second half of stage g
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[u], fa0[t], acc[t][u], 0, 0, 0);
      frags_ready(fa1, fb1);                                 // my last read of stage g is done
      if constexpr (WAIT >= 0) wait_vm<WAIT>();              // stage g + 1 landed (mine)
      __builtin_amdgcn_s_barrier();                          // everyone's stage g + 1 landed; everyone left stage g
      const int slot1 = slot == NS - 1 ? 0 : slot + 1;
      if constexpr (READ) read_frags((unsigned)(slot1 * STAGE), 0, fa0, fb0);          // first half of stage g + 1
      unsigned char* st = smem + so;                         // stage g + 3 -> the slot of stage g
      const int k0 = ISSUE == 1 ? (kt + 3) * BK : (kt + 3 - nk) * BK;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[u], fa1[t], acc[t][u], 0, 0, 0);
        if constexpr (ISSUE == 1) {
          piece(pa, pb, t, k0, st);
          if (t < 2) piece(pa, pb, 4 + t, k0, st);
        } else if constexpr (ISSUE == 2) {
          if (t == 0) tile_ptrs(tile + nwg, pn_a, pn_b);     // recomputed (a few VALU ops) rather than kept live
          piece(pn_a, pn_b, t, k0, st);
          if (t < 2) piece(pn_a, pn_b, 4 + t, k0, st);
        }
      }
      if constexpr (READ) frags_ready(fa0, fb0);
      slot = slot1;
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
    using W6 = std::integral_constant<int, 6>;
    using W0 = std::integral_constant<int, 0>; using WN = std::integral_constant<int, -1>;
    // the epilogue's operands: bias (per column group of the lane's accumulators), activation-derivative operand and
    // residual addend (per 16-byte piece the thread will store) -- requested before the LAST K step
    [[maybe_unused]] bf16x8_t xaux[8], xadd[8];
    [[maybe_unused]] float4 b4[4];
    auto epilogue_loads = [&]() {
      const int lane = lane_now();
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          b4[u] = *reinterpret_cast<const float4*>(p.bias + min(j0 + wn * 64 + 4 * (lane >> 4) + 16 * u, p.N - 4));
      }
      if constexpr (BWD_ACT != IB_ACT_NONE || HAS_ADD) {
        const int tid = (wave << 6) | lane;
        const int ncol = min(j0 + 8 * (tid & 15), p.N - 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int m = min(i0 + (tid >> 4) + 32 * q, p.M - 1);
          if constexpr (BWD_ACT != IB_ACT_NONE) xaux[q] = *reinterpret_cast<const bf16x8_t*>(p.aux + (int64_t)m * p.ldaux + ncol);
          if constexpr (HAS_ADD) xadd[q] = *reinterpret_cast<const bf16x8_t*>(p.addend + (int64_t)m * p.ldadd + ncol);
        }
      }
    };
    using WL = std::integral_constant<int, 6 + EPI_LOADS>;
    // first step: the eight stores of the previous tile's epilogue (uniform per thread unless that tile was ragged) may
    // stay in flight behind stage 1
    // (the last step makes the next tile's stage 0 visible but leaves its fragments to the next tile: the epilogue needs
    // the registers)
    read_frags((unsigned)(slot * STAGE), 0, fa0, fb0);
    frags_ready(fa0, fb0);
    using WF = std::integral_constant<int, SLAB ? 22 : 14>;    // + the previous tile's stores: 8 (16-byte bf16 pieces) or 16 (fp32 slab)
    if (fresh) step(C1{}, W6{}, C1{}, 0); else step(C1{}, WF{}, C1{}, 0);
    for (int kt = 1; kt < nk - 3; ++kt) step(C1{}, W6{}, C1{}, kt);
    if (has_next) {
      step(C2{}, W6{}, C1{}, nk - 3); step(C2{}, W6{}, C1{}, nk - 2);
      epilogue_loads();
      step(C0{}, WL{}, C0{}, nk - 1);                        // issues nothing: its slot becomes the C image
    } else {
      step(C0{}, W6{}, C1{}, nk - 3); step(C0{}, W0{}, C1{}, nk - 2);
      epilogue_loads();
      step(C0{}, WN{}, C0{}, nk - 1);
    }

    NT_STAMP(1 + 2 * round);
    // ---- epilogue: the C image (two halves of 128 rows) goes through the slot of the tile's last stage -- every wave
    // left that stage at the barrier of the last step.  All LDS traffic is inline asm (LDS accesses the compiler can see
    // would be ordered behind the in-flight LDS-DMA of the next tile with s_waitcnt vmcnt(0)).
    const int cslot = slot == 0 ? NS - 1 : slot - 1;          // the slot of the tile's last stage
    if constexpr (SLAB) {
      // split-K: the fp32 accumulators go straight to this split's slab (a lane holds 4 consecutive columns of one row:
      // 16-byte stores); no LDS, no barrier -- the stage slots keep streaming
      const int lane = lane_now();
      float* sl = p.slab + (size_t)(tile / p.tiles_mn) * (size_t)p.M * (size_t)p.N;
      const int m0 = i0 + wm * 64 + (lane & 15), n0 = j0 + wn * 64 + 4 * (lane >> 4);
      if (!ragged) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            *reinterpret_cast<f32x4_t*>(sl + (size_t)(m0 + 16 * t) * p.N + n0 + 16 * u) = acc[t][u];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (m0 + 16 * t < p.M && n0 + 16 * u < p.N)
              *reinterpret_cast<f32x4_t*>(sl + (size_t)(m0 + 16 * t) * p.N + n0 + 16 * u) = acc[t][u];
      }
    } else {
    const unsigned cimg = smem0 + (unsigned)(cslot * STAGE);
    const int lane = lane_now();
    const int tid = (wave << 6) | lane;
    const int pc = tid & 15, r0 = tid >> 4;                  // store phase: piece pc of rows r0 + 32 q of a half
    const int ncol = min(j0 + 8 * pc, p.N - 8);              // column pieces beyond N are clamped onto the last one (ragged
                                                             // tiles only; their duplicate stores are skipped below)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((wave >> 2) == h) {
        const unsigned wbase = cimg + (unsigned)((((wm & 1) * 64 + (lane & 15)) * CS) + (wn * 64 + 4 * (lane >> 4)) * 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          auto put = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            bf16x4_t o;
            float bb[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS) { bb[0] = b4[u].x; bb[1] = b4[u].y; bb[2] = b4[u].z; bb[3] = b4[u].w; }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)nt_act<FWD_ACT>(acc[t][u][r] + bb[r]);
            const u32x2_t raw = __builtin_bit_cast(u32x2_t, o);
            if (t == 0) lds_write8<0 * 16 * CS + u * 32>(wbase, raw);
            else if (t == 1) lds_write8<1 * 16 * CS + u * 32>(wbase, raw);
            else if (t == 2) lds_write8<2 * 16 * CS + u * 32>(wbase, raw);
            else lds_write8<3 * 16 * CS + u * 32>(wbase, raw);
          };
          put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{});
          put(std::integral_constant<int, 2>{}); put(std::integral_constant<int, 3>{});
        }
      }
      lds_drain();
      __builtin_amdgcn_s_barrier();                          // the half image is complete
      bf16x8_t v[4];
      const unsigned rbase = cimg + (unsigned)(r0 * CS + pc * 16);
      v[0] = lds_read16<0 * 32 * CS>(rbase); v[1] = lds_read16<1 * 32 * CS>(rbase);
      v[2] = lds_read16<2 * 32 * CS>(rbase); v[3] = lds_read16<3 * 32 * CS>(rbase);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      __builtin_amdgcn_s_barrier();                          // every thread has its pieces: the image may be overwritten
      bf16x8_t vout[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x8_t o = v[q];
        if constexpr (BWD_ACT != IB_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] * nt_act_bwd<BWD_ACT>((float)xaux[4 * h + q][e]));
        }
        if constexpr (HAS_ADD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] + (float)xadd[4 * h + q][e]);
        }
        vout[q] = o;
      }
      if (!ragged) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<bf16x8_t*>(p.C + (int64_t)(i0 + 128 * h + r0 + 32 * q) * p.ldc + ncol) = vout[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = i0 + 128 * h + r0 + 32 * q;
          if (m < p.M && j0 + 8 * pc < p.N) *reinterpret_cast<bf16x8_t*>(p.C + (int64_t)m * p.ldc + ncol) = vout[q];
        }
      }
    }
    }   // !SLAB
    NT_STAMP(2 + 2 * round);
    ++round;
    fresh = ragged;          // after a ragged tile the store count per thread is not uniform: the next wait is conservative
    if (has_next) {
      // stage 2 of the next tile into the slot the C image occupied (every thread passed the last barrier with its
      // pieces in registers)
      tile_ptrs(tile + nwg, pa, pb);
#pragma unroll
      for (int j = 0; j < 6; ++j) piece(pa, pb, j, 2 * BK, smem + cslot * STAGE);
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int FWD_ACT, int BWD_ACT>
int launch(const NtParams& p, hipStream_t s) {
  const int tiles = p.tiles_m * p.tiles_n;
  const int grid = tiles < 256 ? tiles : 256;                 // one persistent workgroup per CU
  if constexpr (BWD_ACT != IB_ACT_NONE) {                     // backward epilogues never carry a bias
    if (p.addend) hipLaunchKernelGGL((gemm_nt_kernel<FWD_ACT, BWD_ACT, true, false>), dim3(grid), dim3(NT_THREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<FWD_ACT, BWD_ACT, false, false>), dim3(grid), dim3(NT_THREADS), 0, s, p);
  } else if (p.addend) {
    if constexpr (FWD_ACT == IB_ACT_NONE) {
      if (p.bias) return IB_E_UNSUPPORTED;
      hipLaunchKernelGGL((gemm_nt_kernel<IB_ACT_NONE, IB_ACT_NONE, true, false>), dim3(grid), dim3(NT_THREADS), 0, s, p);
    } else {
      return IB_E_UNSUPPORTED;
    }
  } else if (p.bias) {
    hipLaunchKernelGGL((gemm_nt_kernel<FWD_ACT, IB_ACT_NONE, false, true>), dim3(grid), dim3(NT_THREADS), 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_nt_kernel<FWD_ACT, IB_ACT_NONE, false, false>), dim3(grid), dim3(NT_THREADS), 0, s, p);
  }
  IB_CHECK_LAUNCH();
  return IB_OK;
}

}  // namespace

// Returns IB_E_UNSUPPORTED when the problem does not qualify (the caller then takes the generic kernels).
int ib_gemm_nt_try(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, int fwd_act,
                   const void* aux, int64_t ldaux, int bwd_act, const void* addend, int64_t ldadd, int64_t M, int64_t N,
                   int64_t K, hipStream_t s) {
  static const int off = ib_ab_int("IB_NO_NT", 0);
  static const int min_m = ib_ab_int("IB_NT_MIN_M", 640);
  if (off || M < min_m || N < 128 || N % 8 != 0 || K < 4 * BK || K % BK != 0) return IB_E_UNSUPPORTED;
  if (!al16(A) || !al16(B) || !al16(C) || lda % 8 || ldb % 8 || ldc % 8) return IB_E_UNSUPPORTED;
  if (M * lda >= (int64_t(1) << 31) || N * ldb >= (int64_t(1) << 31)) return IB_E_UNSUPPORTED;     // 32-bit element offsets
  if (bias && !al16(bias)) return IB_E_UNSUPPORTED;
  if (aux && (!al16(aux) || ldaux % 8)) return IB_E_UNSUPPORTED;
  if (addend && (!al16(addend) || ldadd % 8)) return IB_E_UNSUPPORTED;
  if (bwd_act != IB_ACT_NONE && (bwd_act == IB_ACT_SILU || !aux || fwd_act != IB_ACT_NONE)) return IB_E_UNSUPPORTED;
  IB_PATH(IB_PATH_NT);
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.C = (bf16_t*)C; p.ldc = ldc; p.bias = bias; p.aux = (const bf16_t*)aux; p.ldaux = ldaux;
  p.addend = (const bf16_t*)addend; p.ldadd = ldadd;
  p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
  p.tiles_mn = p.tiles_m * p.tiles_n; p.splits = 1; p.kchunk = (int)K; p.slab = nullptr;
  p.prof = IB_AB_PROF(g_nt_prof);
  if (bwd_act != IB_ACT_NONE) {
    switch (bwd_act) {
      case IB_ACT_RELU: return launch<IB_ACT_NONE, IB_ACT_RELU>(p, s);
      case IB_ACT_TANH: return launch<IB_ACT_NONE, IB_ACT_TANH>(p, s);
      case IB_ACT_SIGMOID: return launch<IB_ACT_NONE, IB_ACT_SIGMOID>(p, s);
      case IB_ACT_ELU: return launch<IB_ACT_NONE, IB_ACT_ELU>(p, s);
      default: return IB_E_UNSUPPORTED;
    }
  }
  switch (fwd_act) {
    case IB_ACT_NONE: return launch<IB_ACT_NONE, IB_ACT_NONE>(p, s);
    case IB_ACT_RELU: return launch<IB_ACT_RELU, IB_ACT_NONE>(p, s);
    case IB_ACT_TANH: return launch<IB_ACT_TANH, IB_ACT_NONE>(p, s);
    case IB_ACT_SIGMOID: return launch<IB_ACT_SIGMOID, IB_ACT_NONE>(p, s);
    case IB_ACT_SILU: return launch<IB_ACT_SILU, IB_ACT_NONE>(p, s);
    case IB_ACT_ELU: return launch<IB_ACT_ELU, IB_ACT_NONE>(p, s);
    default: return IB_E_UNSUPPORTED;
  }
}

// Split-K form for GEMMs with few output tiles and a long reduction (the sampler's FFN output projection at a few thousand
// rows: 52 tiles x 32 K steps): slab[s][M][N] (fp32) = A[:, s-th k range] B[:, s-th k range]^T, `splits` = the slab count
// ib_gemm_nt_splitk_splits() names.  The caller's reduction (the slab LayerNorm) sums the slabs in order.
int ib_gemm_nt_splitk_splits(int64_t M, int64_t N, int64_t K) {
  static const int off = ib_ab_int("IB_NO_NT_SPLITK", 0);
  static const int min_m = ib_ab_int("IB_NT_MIN_M", 640);
  if (off || M < min_m || N < 128 || N % 8 != 0 || K % BK != 0) return 0;
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int best = 0;
  for (int sp = 2; sp <= 16; ++sp) {                     // the most splits that still give one item per CU, >= 4 K steps each
    if (K % (sp * BK) != 0 || K / sp < 4 * BK) continue;
    if (tiles * sp <= 256) best = sp;
  }
  return best;
}
int ib_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* slab, int splits, int64_t M, int64_t N,
                      int64_t K, hipStream_t s) {
  if (splits < 2 || splits != ib_gemm_nt_splitk_splits(M, N, K)) return IB_E_UNSUPPORTED;
  if (!al16(A) || !al16(B) || !al16(slab) || lda % 8 || ldb % 8) return IB_E_UNSUPPORTED;
  if (M * lda >= (int64_t(1) << 31) || N * ldb >= (int64_t(1) << 31)) return IB_E_UNSUPPORTED;
  IB_PATH(IB_PATH_NT_SPLITK);
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
  p.tiles_mn = p.tiles_m * p.tiles_n; p.splits = splits; p.kchunk = (int)(K / splits); p.slab = slab;
  p.prof = IB_AB_PROF(g_nt_prof);
  const int tiles = p.tiles_mn * splits;
  hipLaunchKernelGGL((gemm_nt_kernel<IB_ACT_NONE, IB_ACT_NONE, false, false, true>), dim3(tiles < 256 ? tiles : 256),
                     dim3(NT_THREADS), 0, s, p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// TIMING-ONLY: device buffer of [workgroups][16] int64 stamps filled by the next NT GEMM launches (NULL = off)
#ifdef IB_AB
extern "C" int ib_debug_set_nt_prof(void* buf) { g_nt_prof = reinterpret_cast<long long*>(buf); return IB_OK; }
#else
extern "C" int ib_debug_set_nt_prof(void*) { return IB_E_UNSUPPORTED; }       // measurement builds only
#endif

// ---- dst_i[c][r] = src_i[r][c] for several bf16 matrices in ONE launch (the transposed weight copies the backward
// GEMMs read k-contiguously; refreshed once per step after the optimizer moved the weights).  64 x 64 tiles through LDS.
namespace {
constexpr int TR_MAX = 32;
struct TrMulti { const bf16_t* src[TR_MAX]; bf16_t* dst[TR_MAX]; int rows[TR_MAX], cols[TR_MAX], lds[TR_MAX], ldd[TR_MAX], blk0[TR_MAX + 1]; int n; unsigned vec; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(TrMulti m) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];      // 144-byte rows: 16-byte row writes stay aligned
  int e = 0;
  for (int j = 1; j < m.n; ++j)
    if ((int)blockIdx.x >= m.blk0[j]) e = j;
  const int b = (int)blockIdx.x - m.blk0[e];
  const int R = m.rows[e], Cc = m.cols[e];
  const int tc = (Cc + 63) / 64;
  const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
  const bf16_t* src = m.src[e];
  bf16_t* dst = m.dst[e];
  if ((m.vec >> e) & 1u) {
    // whole 64 x 64 tiles of 16-byte aligned matrices (every layer weight): 16-byte global accesses on both sides -- two
    // loads and two stores per thread instead of sixteen 2-byte ones each way (the launch was 21.7 us for 52 MB)
    const int pr = threadIdx.x >> 3, pc = threadIdx.x & 7;          // row 0..31 (+32), 16-byte piece 0..7
    uint4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
      v[h] = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + pr + 32 * h) * m.lds[e] + c0 + 8 * pc);
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<uint4*>(&tile[pr + 32 * h][8 * pc]) = v[h];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = pr + 32 * h;                                    // destination row = source column
      bf16x8_t o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = tile[8 * pc + k][c];
      *reinterpret_cast<bf16x8_t*>(dst + (int64_t)(c0 + c) * m.ldd[e] + r0 + 8 * pc) = o;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    tile[r][tx] = (r0 + r < R && c0 + tx < Cc) ? src[(int64_t)(r0 + r) * m.lds[e] + c0 + tx] : (bf16_t)0.f;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < Cc && r0 + tx < R) dst[(int64_t)(c0 + c) * m.ldd[e] + r0 + tx] = tile[tx][c];
}
}  // namespace

extern "C" int ib_transpose_multi(int n, const void* const* src, const int64_t* lds, void* const* dst, const int64_t* ldd,
                                  const int64_t* rows, const int64_t* cols, int dtype, ib_stream_t stream) {
  if (n <= 0 || n > TR_MAX || !src || !dst || !lds || !ldd || !rows || !cols) return IB_E_ARG;
  if (dtype != IB_BF16) return IB_E_DTYPE;
  TrMulti m{};
  m.n = n;
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0 || lds[i] < cols[i] || ldd[i] < rows[i]) return IB_E_ARG;
    m.src[i] = (const bf16_t*)src[i]; m.dst[i] = (bf16_t*)dst[i];
    m.rows[i] = (int)rows[i]; m.cols[i] = (int)cols[i]; m.lds[i] = (int)lds[i]; m.ldd[i] = (int)ldd[i];
    m.blk0[i] = blk;
    if (rows[i] % 64 == 0 && cols[i] % 64 == 0 && lds[i] % 8 == 0 && ldd[i] % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(src[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst[i]) & 15) == 0)
      m.vec |= 1u << i;
    blk += (int)(((rows[i] + 63) / 64) * ((cols[i] + 63) / 64));
  }
  m.blk0[n] = blk;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(blk), dim3(256), 0, ib_s(stream), m);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
