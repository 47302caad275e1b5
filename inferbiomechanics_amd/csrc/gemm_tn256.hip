// Weight-gradient GEMM with a 256 (n) x 256 (k) output tile, bf16, gfx950: dW[N,K] = sum_m dz[m][n] * x[m][k] as fp32
// split slabs, for groups of problems whose N and K are multiples of 256 (the transformer denoiser's layers: [1536,512],
// [512,512], [2048,512], [512,2048] at M = 12800 token rows).
//
// Why beside gemm_tn.hip: that kernel's K step is bound by what one CU takes in from L2 (48 KB per 64-deep step = 0.82 us
// at ~60 GB/s per CU, against 0.49 us of MFMA issue).  A 256 x 256 tile takes in 64 KB per 64 reduction rows for TWICE the
// flops: ingest and MFMA time meet (~1.0 us per 64 rows), i.e. ~1.5x the rate per CU.  The second thing it fixes is the
// grouped launch's balance: gemm_tn.hip gives every problem its own split count (2, 8, 2, 2 for a transformer layer), so
// the out-projection's items are 25 K steps long and the others 100, one item per CU -- the launch lasts 100 steps for 75
// steps of work per CU.  Here every problem of the group takes the SAME split count, so all items are equally long.
//
// Layout of a stage (32 reduction rows): A image [32][256 n] and B image [32][256 k], 512-byte rows, filled by
// global_load_lds_dwordx4 (two rows per 1-KiB wave instruction); the sixteen 32-byte column chunks of a row are
// XOR-swizzled with s(m) = (m & 3) | ((m >> 3) & 1) << 2 on the per-lane SOURCE address (LDS-DMA writes lane-linearly), so
// the ds_read_b64_tr_b16 fragment reads are conflict-free (same scheme as gemm_tn.hip).  Four stages of 32 KiB: one being
// read, three in flight.  Eight waves as 4 (n) x 2 (k): a wave owns 64 n x 128 k = 4 x 8 MFMA 16x16x32 tiles.
// A step = one stage = 32 MFMAs per wave in two groups (k tiles 0-3, 4-7): the second group's B fragments are read while
// the first group runs, the next stage's A / first-half B fragments while the second runs; one barrier per step, between
// the groups.  The stage stream is CONTINUOUS over a workgroup's items (persistent workgroups, XCD-aware walk).
// No ragged tiles (the host refuses them): every address is in range by construction.
#include <type_traits>

#include "ib_common.h"
#include "gemm_nt.h"

namespace {

constexpr int TM = 256, TK = 256, BK = 32, NS = 4, THREADS = 512;
constexpr int ROW = 512;                              // bytes per image row (256 bf16), both operands
constexpr int OP_BYTES = BK * ROW, STAGE = 2 * OP_BYTES, LDS_BYTES = NS * STAGE;      // 16 KiB, 32 KiB, 128 KiB
constexpr int MAXP = 6;
#ifndef TN256_ABL
#define TN256_ABL 0      // TIMING-ONLY build variants (tools/build_variant.sh): 1 no MFMA, 2 no LDS-DMA after the prologue, 3 no fragment reads, 4 neither (four-wave form)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// one MFMA fragment = reduction rows m .. m+3 (lo) and m+4 .. m+7 (hi) of a 16-column chunk, transposed by the LDS
__device__ __forceinline__ bf16x8_t lds_read_tr(unsigned addr) {
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * ROW));
  u32x4_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void frags_ready(bf16x8_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}
__device__ __forceinline__ void frags_ready(bf16x8_t (&a)[4], bf16x8_t (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Prob {
  const bf16_t* A; const bf16_t* B; int lda, ldb;        // dz [M, N], x [M, K]
  int N, K;
  float* C;                                               // slabs [splits][N][K]
  float* dbias;                                           // optional [splits][N]
  int tiles_k, per_split, item0;                          // K / 256 ; tiles per split ; first work item
};
struct Params {
  Prob pr[MAXP]; int n, items;
  int splits, chunk, nk;             // common to the group: reduction rows per split, stages (of 32 rows) per item
};

template <bool BIAS>
__global__ __launch_bounds__(THREADS, 2) void gemm_tn256_kernel(Params P) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // waves w and w + 4 share a SIMD: they take the two k halves of the same n rows, so the bias MFMAs (wn == 0 only) load
  // every SIMD equally
  const int wm = wave & 3, wn = wave >> 2;
  const int nwg = (int)gridDim.x;
  const int first = ib_xcd_remap((int)blockIdx.x, nwg);
  if (first >= P.items) return;
  const unsigned smem0 = lds_off(smem);
  const int nk = P.nk;

  // fragment read bases (see the header: chunk = (wave's first chunk + tile) ^ s(row); the tile index only XORs bits 5..)
  unsigned ab0, bb0;
  {
    const int lane = lane_now();
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    ab0 = smem0 + (unsigned)((8 * g + q) * ROW + 32 * (((wm ^ (g & 1)) << 2) | q) + 8 * pq);
    bb0 = smem0 + OP_BYTES + (unsigned)((8 * g + q) * ROW + 32 * ((wn << 3) | ((g & 1) << 2) | q) + 8 * pq);
  }
  auto read_a = [&](unsigned so, bf16x8_t (&fa)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[t] = lds_read_tr((ab0 + so) ^ (unsigned)(t << 5));
  };
  auto read_b = [&](unsigned so, int half, bf16x8_t (&fb)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) fb[u] = lds_read_tr((bb0 + so) ^ (unsigned)((4 * half + u) << 5));
  };

  // ---- the issue side of the stage stream: (item, stage of the item) -> four LDS-DMA instructions per wave
  struct Src { const bf16_t* A; const bf16_t* B; unsigned qa, qb, sa, sb; };   // per-lane element offsets; row strides
  auto locate = [&](int item, Src& s) {
    int e = 0;
    for (int j = 1; j < P.n; ++j)
      if (item >= P.pr[j].item0) e = j;
    const Prob& q = P.pr[e];
    const int local = item - q.item0;
    const int split = local / q.per_split, tl = local % q.per_split;
    const int i0 = (tl / q.tiles_k) * TM, j0 = (tl % q.tiles_k) * TK;
    const int lane = lane_now();
    const int r = 2 * wave + (lane >> 5);                   // row of the stage, modulo 16 (the second piece adds 16)
    const int sw = ((r & 3) | (((r >> 3) & 1) << 2)) << 1;  // swizzle of that row, in 16-byte pieces
    const int src = (lane & 31) ^ sw;
    s.A = q.A; s.B = q.B;
    s.sa = (unsigned)q.lda; s.sb = (unsigned)q.ldb;
    s.qa = (unsigned)(split * P.chunk + r) * (unsigned)q.lda + (unsigned)(i0 + 8 * src);
    s.qb = (unsigned)(split * P.chunk + r) * (unsigned)q.ldb + (unsigned)(j0 + 8 * src);
  };
  // piece j (0, 1: A rows r, r + 16; 2, 3: B rows r, r + 16) of stage kt of the item behind `s` into stage slot `st`
  auto piece = [&](const Src& s, int j, int kt, unsigned char* st) {
    if (j < 2)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(s.A + (size_t)s.qa + (size_t)((unsigned)(kt * BK + 16 * j) * s.sa)),
                                       (lds_void_t*)(st + (wave + 8 * j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_void_t*)(s.B + (size_t)s.qb + (size_t)((unsigned)(kt * BK + 16 * (j - 2)) * s.sb)),
                                       (lds_void_t*)(st + OP_BYTES + (wave + 8 * (j - 2)) * 1024), 16, 0, 0);
  };
  Src isrc;
  int iss_item = first, iss_kt = 0;      // next stage to issue
  bool iss_more = true;
  int issued = 0;                        // stages issued so far (the stage stream's index)
  locate(iss_item, isrc);
  auto advance = [&]() {                 // after a stage's four pieces have been issued
    ++issued;
    if (++iss_kt == nk) {
      iss_kt = 0;
      iss_item += nwg;
      iss_more = iss_item < P.items;
      if (iss_more) locate(iss_item, isrc);
    }
  };

  // prologue: all four slots (nk >= NS by the host's choice of the split)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) piece(isrc, j, iss_kt, smem + s * STAGE);
    advance();
  }
  wait_vm<12>();
  __builtin_amdgcn_s_barrier();
  bf16x8_t fa0[4], fa1[4], fbl[4], fbh[4];
  read_a(0u, fa0);
  read_b(0u, 0, fbl);
  frags_ready(fa0, fbl);
  [[maybe_unused]] bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  int slot = 0;
  int consumed = 0;                      // stages consumed so far
  int post_epi = 0;                      // steps left in which the previous item's 32 slab stores may still be in flight

  for (int item = first; item < P.items; item += nwg) {
    int e = 0;
    for (int j = 1; j < P.n; ++j)
      if (item >= P.pr[j].item0) e = j;
    const Prob& q = P.pr[e];
    const int local = item - q.item0;
    const int split = local / q.per_split, tl = local % q.per_split;
    const int i0 = (tl / q.tiles_k) * TM, j0 = (tl % q.tiles_k) * TK;

    [[maybe_unused]] const bool want_bias = BIAS && q.dbias != nullptr && j0 == 0 && wn == 0;
    f32x4_t acc[4][8];
    [[maybe_unused]] f32x4_t accb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (BIAS) accb[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // one step; FA: the A fragments of this stage, FN: the buffer that receives the next stage's
    auto step = [&](bf16x8_t (&FA)[4], bf16x8_t (&FN)[4]) {
      const unsigned so = (unsigned)(slot * STAGE);
      if (TN256_ABL != 3) read_b(so, 1, fbh);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (TN256_ABL != 1) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbl[u], FA[t], acc[t][u], 0, 0, 0);
      if constexpr (BIAS) {
        if (want_bias) {             // wave- and item-uniform: only the k = 0 tiles of a problem with a bias, only the wn = 0 waves
#pragma unroll
          for (int t = 0; t < 4; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, FA[t], accb[t], 0, 0, 0);
        }
      }
      frags_ready(fbh);
      // The next stage must have landed before the barrier.  vmcnt retires in order: the operations younger than its four
      // pieces are the stages issued behind it (4 each) and, for three steps after an epilogue, that item's 32 slab stores
      // per thread (bias stores only add to that: waiting for more is safe).  When the stream is draining (the workgroup's
      // last steps) everything is waited for.
      const int ahead = issued - consumed - 1;     // stages issued beyond the one being consumed (wave-uniform)
      if (ahead >= 3) { if (post_epi > 0) wait_vm<40>(); else wait_vm<8>(); }
      else wait_vm<0>();
      if (post_epi > 0) --post_epi;
      __builtin_amdgcn_s_barrier();                // every wave has read this stage's fragments: its slot is free
      const int slot1 = slot == NS - 1 ? 0 : slot + 1;
      if (ahead > 0 && TN256_ABL != 3) { read_a((unsigned)(slot1 * STAGE), FN); read_b((unsigned)(slot1 * STAGE), 0, fbl); }
      unsigned char* st = smem + so;
      // (one branch around all four pieces, with the MFMAs in both arms, was tried: the accumulators' phi copies spilled -- 256
      // VGPRs + 108 B of scratch, the launch 105 -> 233 us; the four short wave-uniform branches stay)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (TN256_ABL != 1) acc[t][4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbh[u], FA[t], acc[t][4 + u], 0, 0, 0);
        if (iss_more && TN256_ABL != 2) piece(isrc, t, iss_kt, st);  // stage consumed + NS goes into the slot just freed
      }
      if (iss_more) advance();
      if (ahead > 0) frags_ready(FN, fbl);
      slot = slot1;
      ++consumed;
    };
    for (int kt = 0; kt + 1 < nk; kt += 2) { step(fa0, fa1); step(fa1, fa0); }      // nk is even (host)

    // ---- epilogue: fp32 accumulators straight to the slab of this split (the stage slots keep streaming)
    {
      const int lane = lane_now();
      float* slab = q.C + (size_t)split * (size_t)q.N * (size_t)q.K;
      const int n0 = i0 + wm * 64 + (lane & 15), k0 = j0 + wn * 128 + 4 * (lane >> 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          *reinterpret_cast<f32x4_t*>(slab + (size_t)(n0 + 16 * t) * q.K + k0 + 16 * u) = acc[t][u];
      if constexpr (BIAS) {
        if (want_bias && (lane >> 4) == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) q.dbias[(size_t)split * q.N + n0 + 16 * t] = accb[t][0];
        }
      }
    }
    post_epi = 3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave form (round 5): the same 256 x 256 tile and stage ring, 256 threads as 2 (n) x 2 (k) waves of 128 x 128 =
// 8 x 8 MFMA tiles (256 accumulator registers per lane: one wave per SIMD, the 512-register budget).
// Why: the eight-wave form reads (64 + 128) x 32 x 2 B = 12 KiB of fragments per wave and stage = 96 KiB per workgroup
// against 32 KiB that the LDS-DMA writes -- 1024 LDS cycles (128 B / clk) per stage next to 1024 MFMA cycles: the two pipes
// could only lose to each other (timing-only ablations: no fragment reads 84 us, no LDS-DMA 86 us, both 102 us).  A
// 128 x 128 wave tile reads 16 KiB per wave = 64 KiB per workgroup and stage: 768 LDS cycles under the same 1024 MFMA cycles.
// A stage's sixteen fragments are held whole (64 registers) and the NEXT stage's sixteen are read under the 64 MFMAs, so
// the stage's slot is free from the barrier at the head of its own step: one barrier per step, at the step boundary.
struct Frags { bf16x8_t a[8], b[8]; };
__device__ __forceinline__ void frags_ready(Frags& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.a[4]), "+v"(f.a[5]), "+v"(f.a[6]), "+v"(f.a[7]),
                 "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5]), "+v"(f.b[6]), "+v"(f.b[7]));
}
constexpr int THREADS4 = 256;
#ifndef TN256_NS4
#define TN256_NS4 4
#endif
constexpr int NS4 = TN256_NS4;          // ring depth of the four-wave form (5 x 32 KiB = the whole LDS measured the same: 104.9-108.5 vs 106.1-107.9 us)
// accumulators pinned to the AccVGPR half of the wave's 512 registers (left to itself the allocator spreads them over both
// halves and copies: 179 v_accvgpr moves per 128 MFMAs); volatile: issue order = program order
__device__ __forceinline__ void mfma_acc(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <bool BIAS>
__global__ __launch_bounds__(THREADS4, 1) void gemm_tn256w4_kernel(Params P) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS4 * STAGE];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int nwg = (int)gridDim.x;
  const int first = ib_xcd_remap((int)blockIdx.x, nwg);
  if (first >= P.items) return;
  const unsigned smem0 = lds_off(smem);
  const int nk = P.nk;

  unsigned ab0, bb0;
  {
    const int lane = lane_now();
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const unsigned ro = (unsigned)((8 * g + q) * ROW + 8 * pq);
    ab0 = smem0 + ro + (unsigned)(32 * ((wm << 3) | ((g & 1) << 2) | q));
    bb0 = smem0 + OP_BYTES + ro + (unsigned)(32 * ((wn << 3) | ((g & 1) << 2) | q));
  }
  [[maybe_unused]] bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  for (int item = first; item < P.items; item += nwg) {
    int e = 0;
    for (int j = 1; j < P.n; ++j)
      if (item >= P.pr[j].item0) e = j;
    const Prob& q = P.pr[e];
    const int local = item - q.item0;
    const int split = local / q.per_split, tl = local % q.per_split;
    const int i0 = (tl / q.tiles_k) * TM, j0 = (tl % q.tiles_k) * TK;

    // LDS-DMA sources.  A wave's piece j (0..3) of an operand = stage rows r0 + 8 j, r0 = 2 wave + (lane >> 5): the row's
    // swizzle s(m) = (m & 3) | ((m >> 3) & 1) << 2 has its high bit from j's parity, so two per-lane column offsets
    const bf16_t* pa[2];
    const bf16_t* pb[2];
    const unsigned sa = (unsigned)q.lda, sb = (unsigned)q.ldb;
    {
      const int lane = lane_now();
      const int r0 = 2 * wave + (lane >> 5);
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int src = (lane & 31) ^ (((r0 & 3) | (par << 2)) << 1);
        pa[par] = q.A + (size_t)(split * P.chunk + r0) * sa + (size_t)(i0 + 8 * src);
        pb[par] = q.B + (size_t)(split * P.chunk + r0) * sb + (size_t)(j0 + 8 * src);
      }
    }
    // piece j (0..3: operand A, 4..7: operand B) of stage kt into the slot at `st`
    auto piece = [&](int j, int kt, unsigned char* st) {
      if (j < 4)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[j & 1] + (size_t)((unsigned)(kt * BK + 8 * j) * sa)),
                                         (lds_void_t*)(st + (wave + 4 * j) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[j & 1] + (size_t)((unsigned)(kt * BK + 8 * (j - 4)) * sb)),
                                         (lds_void_t*)(st + OP_BYTES + (wave + 4 * (j - 4)) * 1024), 16, 0, 0);
    };

    [[maybe_unused]] const bool want_bias = BIAS && q.dbias != nullptr && j0 == 0;
    f32x4_t acc[8][8];
    [[maybe_unused]] f32x4_t accb[4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) accb[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // prologue: NS stages in flight (nk >= 8 by the host's choice of the split), the first one's fragments
#pragma unroll
    for (int s = 0; s < NS4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) piece(j, s, smem + s * STAGE);
    wait_vm<8 * (NS4 - 1)>();
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f0.a[t] = lds_read_tr(ab0 ^ (unsigned)(t << 5));
      f0.b[t] = lds_read_tr(bb0 ^ (unsigned)(t << 5));
    }
    frags_ready(f0);

    // step kt: the 64 MFMAs of stage kt (fragments in `cur`); under them the fragments of stage kt + 1 are read into `nxt`
    // (NEXT) and stage kt + NS is requested into the slot stage kt came from (ISSUE).  VM = LDS-DMA instructions that may
    // still be in flight when stage kt + 1 must have landed (the stages issued behind it, 8 per stage; loads retire in order).
    // The order below IS the issue order (MFMAs and fragment reads are volatile asm, scheduling barriers pin the rest): the
    // first row of MFMAs runs ahead of the barrier (it needs neither the freed slot nor the next stage), the sixteen
    // fragment reads and eight requests are spread over the other seven rows.
    int slot = 0;                          // the slot stage kt lies in (kt mod NS4)
    auto step = [&](auto next_c, auto issue_c, auto vm_c, int kt, Frags& cur, Frags& nxt) {
      constexpr bool NEXT = decltype(next_c)::value, ISSUE = decltype(issue_c)::value;
      constexpr int VM = decltype(vm_c)::value;
      const int slot1 = slot == NS4 - 1 ? 0 : slot + 1;
      const unsigned so1 = (unsigned)(slot1 * STAGE);
      unsigned char* st = smem + slot * STAGE;
      auto read_pair = [&](int t) {
        nxt.a[t] = lds_read_tr((ab0 + so1) ^ (unsigned)(t << 5));
        nxt.b[t] = lds_read_tr((bb0 + so1) ^ (unsigned)(t << 5));
      };
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t == 1 && NEXT) {
          wait_vm<VM>();
          __builtin_amdgcn_s_barrier();    // every wave holds stage kt in registers (slot free) and stage kt + 1 is visible
        }
        if (t >= 1 && NEXT && TN256_ABL != 3 && TN256_ABL != 4) {
          read_pair(t);
          if (t == 1) read_pair(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (TN256_ABL != 1) mfma_acc(acc[t][u], cur.b[u], cur.a[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (t >= 1 && ISSUE && TN256_ABL != 2 && TN256_ABL != 4) {
          piece(t, kt + NS4, st);
          if (t == 1) piece(0, kt + NS4, st);
        }
      }
      if constexpr (BIAS) {
        if (want_bias) {                   // item-uniform; the two k-half waves of an n half take four row tiles each
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16x8_t fsel = wn ? cur.a[4 + i] : cur.a[i];
            // (the builtin would take the AccVGPR form and shuttle these four through a[16:19] every step)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb[i]) : "v"(ones), "v"(fsel));
          }
        }
      }
      if constexpr (NEXT) frags_ready(nxt);
      slot = slot1;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto vm = [](auto c) { return std::integral_constant<int, 8 * decltype(c)::value>{}; };   // stages -> instructions
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // main steps (a request each) in pairs, then the NS4 steps that only drain the ring; nk is even (host), so with an odd
    // ring the first step stands alone and the fragment buffers swap roles
    auto run = [&](int kt, Frags& x, Frags& y) {
      for (; kt < nk - NS4; kt += 2) {
        step(T_{}, T_{}, vm(std::integral_constant<int, NS4 - 2>{}), kt, x, y);
        step(T_{}, T_{}, vm(std::integral_constant<int, NS4 - 2>{}), kt + 1, y, x);
      }
      if constexpr (NS4 == 4) {
        step(T_{}, F_{}, vm(I2{}), kt, x, y);
        step(T_{}, F_{}, vm(I1{}), kt + 1, y, x);
        step(T_{}, F_{}, vm(I0{}), kt + 2, x, y);
        step(F_{}, F_{}, vm(I0{}), kt + 3, y, x);
      } else {
        static_assert(NS4 == 4 || NS4 == 5, "ring depth");
        step(T_{}, F_{}, vm(I3{}), kt, x, y);
        step(T_{}, F_{}, vm(I2{}), kt + 1, y, x);
        step(T_{}, F_{}, vm(I1{}), kt + 2, x, y);
        step(T_{}, F_{}, vm(I0{}), kt + 3, y, x);
        step(F_{}, F_{}, vm(I0{}), kt + 4, x, y);
      }
    };
    if constexpr (NS4 & 1) {
      step(T_{}, T_{}, vm(std::integral_constant<int, NS4 - 2>{}), 0, f0, f1);
      run(1, f1, f0);
    } else {
      run(0, f0, f1);
    }

    // ---- epilogue: fp32 accumulators straight to the slab of this split
    // (the asm MFMAs are opaque to the compiler's hazard recognizer: the last results need 2 + 4 x 2 passes to land)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if constexpr (BIAS) {
      // ALL four registers of each bias accumulator stay live up to here.  Only element 0 is stored, so after the last
      // step's MFMAs the allocator took elements 2-3 of accb[0] as temporaries for the next MFMA's operand while the
      // (opaque) MFMA was still writing them: one or two non-finite bias gradients per ~100 training steps.
      asm volatile("" : "+v"(accb[0]), "+v"(accb[1]), "+v"(accb[2]), "+v"(accb[3]));
    }
    {
      const int lane = lane_now();
      float* slab = q.C + (size_t)split * (size_t)q.N * (size_t)q.K;
      const int n0 = i0 + wm * 128 + (lane & 15), k0 = j0 + wn * 128 + 4 * (lane >> 4);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          *reinterpret_cast<f32x4_t*>(slab + (size_t)(n0 + 16 * t) * q.K + k0 + 16 * u) = acc[t][u];
      if constexpr (BIAS) {
        if (want_bias && (lane >> 4) == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) q.dbias[(size_t)split * q.N + n0 + 16 * (4 * wn + i)] = accb[i][0];
        }
      }
    }
    // the next item's prologue reuses all four slots: every read of them is behind the last step's barrier; its 32 requests
    // and these 64 stores do not fit the 6-bit counter together, so the stores drain first (one item per workgroup at the
    // transformer's shapes: this is the kernel's end there)
    wait_vm<0>();
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Common split count of a group of qualifying problems (0: the group does not qualify): the largest divisor s of M / 32
// with s x (total tiles) <= 224 work items (at most one per CU), items at least 8 stages long and an even stage count, s <= 32
// (the bias partial sums have 32 rows).
int ib_gemm_tn256_splits(int n, const int64_t* M, const int64_t* N, const int64_t* K) {
  static const bool off = ib_ab_set("IB_NO_TN256");
  if (off || n <= 0 || n > MAXP) return 0;
  int64_t tiles = 0;
  for (int j = 0; j < n; ++j) {
    if (M[j] != M[0] || M[j] % (2 * BK) != 0 || M[j] < 4096 || N[j] <= 0 || K[j] <= 0 || N[j] % TM != 0 || K[j] % TK != 0) return 0;
    tiles += (N[j] / TM) * (K[j] / TK);
  }
  const int64_t stages = M[0] / BK;
  static const int forced = ib_ab_int("IB_TN256_SPLITS", 0);   // tuning override
  if (forced > 0 && forced <= 32 && stages % forced == 0 && stages / forced >= 8 && !((stages / forced) & 1)) return forced;
  int best = 0;
  for (int64_t s = 1; s <= 32; ++s) {
    if (stages % s != 0) continue;
    const int64_t nk = stages / s;
    if (nk < 8 || (nk & 1)) continue;
    // 224, not 256: the launch runs on a side stream beside the backward's main-stream GEMMs, and every split is one more
    // fp32 slab for the optimizer to read (transformer layer, 48 tiles, same box: 4 splits 2.110-2.116 ms per step, 5 splits
    // 2.126-2.159)
    if (s * tiles <= 224 || best == 0) best = (int)s;      // s = 1 when even that exceeds the chip (several rounds)
  }
  return best;
}

int ib_gemm_tn256_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x, const int64_t* ldx,
                        void* const* workspace, const size_t* workspace_bytes, float* const* dbias_part, int32_t* nslab_out,
                        const int64_t* M, const int64_t* N, const int64_t* K, hipStream_t s) {
  const int sp = ib_gemm_tn256_splits(n, M, N, K);
  if (sp <= 0) return IB_E_UNSUPPORTED;
  Params P{};
  P.n = n; P.splits = sp; P.chunk = (int)(M[0] / sp); P.nk = P.chunk / BK;
  int items = 0;
  bool any_bias = false;
  for (int j = 0; j < n; ++j) {
    if (!dz[j] || !x[j] || !workspace[j]) return IB_E_ARG;
    if (!al16(dz[j]) || !al16(x[j]) || !al16(workspace[j]) || lddz[j] % 8 || ldx[j] % 8) return IB_E_UNSUPPORTED;
    if (lddz[j] < N[j] || ldx[j] < K[j]) return IB_E_UNSUPPORTED;
    if (M[j] * lddz[j] >= (int64_t(1) << 31) || M[j] * ldx[j] >= (int64_t(1) << 31)) return IB_E_UNSUPPORTED;
    if ((size_t)sp * (size_t)N[j] * (size_t)K[j] * sizeof(float) > workspace_bytes[j]) return IB_E_UNSUPPORTED;
    Prob& q = P.pr[j];
    q.A = (const bf16_t*)dz[j]; q.B = (const bf16_t*)x[j]; q.lda = (int)lddz[j]; q.ldb = (int)ldx[j];
    q.N = (int)N[j]; q.K = (int)K[j];
    q.C = (float*)workspace[j]; q.dbias = dbias_part ? dbias_part[j] : nullptr;
    any_bias = any_bias || q.dbias != nullptr;
    q.tiles_k = (int)(K[j] / TK); q.per_split = (int)(N[j] / TM) * q.tiles_k;
    q.item0 = items;
    items += q.per_split * sp;
    nslab_out[j] = sp;
  }
  P.items = items;
  const int grid = items < 256 ? items : 256;
  IB_PATH(IB_PATH_TN256);
  static const bool v8 = ib_ab_set("IB_TN256_W8");     // measurement builds: the eight-wave form
  if (v8) {
    if (any_bias) hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(grid), dim3(THREADS), 0, s, P);
    else hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(grid), dim3(THREADS), 0, s, P);
  } else {
    if (any_bias) hipLaunchKernelGGL(gemm_tn256w4_kernel<true>, dim3(grid), dim3(THREADS4), 0, s, P);
    else hipLaunchKernelGGL(gemm_tn256w4_kernel<false>, dim3(grid), dim3(THREADS4), 0, s, P);
  }
  IB_CHECK_LAUNCH();
  return IB_OK;
}
