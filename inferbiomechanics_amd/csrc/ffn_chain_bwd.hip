// Backward of the fused token-local half of a transformer layer (see ffn_chain.hip for the design; shared helpers in
// ffn_chain.h).  A translation unit of its own: see ffn_chain.h.
#include "ffn_chain.h"

namespace {

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
  // QKV head (QKVH): dy is not read but computed here = dqkv_next . Wqkv_next + ds1_next (the NEXT layer's in-projection
  // dgrad + its residual addend), three 512-column chunks of dqkv
  const bf16_t* wqkvtp; const bf16_t* dqkv_next; const bf16_t* ds1_next;
  // attention tail (ATT; panel = one window of P = T frames): dattn is not stored; the attention backward of this layer's
  // eight (window, head) pairs runs wave-private on it -> dqkv [M, 1536] (the in-projection's weight-gradient operand), then
  // dx = dqkv . Wqkv + ds1 (the in-projection's dgrad + the residual addend) -> dx [M, 512]: the whole layer in one launch
  const bf16_t* qkv; const float* lse; bf16_t* dqkv; bf16_t* dx; const bf16_t* wqkvtp_own;
  int M, P, FF, nchunk;
  long long* prof;
};

// dst rows (image Z) = dqkv . Wqkv + addend: the addend rows wait in image Z, dqkv comes through image D one 512-column
// chunk at a time (the next chunk's rows are requested during this chunk's GEMM and land after the barrier).  Leaves image Z
// complete behind a barrier.
__device__ __forceinline__ void ff_qkv_dgrad(const bf16_t* addend, const bf16_t* dq, const bf16_t* wqkvtp, unsigned char* imgZ,
                                             unsigned char* imgD, int nrows, int tid, int wave_s, int l16, int colb) {
  ff_panel_in(addend, imgZ, nrows, tid);
  {
    uint4 xr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = tid + j * FF_THREADS;
      xr[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(dq + (int64_t)min(idx >> 6, nrows - 1) * (3 * FF_D)) + (idx & 63) * 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = tid + j * FF_THREADS;
      *reinterpret_cast<uint4*>(imgD + (idx >> 6) * FF_RS + (idx & 63) * 16) = xr[j];
    }
  }
  __syncthreads();
  f32x4_t accd[4][FF_NT];
  ff_zero(accd);
  for (int c = 0; c < 3; ++c) {
    // the next chunk's eight row pieces, requested one per two k-blocks behind the weight stream and parked in EIGHT NAMED
    // registers until the barrier (as an array captured by the side job they stayed in scratch: every piece was waited
    // for with vmcnt(0) right behind its load, draining the weight ring sixteen times per chunk)
    uint4 n0, n1, n2, n3, n4, n5, n6, n7;
    n0 = n1 = n2 = n3 = n4 = n5 = n6 = n7 = make_uint4(0u, 0u, 0u, 0u);
    // (the last trip re-requests chunk 2 and drops it: no branch around a load inside the k-loop)
    const bf16_t* dqn = dq + min(c + 1, 2) * FF_CHUNK;
    auto piece = [&](int j) {
      const int idx = FF_TIDV + j * FF_THREADS;
      return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(dqn + (int64_t)min(idx >> 6, nrows - 1) * (3 * FF_D)) + (idx & 63) * 16);
    };
    auto sided = [&](auto kbc, int) {
      constexpr int kb = decltype(kbc)::value;
      if constexpr (kb == 0) n0 = piece(0);
      if constexpr (kb == 2) n1 = piece(1);
      if constexpr (kb == 4) n2 = piece(2);
      if constexpr (kb == 6) n3 = piece(3);
      if constexpr (kb == 8) n4 = piece(4);
      if constexpr (kb == 10) n5 = piece(5);
      if constexpr (kb == 12) n6 = piece(6);
      if constexpr (kb == 14) n7 = piece(7);
    };
    ff_gemm<FF_RING_BB>(wqkvtp + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgD, ff_lane(), accd, sided);
    __syncthreads();                   // every wave is done reading this chunk out of image D
    if (c + 1 < 3) {
      auto put = [&](int j, const uint4& v) {
        const int idx = tid + j * FF_THREADS;
        *reinterpret_cast<uint4*>(imgD + (idx >> 6) * FF_RS + (idx & 63) * 16) = v;
      };
      put(0, n0); put(1, n1); put(2, n2); put(3, n3); put(4, n4); put(5, n5); put(6, n6); put(7, n7);
      __syncthreads();
    }
  }
#pragma unroll
  for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x4_t* slot = reinterpret_cast<bf16x4_t*>(imgZ + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2);
      const bf16x4_t av = *slot;
      *slot = ff_pack4(accd[mt][u][0] + (float)av[0], accd[mt][u][1] + (float)av[1], accd[mt][u][2] + (float)av[2],
                       accd[mt][u][3] + (float)av[3]);
    }
  }
  __syncthreads();                     // image Z = dqkv . Wqkv + addend
}

template <bool OUT, bool QKVH, bool ATT>
__global__ __launch_bounds__(FF_THREADS) void ffn_chain_bwd_kernel(FfnBwdParams p) {
  static_assert(!ATT || OUT, "the attention backward rides behind the out-projection's dgrad");
  __shared__ float attL[ATT ? FF_WAVES : 1][64], attD[ATT ? FF_WAVES : 1][64];   // per head: the rows' lse, D = sum_key P dP
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
  FF_STAMP(0);
  if constexpr (QKVH) {
    // ---- dy = dqkv_next . Wqkv_next + ds1_next (the NEXT layer's in-projection dgrad + its residual addend) -> image Z
    ff_qkv_dgrad(p.ds1_next + (int64_t)r0 * FF_D, p.dqkv_next + (int64_t)r0 * (3 * FF_D), p.wqkvtp, imgZ, imgD, nrows, tid,
                 wave_s, l16, colb);
  }
  // ---- LayerNorm2 backward, row-wise: dz2 rows -> image Z (+ HBM); its dgamma | dbeta through image D's storage
  {
    float dgam[8], dbet[8];
    ff_ln_rows_bwd<QKVH>(p.dy + (int64_t)r0 * FF_D, imgZ, p.s2 + (int64_t)r0 * FF_D, p.mean + r0, p.rstd + r0, p.gamma, nrows,
                         wave_s, lane, imgZ, p.ds2 + (int64_t)r0 * FF_D, dgam, dbet);
    ff_colsum_put(reinterpret_cast<float*>(imgD), wave, lane, dgam, dbet);
  }
  __syncthreads();                       // image Z = dz2 complete; the exchange rows are written
  ff_colsum_out(reinterpret_cast<const float*>(imgD), p.partial, 0, tid);
  __syncthreads();                       // image D is free for the first chunk
  FF_STAMP(1);

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
  FF_STAMP(2);
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
    ff_ln_rows_bwd<true>(nullptr, imgD, p.s1 + (int64_t)r0 * FF_D, p.mean1 + r0, p.rstd1 + r0, p.gamma1, nrows, wave_s, lane,
                   imgZ, p.ds1 + (int64_t)r0 * FF_D, dgam, dbet);
    __syncthreads();                     // image Z = ds1 complete; image D (dx1 rows) no longer read: it takes the exchange
    ff_colsum_put(reinterpret_cast<float*>(imgD), wave, lane, dgam, dbet);
    __syncthreads();
    ff_colsum_out(reinterpret_cast<const float*>(imgD), p.partial, 2, tid);
    __syncthreads();                     // ... and is free again for the out-projection's result
    FF_STAMP(3);
    f32x4_t acco[4][FF_NT];
    ff_zero(acco);
    // (ATT: requesting the K / Q row fragments of the head as side jobs of this phase was measured a wash -- the phase got
    // 3.1 us longer, the load step behind it 2.8 us shorter: the same bytes through the same L2 -> CU path, plus 3 spills)
    const bf16_t* qb = ATT ? p.qkv + (int64_t)r0 * (3 * FF_D) + 64 * wave_s + 8 * g : nullptr;
    auto sideo = [&](auto, int) {};
    ff_gemm<FF_RING_BB>(p.wotp, wave_s * FF_NT, imgZ, ff_lane(), acco, sideo);
#pragma unroll
    for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        *reinterpret_cast<bf16x4_t*>(imgD + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) =
            ff_pack4(acco[mt][u][0], acco[mt][u][1], acco[mt][u][2], acco[mt][u][3]);
    }
    if constexpr (!ATT) {
      __syncthreads();
      ff_panel_out(imgD, p.dattn + (int64_t)r0 * FF_D, nrows, tid);
    } else {
      // ---- attention backward of head `wave` of this window, wave-private (attention_mfma.hip's two phases; all sums in
      // registers, fixed order).  dO = the dattn columns just written into this wave's slice of image D (SB); Q, K, V row
      // fragments of the head straight from HBM; K (phase 1), then Q (phase 2) in this wave's slice of
      // image Z (SA) for the transposed reads.  The slices of other waves are never touched: no workgroup barrier until the
      // results meet in the images.  dQ, dK, dV stay ON CHIP: bf16 in registers until both slices are free, then dQ -> SA,
      // dK -> SB (the operands of the in-projection dgrad's first two phases), dV in registers until image Z is free again.
      FF_STAMP(4);
      __syncthreads();                   // every wave is past the out-projection GEMM: image Z (ds1) may be overwritten
      FF_STAMP(5);
      unsigned char* SA = imgZ + 128 * wave_s;
      unsigned char* SB = imgD + 128 * wave_s;
      bf16x8_t qf[4][2], kf[4][2], vf[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16_t* row = qb + (int64_t)min(16 * t + l16, nrows - 1) * (3 * FF_D);   // rows beyond the window: finite duplicates
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[t][ks] = *reinterpret_cast<const bf16x8_t*>(row + 32 * ks);
          kf[t][ks] = *reinterpret_cast<const bf16x8_t*>(row + FF_D + 32 * ks);
          vf[t][ks] = *reinterpret_cast<const bf16x8_t*>(row + 2 * FF_D + 32 * ks);
        }
      }
      attL[wave_s][lane] = lane < nrows ? p.lse[((int64_t)blockIdx.x * FF_HEADS + wave_s) * p.P + lane] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          *reinterpret_cast<bf16x8_t*>(SA + (16 * t + l16) * FF_RS + 64 * ks + 16 * g) = kf[t][ks];
      ff_wave_sync();
      FF_STAMP(6);
      // ---------------- phase 1: dQ per 16-query tile (the lane owns query 16 it + lane % 16)
      bf16x4_t dqp[4][4], dkp[4][4], dvp[4][4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = 16 * it + l16;
        const bf16x8_t gf0 = ff_sl_row(SB, q, 0, lane), gf1 = ff_sl_row(SB, q, 1, lane);
        const bf16x8_t q0 = qf[it][0], q1 = qf[it][1];
        const float Lq = attL[wave_s][q];
        f32x4_t pr[4], dpr[4];
        float Dq = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][0], q0, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][1], q1, a, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt][0], gf0, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt][1], gf1, dp, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = (16 * jt + 4 * g + r) < nrows ? __expf(a[r] * FF_ATT_SCALE - Lq) : 0.f;
            Dq += pv * dp[r];
            a[r] = pv;
          }
          pr[jt] = a;
          dpr[jt] = dp;
        }
        Dq = ff_g4_sum(Dq);
        if (g == 0) attD[wave_s][q] = Dq;
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          f32x4_t ds2[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds2[hh][r] = pr[2 * kp + hh][r] * (dpr[2 * kp + hh][r] - Dq);
          const bf16x8_t sf = ff_acc_frag(ds2[0], ds2[1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_tr(SA, 32 * kp, 32 * kp + 16, dt, lane), sf, dq[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqp[it][dt] = ff_pack4(dq[dt][0] * FF_ATT_SCALE, dq[dt][1] * FF_ATT_SCALE, dq[dt][2] * FF_ATT_SCALE,
                                 dq[dt][3] * FF_ATT_SCALE);
        // (the tiles are independent: left alone the scheduler interleaves them and the live set outgrows 256 VGPRs --
        // 124 spilled registers measured; one tile at a time fits)
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_STAMP(7);
      ff_wave_sync();                    // phase 1 is done reading K out of SA; D of every query is in LDS
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          *reinterpret_cast<bf16x8_t*>(SA + (16 * t + l16) * FF_RS + 64 * ks + 16 * g) = qf[t][ks];
      ff_wave_sync();
      // ---------------- phase 2: dK, dV per 16-key tile (the lane owns key 16 jt + lane % 16)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const bf16x8_t k0 = kf[jt][0], k1 = kf[jt][1], v0 = vf[jt][0], v1 = vf[jt][1];
        f32x4_t dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          f32x4_t pp2[2], ds2[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int qt = 2 * qp + hh;
            const int qrow = 16 * qt + l16;
            f32x4_t a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_row(SA, qrow, 0, lane), k0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_row(SA, qrow, 1, lane), k1, a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_row(SB, qrow, 0, lane), v0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_row(SB, qrow, 1, lane), v1, dp, 0, 0, 0);
            const float4 L4 = *reinterpret_cast<const float4*>(&attL[wave_s][16 * qt + 4 * g]);
            const float4 D4 = *reinterpret_cast<const float4*>(&attD[wave_s][16 * qt + 4 * g]);
            const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = (16 * qt + 4 * g + r) < nrows ? __expf(a[r] * FF_ATT_SCALE - Lr[r]) : 0.f;
              a[r] = pv;
              dp[r] = pv * (dp[r] - Dr[r]);
            }
            pp2[hh] = a;
            ds2[hh] = dp;
          }
          const bf16x8_t pf = ff_acc_frag(pp2[0], pp2[1]);
          const bf16x8_t sf = ff_acc_frag(ds2[0], ds2[1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_tr(SB, 32 * qp, 32 * qp + 16, dt, lane), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff_sl_tr(SA, 32 * qp, 32 * qp + 16, dt, lane), sf, dk[dt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dkp[jt][dt] = ff_pack4(dk[dt][0] * FF_ATT_SCALE, dk[dt][1] * FF_ATT_SCALE, dk[dt][2] * FF_ATT_SCALE,
                                 dk[dt][3] * FF_ATT_SCALE);
          dvp[jt][dt] = ff_pack4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_STAMP(8);
      ff_wave_sync();                    // this wave is done reading Q / dO out of its slices: they take dQ / dK
      // rows (16 tile + lane % 16), columns (16 dt + 4 g ..) of the slice: the layout every GEMM epilogue writes
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          *reinterpret_cast<bf16x4_t*>(SA + (16 * t + l16) * FF_RS + (16 * dt + 4 * g) * 2) = dqp[t][dt];
          *reinterpret_cast<bf16x4_t*>(SB + (16 * t + l16) * FF_RS + (16 * dt + 4 * g) * 2) = dkp[t][dt];
        }
      __syncthreads();                   // image Z = dQ, image D = dK of all eight heads
      FF_STAMP(9);
      // ---- dx = dqkv . Wqkv + ds1: the in-projection's dgrad straight from the images; the dqkv rows leave for HBM (the
      // in-projection's weight-gradient operand) as the side jobs of the phases that read them, one piece per two k-blocks
      bf16_t* dqg = p.dqkv + (int64_t)r0 * (3 * FF_D);
      f32x4_t accd[4][FF_NT];
      ff_zero(accd);
      auto side_q = [&](auto, int kb) {
        if ((kb & 1) == 0) ff_out_piece(imgZ, dqg, 3 * FF_D, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
      };
      ff_gemm<FF_RING_BB>(p.wqkvtp_own, wave_s * FF_NT, imgZ, ff_lane(), accd, side_q);
      auto side_k = [&](auto, int kb) {
        if ((kb & 1) == 0) ff_out_piece(imgD, dqg + FF_CHUNK, 3 * FF_D, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
      };
      ff_gemm<FF_RING_BB>(p.wqkvtp_own + FF_WELEMS, wave_s * FF_NT, imgD, ff_lane(), accd, side_k);
      __syncthreads();                   // every wave is past both phases: the images may be rewritten
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<bf16x4_t*>(SA + (16 * t + l16) * FF_RS + (16 * dt + 4 * g) * 2) = dvp[t][dt];
      __syncthreads();                   // image Z = dV
      // third phase: dV rows out on the even k-blocks, the residual addend's rows (ds1, written by this workgroup's
      // LayerNorm1 backward above: visible at workgroup scope) requested on the odd ones and parked in named registers
      const bf16_t* ad = p.ds1 + (int64_t)r0 * FF_D;
      uint4 n0, n1, n2, n3, n4, n5, n6, n7;
      n0 = n1 = n2 = n3 = n4 = n5 = n6 = n7 = make_uint4(0u, 0u, 0u, 0u);
      auto piece = [&](int j) {
        const int idx = FF_TIDV + j * FF_THREADS;
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(ad + (int64_t)min(idx >> 6, nrows - 1) * FF_D) + (idx & 63) * 16);
      };
      auto side_v = [&](auto kbc, int) {
        constexpr int kb = decltype(kbc)::value;
        if constexpr ((kb & 1) == 0) ff_out_piece(imgZ, dqg + 2 * FF_CHUNK, 3 * FF_D, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
        if constexpr (kb == 1) n0 = piece(0);
        if constexpr (kb == 3) n1 = piece(1);
        if constexpr (kb == 5) n2 = piece(2);
        if constexpr (kb == 7) n3 = piece(3);
        if constexpr (kb == 9) n4 = piece(4);
        if constexpr (kb == 11) n5 = piece(5);
        if constexpr (kb == 13) n6 = piece(6);
        if constexpr (kb == 15) n7 = piece(7);
      };
      ff_gemm<FF_RING_BB>(p.wqkvtp_own + 2 * FF_WELEMS, wave_s * FF_NT, imgZ, ff_lane(), accd, side_v);
      {
        auto put = [&](int j, const uint4& v) {           // image D is free since the barrier behind the second phase
          const int idx = tid + j * FF_THREADS;
          *reinterpret_cast<uint4*>(imgD + (idx >> 6) * FF_RS + (idx & 63) * 16) = v;
        };
        put(0, n0); put(1, n1); put(2, n2); put(3, n3); put(4, n4); put(5, n5); put(6, n6); put(7, n7);
      }
      __syncthreads();                   // image D = ds1 rows
#pragma unroll
      for (int u = 0; u < FF_NT; ++u) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          bf16x4_t* slot = reinterpret_cast<bf16x4_t*>(imgD + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2);
          const bf16x4_t av = *slot;
          *slot = ff_pack4(accd[mt][u][0] + (float)av[0], accd[mt][u][1] + (float)av[1], accd[mt][u][2] + (float)av[2],
                           accd[mt][u][3] + (float)av[3]);
        }
      }
      __syncthreads();                   // image D = dx
      FF_STAMP(10);
      ff_panel_out(imgD, p.dx + (int64_t)r0 * FF_D, nrows, tid);
      FF_STAMP(11);
    }
  }
}

}  // namespace

namespace {
int ffn_chain_bwd_launch(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                         const void* packed, const void* mask, void* ds2, void* dz1, void* dx1, float* partial,
                         const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1,
                         void* dattn, const void* packed_next, const void* dqkv_next, const void* ds1_next,
                         const void* qkv, const float* lse, void* dqkv, void* dx, int64_t T, int64_t M, int64_t d,
                         int64_t ffn, ib_stream_t stream) {
  FfnBwdParams p{};
  int P = 0, nc = 0;
  const bool att = dqkv != nullptr;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc, att ? T : 0);
  if (!nwg) return IB_E_UNSUPPORTED;
  const bool head = dqkv_next != nullptr;
  if ((!dy && !head) || !s2 || !mean || !rstd || !gamma || !packed || !mask || !ds2 || !dz1 || !partial) return IB_E_ARG;
  if (!ff_al16({dy, s2, gamma, packed, mask, ds2, dz1, dx1, partial})) return IB_E_ARG;
  const bool out = s1 != nullptr;
  if (head && (!out || !packed_next || !ds1_next || !ff_al16({packed_next, dqkv_next, ds1_next}))) return IB_E_ARG;
  if (out ? (!mean1 || !rstd1 || !gamma1 || !ds1 || (!dattn && !att) || !ff_al16({s1, gamma1, ds1, dattn})) : !dx1) return IB_E_ARG;
  if (att && (!out || head || !qkv || !lse || !dx || T <= 0 || P != T || !ff_al16({qkv, dqkv, dx}))) return IB_E_ARG;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.dy = (const bf16_t*)dy; p.s2 = (const bf16_t*)s2; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.w2tp = pk + (int64_t)2 * nc * FF_WELEMS; p.w1tp = pk + (int64_t)3 * nc * FF_WELEMS;
  p.mask = (const uint2*)mask; p.ds2 = (bf16_t*)ds2; p.dz1 = (bf16_t*)dz1; p.dx1 = (bf16_t*)dx1;
  p.partial = partial;
  p.s1 = (const bf16_t*)s1; p.mean1 = mean1; p.rstd1 = rstd1; p.gamma1 = gamma1;
  p.wotp = pk + (int64_t)(4 * nc + 1) * FF_WELEMS; p.ds1 = (bf16_t*)ds1; p.dattn = (bf16_t*)dattn;
  // (the neighbour's packed image is addressed with THIS layer's chunk count: both layers must have the same hidden width)
  p.wqkvtp = head ? reinterpret_cast<const bf16_t*>(packed_next) + (int64_t)(4 * nc + 5) * FF_WELEMS : nullptr;
  p.dqkv_next = (const bf16_t*)dqkv_next; p.ds1_next = (const bf16_t*)ds1_next;
  p.qkv = (const bf16_t*)qkv; p.lse = lse; p.dqkv = (bf16_t*)dqkv; p.dx = (bf16_t*)dx;
  p.wqkvtp_own = pk + (int64_t)(4 * nc + 5) * FF_WELEMS;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc;
  p.prof = IB_AB_PROF(g_ffn_prof);
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (att) hipLaunchKernelGGL((ffn_chain_bwd_kernel<true, false, true>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (head) hipLaunchKernelGGL((ffn_chain_bwd_kernel<true, true, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (out) hipLaunchKernelGGL((ffn_chain_bwd_kernel<true, false, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL((ffn_chain_bwd_kernel<false, false, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
}  // namespace

extern "C" int ib_ffn_chain_bwd(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                                const void* packed, const void* mask, void* ds2, void* dz1, void* dx1, float* partial,
                                const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1,
                                void* dattn, const void* packed_next, const void* dqkv_next, const void* ds1_next, int64_t M,
                                int64_t d, int64_t ffn, ib_stream_t stream) {
  return ffn_chain_bwd_launch(dy, s2, mean, rstd, gamma, packed, mask, ds2, dz1, dx1, partial, s1, mean1, rstd1, gamma1, ds1,
                              dattn, packed_next, dqkv_next, ds1_next, nullptr, nullptr, nullptr, nullptr, 0, M, d, ffn, stream);
}
// The whole layer's backward in one launch (panels of exactly one window of T frames, 16 <= T <= 64, eight heads of 64):
// behind the out-projection's dgrad the attention backward of the panel's eight (window, head) pairs (qkv [M, 1536] and
// lse [M / T, 8, T] as the forward left them) -> dqkv [M, 1536], then dx [M, 512] = dqkv . Wqkv + ds1
// (TransformerBaseline.py:12-13,29-31 backward).  dattn is not stored.
extern "C" int ib_ffn_chain_bwd_attn(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                                     const void* packed, const void* mask, void* ds2, void* dz1, float* partial,
                                     const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1,
                                     const void* qkv, const float* lse, void* dqkv, void* dx, int64_t T, int64_t M,
                                     int64_t d, int64_t ffn, ib_stream_t stream) {
  if (!dqkv || !dy || T <= 0) return IB_E_ARG;
  return ffn_chain_bwd_launch(dy, s2, mean, rstd, gamma, packed, mask, ds2, dz1, nullptr, partial, s1, mean1, rstd1, gamma1, ds1,
                              nullptr, nullptr, nullptr, nullptr, qkv, lse, dqkv, dx, T, M, d, ffn, stream);
}
