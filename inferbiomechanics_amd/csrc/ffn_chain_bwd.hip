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
  int M, P, FF, nchunk;
  long long* prof;
};

template <bool OUT, bool QKVH>
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
  if constexpr (QKVH) {
    // ---- dy = dqkv_next . Wqkv_next + ds1_next: the addend rows wait in image Z, dqkv comes through image D one 512-column
    // chunk at a time (the next chunk's rows are requested during this chunk's GEMM and land after the barrier)
    ff_panel_in(p.ds1_next + (int64_t)r0 * FF_D, imgZ, nrows, tid);
    const bf16_t* dq = p.dqkv_next + (int64_t)r0 * (3 * FF_D);
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
      ff_gemm<FF_RING_BB>(p.wqkvtp + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgD, ff_lane(), accd, sided);
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
    __syncthreads();                     // image Z = dy
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
    ff_ln_rows_bwd<true>(nullptr, imgD, p.s1 + (int64_t)r0 * FF_D, p.mean1 + r0, p.rstd1 + r0, p.gamma1, nrows, wave_s, lane,
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

}  // namespace

extern "C" int ib_ffn_chain_bwd(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                                const void* packed, const void* mask, void* ds2, void* dz1, void* dx1, float* partial,
                                const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1,
                                void* dattn, const void* packed_next, const void* dqkv_next, const void* ds1_next, int64_t M,
                                int64_t d, int64_t ffn, ib_stream_t stream) {
  FfnBwdParams p{};
  int P = 0, nc = 0;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc);
  if (!nwg) return IB_E_UNSUPPORTED;
  const bool head = dqkv_next != nullptr;
  if ((!dy && !head) || !s2 || !mean || !rstd || !gamma || !packed || !mask || !ds2 || !dz1 || !partial) return IB_E_ARG;
  if (!ff_al16({dy, s2, gamma, packed, mask, ds2, dz1, dx1, partial})) return IB_E_ARG;
  const bool out = s1 != nullptr;
  if (head && (!out || !packed_next || !ds1_next || !ff_al16({packed_next, dqkv_next, ds1_next}))) return IB_E_ARG;
  if (out ? (!mean1 || !rstd1 || !gamma1 || !ds1 || !dattn || !ff_al16({s1, gamma1, ds1, dattn})) : !dx1) return IB_E_ARG;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.dy = (const bf16_t*)dy; p.s2 = (const bf16_t*)s2; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.w2tp = pk + (int64_t)2 * nc * FF_WELEMS; p.w1tp = pk + (int64_t)3 * nc * FF_WELEMS;
  p.mask = (const uint2*)mask; p.ds2 = (bf16_t*)ds2; p.dz1 = (bf16_t*)dz1; p.dx1 = (bf16_t*)dx1;
  p.partial = partial;
  p.s1 = (const bf16_t*)s1; p.mean1 = mean1; p.rstd1 = rstd1; p.gamma1 = gamma1;
  p.wotp = pk + (int64_t)(4 * nc + 1) * FF_WELEMS; p.ds1 = (bf16_t*)ds1; p.dattn = (bf16_t*)dattn;
  p.wqkvtp = head ? reinterpret_cast<const bf16_t*>(packed_next) + (int64_t)(4 * nc + 5) * FF_WELEMS : nullptr;
  p.dqkv_next = (const bf16_t*)dqkv_next; p.ds1_next = (const bf16_t*)ds1_next;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc;
  p.prof = nullptr;
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (head) hipLaunchKernelGGL((ffn_chain_bwd_kernel<true, true>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (out) hipLaunchKernelGGL((ffn_chain_bwd_kernel<true, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL((ffn_chain_bwd_kernel<false, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
