// Backward of the time-embedding MLP's hidden layer as ONE set of independent workgroups (csrc/chain.hip explains the math):
// shared by the standalone launch (ib_time_mlp_bwd: 4 waves per workgroup) and by the rider workgroups of the grouped
// weight-gradient launch (csrc/gemm_tn.hip: 8 waves, the launch's own 144-KB LDS array).
#pragma once
#include "ib_common.h"

constexpr int TB_ROWS = 64, TB_MAXOUT = 2048;
struct TimeBwdParams {
  const bf16_t* de; int64_t ld_de; const bf16_t* w2; int64_t ldw2; const bf16_t* zu; int64_t ldzu;
  const bf16_t* s; int64_t lds; float* dw1; float* db1; int B, out, hidden;
};
// bytes of LDS the body carves from `smem` (16-byte aligned base)
template <int TE, int NW>
constexpr int time_bwd_lds(int out) {
  return 16 * (out + 8) * 2 + NW * TB_ROWS * 16 * 4 + 16 * (TB_ROWS + 8) * 2 + TE * (TB_ROWS + 8) * 2 + TB_ROWS * 16 * 4;
}

// workgroup (cg, rg0 .. rg0 + nrg): hidden columns 16 cg .. +15, windows 64 rg .. +63 of each row group in turn (the W2
// slice is transposed into LDS once per workgroup).  NW waves (64 NW threads), all of them must call.
template <int TE, int NW>
__device__ __forceinline__ void time_bwd_body(const TimeBwdParams& p, int cg, int rg0, int nrg, unsigned char* smem) {
  constexpr int THR = 64 * NW;
  constexpr int SS = TB_ROWS + 8;                            // row strides (elements) of the two small transposed images
  const int WS = p.out + 8;
  bf16_t* wimg = reinterpret_cast<bf16_t*>(smem);                                  // W2[:, c0 .. c0+16)^T : [c][n]
  float* xacc = reinterpret_cast<float*>(smem + 16 * (p.out + 8) * 2);             // [NW][64][16] partial accumulators
  bf16_t* dzT = reinterpret_cast<bf16_t*>(xacc + NW * TB_ROWS * 16);               // dzu^T : [c][row]
  bf16_t* sT = dzT + 16 * SS;                                                      // s^T   : [te][row]
  float* cred = reinterpret_cast<float*>(sT + TE * SS);                            // [64][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = cg * 16;
  const int r = lane & 15, kq = lane >> 4;
  const int nkb = p.out / 32, kpw = nkb / NW;                // k-blocks per wave (out % (32 NW) == 0): kb = wave, wave + NW, ...
  // Every global request of a row group goes out BEFORE anything is waited for (each dependent first-touch round trip is
  // 2-3 us here): the de fragments of this wave's first k-blocks, the s rows, the pre-activations (+ the W2 slice once).
  // With several row groups per workgroup (the rider form) the NEXT group's requests are issued as soon as the current
  // group's registers are dead (after step 3), so a further group costs its ~3 us of arithmetic, not another round trip.
  constexpr int PF = NW == 4 ? 8 : 4;
  constexpr int EPT = TB_ROWS * 16 / THR;                    // 4 (NW = 4) or 2 (NW = 8) consecutive columns of one row
  const int xrow = tid / (16 / EPT), xc = EPT * (tid % (16 / EPT));
  constexpr int SPN = TB_ROWS * (TE / 8);
  constexpr int SP = (SPN + THR - 1) / THR;
  bf16x8_t b[PF][4], bn[PF][4];
  bf16_t zu_[EPT], zun[EPT];
  bf16x8_t sv[SP], svn[SP];
  auto issue = [&](int rg, bf16x8_t (&bb)[PF][4], bf16_t (&zz)[EPT], bf16x8_t (&ss)[SP]) {
    const int r0 = rg * TB_ROWS;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int kb = wave + NW * min(i, kpw - 1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bb[i][mt] = *reinterpret_cast<const bf16x8_t*>(p.de + (int64_t)min(r0 + 16 * mt + r, p.B - 1) * p.ld_de + 8 * kq + 32 * kb);
    }
    const bf16_t* zp = p.zu + (int64_t)min(r0 + xrow, p.B - 1) * p.ldzu + c0 + xc;
#pragma unroll
    for (int e = 0; e < EPT; ++e) zz[e] = zp[e];
    // (piece -> (row = pc % 64, column group = pc / 64): consecutive lanes take consecutive ROWS, so the transposing 2-byte LDS
    // writes below fall on consecutive addresses.  With the column group on the lane -- the coalesced choice for the loads --
    // sixteen lanes hit one bank with sixteen addresses (16-byte-aligned image rows are a multiple of 8 banks apart): the
    // write pass was 3.6 us of LDS time per row group, most of the standalone launch's arithmetic)
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      const int pc = min(tid + j * THR, SPN - 1), row = pc % TB_ROWS, t8 = pc / TB_ROWS;
      ss[j] = *reinterpret_cast<const bf16x8_t*>(p.s + (int64_t)min(r0 + row, p.B - 1) * p.lds + 8 * t8);
    }
  };
  issue(rg0, b, zu_, sv);
  for (int rg = rg0; rg < rg0 + nrg; ++rg) {
  const int r0 = rg * TB_ROWS;
  const int nrows = min(TB_ROWS, p.B - r0);
  // the NEXT row group's requests go out before this one's arithmetic (a second register set: a whole group's ~3 us of
  // barriers and MFMAs cover their round trip; issued after step 3 they had half a microsecond)
  const bool more = rg + 1 < rg0 + nrg;
  if (more) issue(rg + 1, bn, zun, svn);
  const bf16_t* drow[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) drow[mt] = p.de + (int64_t)min(r0 + 16 * mt + r, p.B - 1) * p.ld_de + 8 * kq;
  // 1. W2 slice -> LDS, transposed (16-byte pieces = 8 columns of one reduction row; up to eight pieces per thread in flight)
  if (rg == rg0)
  for (int p0 = tid; p0 < p.out * 2; p0 += THR * 8) {
    bf16x8_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pce = min(p0 + THR * u, p.out * 2 - 1), k = pce >> 1, half = pce & 1;
      v[u] = *reinterpret_cast<const bf16x8_t*>(p.w2 + (int64_t)k * p.ldw2 + c0 + 8 * half);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pce = p0 + THR * u, k = pce >> 1, half = pce & 1;
      if (pce < p.out * 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) wimg[(8 * half + e) * WS + k] = v[u][e];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SP; ++j) {
    const int pc = tid + j * THR, row = pc % TB_ROWS, t8 = pc / TB_ROWS;
    if (pc < SPN) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sT[(8 * t8 + e) * SS + row] = sv[j][e];
    }
  }
  __syncthreads();
  // 2. partial dzu^T[c][row] over this wave's k-blocks
  f32x4_t acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bf16_t* wrow = wimg + r * WS + 8 * kq;
  for (int i0 = 0; i0 < kpw; i0 += PF) {
    if (i0 > 0) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int kb = wave + NW * min(i0 + i, kpw - 1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) b[i][mt] = *reinterpret_cast<const bf16x8_t*>(drow[mt] + 32 * kb);
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (i0 + i < kpw) {
        const int kb = wave + NW * (i0 + i);
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(wrow + 32 * kb);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i][mt], acc[mt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)                              // lane: columns 4 kq .. +3 of row 16 mt + r
    *reinterpret_cast<f32x4_t*>(xacc + (wave * TB_ROWS + 16 * mt + r) * 16 + 4 * kq) = acc[mt];
  __syncthreads();
  // 3. sum the waves' partials in wave order, * silu'(zu), round; dzu^T image + this row's share of the bias sums
  {
    const bool live = xrow < nrows;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += xacc[(w * TB_ROWS + xrow) * 16 + xc + e];
      const float x = (float)zu_[e];
      const float sg = 1.f / (1.f + __expf(-x));
      const bf16_t o = (bf16_t)(live ? v * (sg * (1.f + x * (1.f - sg))) : 0.f);
      dzT[(xc + e) * SS + xrow] = o;
      cred[xrow * 16 + xc + e] = (float)o;                     // the bias gradient sums the STORED values
    }
  }
  __syncthreads();
  if (tid < 16) {
    float sm = 0.f;
#pragma unroll 16
    for (int row = 0; row < TB_ROWS; ++row) sm += cred[row * 16 + tid];
    p.db1[(int64_t)rg * p.hidden + c0 + tid] = sm;
  }
  // 4. partial dW1[c0 .. c0+16)[:] = dzu^T s over the group's 64 windows (two k-blocks), n-tiles over the waves
  for (int nt = wave; nt < TE / 16; nt += NW) {
    f32x4_t a2 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < TB_ROWS / 32; ++kb) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(dzT + r * SS + 32 * kb + 8 * kq);
      const bf16x8_t bb = *reinterpret_cast<const bf16x8_t*>(sT + (16 * nt + r) * SS + 32 * kb + 8 * kq);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, a2, 0, 0, 0);
    }
    // D[c][te]: this lane holds rows c = 4 kq + e of column te = 16 nt + r
    float* dst = p.dw1 + ((int64_t)rg * p.hidden + c0 + 4 * kq) * TE + 16 * nt + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(int64_t)e * TE] = a2[e];
  }
  __syncthreads();                                             // the images are rewritten by the next row group
  if (more) {
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) b[i][mt] = bn[i][mt];
#pragma unroll
    for (int e = 0; e < EPT; ++e) zu_[e] = zun[e];
#pragma unroll
    for (int j = 0; j < SP; ++j) sv[j] = svn[j];
  }
  }   // row groups
}
