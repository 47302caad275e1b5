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
#include "ffn_chain.h"

#ifdef IB_AB
long long* g_ffn_prof = nullptr;     // TIMING-ONLY (tools/ffn_prof.py, tools/layer_prof.py): [workgroups][64] wall-clock stamps,
                                     // measurement builds; shared with ffn_chain_bwd.hip (declared in ffn_chain.h)
#endif
namespace {

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
  // QKV tail (QKV): the NEXT layer's in-projection qkv = y . Wqkv^T + bqkv, three 512-column chunks
  const bf16_t* wqkvp; const float* bqkv; bf16_t* qkv;     // qkv: [M, 1536]
  // attention of the next layer (ATT; panel = one window of P = T frames): attn_next [M, 512] = softmax(Q K^T / 8) V per
  // (window, head), lse_next [windows, 8, T] = the rows' log-sum-exp (the backward recomputes the probabilities from it)
  bf16_t* attn_next; float* lse_next;
  int M, P, FF, nchunk;
  float ln_eps;
  long long* prof;
};

// INFER: the frozen-weight forward (DDIM sampler at batches beyond the row-panel kernels of linln_panel.hip): nothing is saved
// for a backward -- no f1 / s1 / s2 / x1 rows, no statistics, no ReLU bits leave the workgroup
template <bool OUT, bool QKV, bool ATT, bool INFER = false>
__global__ __launch_bounds__(FF_THREADS) void ffn_chain_fwd_kernel(FfnFwdParams p) {
  static_assert(!ATT || QKV, "the attention rides behind the QKV tail");
  static_assert(!INFER || (OUT && !ATT), "the frozen-weight form: attention epilogue, no attention tail");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FF_BUF];
  unsigned char* imgX = smem;
  unsigned char* imgH = smem + FF_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
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
    if constexpr (INFER)
      ff_ln_rows_fwd(imgX, true, nrows, wave_s, lane, gm1, bt1, p.ln_eps, nullptr, nullptr, nullptr, nullptr);
    else
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
    if constexpr (!INFER) {
      if (p.mask) p.mask[((int64_t)blockIdx.x * p.nchunk + c) * FF_THREADS + tid] = make_uint2(mlo, mhi);
    }
    __syncthreads();                     // image H = ReLU chunk complete
    FF_STAMP(4 + 4 * c);
    bf16_t* f1g = p.f1 + (int64_t)r0 * p.FF + c * FF_CHUNK;
    auto side2 = [&](auto, int kb) {     // the chunk's rows -> HBM: 8 pieces per thread, one per two k-blocks
      if constexpr (!INFER) {
        if ((kb & 1) == 0) ff_out_piece(imgH, f1g, p.FF, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
      }
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
  if constexpr (INFER)
    ff_ln_rows_fwd(imgH, QKV, nrows, wave_s, lane, gm, bt, p.ln_eps, p.y + (int64_t)r0 * FF_D, nullptr, nullptr, nullptr);
  else
    ff_ln_rows_fwd(imgH, QKV, nrows, wave_s, lane, gm, bt, p.ln_eps, p.y + (int64_t)r0 * FF_D, p.s2 + (int64_t)r0 * FF_D,
                   p.mean + r0, p.rstd + r0);
  FF_STAMP(2 + 4 * p.nchunk);
  if constexpr (QKV) {
    // ---- the next layer's in-projection on the rows just normalised (image H = y): per 512-column chunk a GEMM, bias, bf16
    // into image X (free now), whose rows leave for qkv[:, 512 c ..] as the side job of the NEXT chunk's GEMM
    __syncthreads();                     // image H = y complete
    bf16_t* qg = p.qkv + (int64_t)r0 * (3 * FF_D);
    // ATT: wave w's 64 columns of every chunk are head w.  The bf16 values it stores into image X are kept as MFMA operands
    // too: qf / kf[tile][k-step] = rows 16 tile + lane % 16, head columns {32 ks + 4 g + (0..3), 32 ks + 16 + 4 g + (0..3)} --
    // the same permutation of the head dimension on both, which is all a product that reduces over it needs
    bf16x8_t qf[ATT ? 4 : 1][2], pf[ATT ? 4 : 1][2];
    float inv_l[ATT ? 4 : 1];
    // chunk c's GEMM carries the row stores of chunk c - 1 as its side job; the first trip has none (peeled, so that no
    // runtime branch sits around a store inside the k-loop)
    auto qkv_chunk = [&](int c, auto with_store, auto keep) {
      constexpr bool STORE = decltype(with_store)::value != 0;
      constexpr int KEEP = decltype(keep)::value;          // 1: Q -> qf, 2: K -> scores + softmax -> pf, 0: nothing kept
      f32x4_t acq[4][FF_NT];
      ff_zero(acq);
      float4 bq[FF_NT];
      auto sideq = [&](auto, int kb) {
        if constexpr (STORE) {
          if ((kb & 1) == 0) ff_out_piece(imgX, qg + (c - 1) * FF_CHUNK, 3 * FF_D, nrows, FF_TIDV + (kb >> 1) * FF_THREADS);
        }
        if (kb == FF_KB - 1) {
#pragma unroll
          for (int u = 0; u < FF_NT; ++u) bq[u] = *reinterpret_cast<const float4*>(p.bqkv + c * FF_CHUNK + colb + 16 * u);
        }
      };
      ff_gemm<FF_RING_B>(p.wqkvp + (int64_t)c * FF_WELEMS, wave_s * FF_NT, imgH, ff_lane(), acq, sideq);
      FF_STAMP(3 + 4 * p.nchunk + 2 * c);
      __syncthreads();                   // every wave's row pieces of the previous chunk have been read out of image X
      bf16x8_t kf[KEEP == 2 ? 4 : 1][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4_t pk[FF_NT];
#pragma unroll
        for (int u = 0; u < FF_NT; ++u) {
          pk[u] = ff_pack4(acq[mt][u][0] + bq[u].x, acq[mt][u][1] + bq[u].y, acq[mt][u][2] + bq[u].z, acq[mt][u][3] + bq[u].w);
          *reinterpret_cast<bf16x4_t*>(imgX + (16 * mt + l16) * FF_RS + (colb + 16 * u) * 2) = pk[u];
        }
        if constexpr (KEEP == 1) { qf[mt][0] = ff_cat4(pk[0], pk[1]); qf[mt][1] = ff_cat4(pk[2], pk[3]); }
        if constexpr (KEEP == 2) { kf[mt][0] = ff_cat4(pk[0], pk[1]); kf[mt][1] = ff_cat4(pk[2], pk[3]); }
      }
      if constexpr (KEEP == 2) {
        // S^T[key][query] = K . Q^T per (key tile, query tile): the lane owns query 16 it + lane % 16 and, per key tile, the
        // keys 16 jt + 4 g + (0..3); softmax over the keys in registers + two xor-shuffles across the four lane groups;
        // the unnormalised probabilities go straight into the operand form of P . V (attention_mfma.hip)
        float* lse = p.lse_next + ((int64_t)blockIdx.x * FF_HEADS + wave_s) * p.P;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          f32x4_t sc[4];
          float m = -INFINITY;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][0], qf[it][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][1], qf[it][1], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a[r] = (16 * jt + 4 * g + r) < nrows ? a[r] * FF_ATT_SCALE : -INFINITY;
              m = fmaxf(m, a[r]);
            }
            sc[jt] = a;
          }
          m = ff_g4_max(m);
          float l = 0.f;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = __expf(sc[jt][r] - m);      // -inf -> 0
              l += pv;
              sc[jt][r] = pv;
            }
          l = ff_g4_sum(l);
          inv_l[it] = 1.f / l;
          pf[it][0] = ff_acc_frag(sc[0], sc[1]);
          pf[it][1] = ff_acc_frag(sc[2], sc[3]);
          const int q = 16 * it + l16;
          if (g == 0 && q < nrows) lse[q] = m + logf(l);
        }
      }
      __syncthreads();                   // image X = qkv chunk c complete
      FF_STAMP(4 + 4 * p.nchunk + 2 * c);
    };
    qkv_chunk(0, FfIntC<0>{}, FfIntC<ATT ? 1 : 0>{});
    qkv_chunk(1, FfIntC<1>{}, FfIntC<ATT ? 2 : 0>{});
    qkv_chunk(2, FfIntC<1>{}, FfIntC<0>{});
    if constexpr (ATT) {
      // O^T[d][query] = V^T . P^T: V's transposed fragments from the wave's own columns of image X (all eight, once), P from
      // the registers; the normalised rows -> this wave's columns of image H (y is no longer read: every wave is past the
      // last GEMM), from where they leave as whole rows beside V's
      const unsigned char* slV = imgX + 128 * wave_s;
      bf16x8_t vt[2][4];
#pragma unroll
      for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vt[kp][dt] = ff_sl_tr(slV, 32 * kp, 32 * kp + 16, dt, lane);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[0][dt], pf[it][0], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[1][dt], pf[it][1], o[dt], 0, 0, 0);
          *reinterpret_cast<bf16x4_t*>(imgH + (16 * it + l16) * FF_RS + (colb + 16 * dt) * 2) =
              ff_pack4(o[dt][0] * inv_l[it], o[dt][1] * inv_l[it], o[dt][2] * inv_l[it], o[dt][3] * inv_l[it]);
        }
      }
      __syncthreads();                   // image H = the attention output of all eight heads
      bf16_t* ag = p.attn_next + (int64_t)r0 * FF_D;
#pragma unroll
      for (int j = 0; j < 8; ++j) ff_out_piece(imgH, ag, FF_D, nrows, tid + j * FF_THREADS);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ff_out_piece(imgX, qg + 2 * FF_CHUNK, 3 * FF_D, nrows, tid + j * FF_THREADS);
    FF_STAMP(9 + 4 * p.nchunk);
  }
}


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

}  // namespace

#ifdef IB_AB
extern "C" int ib_debug_set_ffn_prof(void* buf) { g_ffn_prof = reinterpret_cast<long long*>(buf); return IB_OK; }
#else
extern "C" int ib_debug_set_ffn_prof(void*) { return IB_E_UNSUPPORTED; }       // measurement builds only
#endif

extern "C" int ib_ffn_chain_supported(int64_t d, int64_t ffn) {
  return (d == FF_D && ffn > 0 && ffn % FF_CHUNK == 0 && ffn / FF_CHUNK <= FF_MAXCHUNK) ? 1 : 0;
}
// elements of ONE layer's packed image: {W1 chunks | W2 chunks | W2^T chunks | W1^T chunks | Wo | Wo^T | Wqkv x 3 | Wqkv^T x 3},
// each 512 x 512
extern "C" size_t ib_ffn_chain_packed_elems(int64_t d, int64_t ffn) {
  return ib_ffn_chain_supported(d, ffn) ? (size_t)((4 * (ffn / FF_CHUNK) + 8) * FF_WELEMS) : 0;
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
// the launches with the attention inside: panels of exactly one window of T frames (0 workgroups: not supported)
extern "C" int ib_ffn_chain_attn_workgroups(int64_t M, int64_t d, int64_t ffn, int64_t T) {
  int P = 0, nc = 0;
  return T > 0 ? ffn_geometry(M, d, ffn, &P, &nc, T) : 0;
}
extern "C" size_t ib_ffn_chain_attn_mask_bytes(int64_t M, int64_t d, int64_t ffn, int64_t T) {
  int P = 0, nc = 0;
  const int n = T > 0 ? ffn_geometry(M, d, ffn, &P, &nc, T) : 0;
  return (size_t)n * nc * FF_THREADS * sizeof(uint2);
}

extern "C" int ib_ffn_chain_pack(const void* const* w1, const int64_t* ld1, const void* const* w2, const int64_t* ld2,
                                 const void* const* wo, const int64_t* ldo, const void* const* wqkv, const int64_t* ldq,
                                 void* const* packed, int layers, int64_t d, int64_t ffn, ib_stream_t stream) {
  if (!w1 || !ld1 || !w2 || !ld2 || !packed || layers < 1 || !ib_ffn_chain_supported(d, ffn)) return IB_E_ARG;
  const int nc = (int)(ffn / FF_CHUNK);
  if (layers * (4 * nc + 8) > FF_MAXDESC) return IB_E_UNSUPPORTED;
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
    if (wqkv && wqkv[l]) {               // this layer's in-projection [3 d, d]: three forward chunks, three transposed chunks
      if (!ldq || ldq[l] < d || ldq[l] % 8 || reinterpret_cast<uintptr_t>(wqkv[l]) % 16) return IB_E_ARG;
      for (int q = 0; q < 2; ++q)
        for (int k = 0; k < 3; ++k) {
          FfnPackDesc& e = pp.d[c++];
          // q = 0: W_eff[n][kk] = Wqkv[512 k + n][kk] (qkv chunk k from y);  q = 1: W_eff[n][kk] = Wqkv[512 k + kk][n] (dy from dqkv chunk k)
          e.src = reinterpret_cast<const bf16_t*>(wqkv[l]) + (int64_t)FF_CHUNK * k * ldq[l]; e.ld = ldq[l]; e.transpose = q;
          e.dst = dst + (int64_t)(4 * nc + 2 + 3 * q + k) * FF_WELEMS;
        }
    }
  }
  pp.count = c;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(ib_grid_1d((int64_t)c * FF_BLOCKS_PER_DESC, 4, 2048)), dim3(256), 0, ib_s(stream), pp);
  IB_CHECK_LAUNCH();
  return IB_OK;
}


namespace {
int ffn_chain_fwd_launch(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                         const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                         const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1,
                         void* x1_out, float* mean1, float* rstd1, const void* packed_next, const float* bqkv_next,
                         void* qkv_next, void* attn_next, float* lse_next, int64_t T, int64_t M, int64_t d, int64_t ffn,
                         float ln_eps, ib_stream_t stream) {
  FfnFwdParams p{};
  int P = 0, nc = 0;
  const bool att = attn_next != nullptr;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc, T);
  if (!nwg) return IB_E_UNSUPPORTED;
  if (!x1 || !packed || !b1 || !b2 || !gamma || !beta || !f1 || !s2 || !y || !mean || !rstd || !mask) return IB_E_ARG;
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
  const bool tail = qkv_next != nullptr;
  if (tail && (!out || !packed_next || !bqkv_next || !ff_al16({packed_next, bqkv_next, qkv_next}))) return IB_E_ARG;
  if (att && (!tail || !lse_next || T <= 0 || P != T || !ff_al16({attn_next}))) return IB_E_ARG;
  // the packed images of both layers are addressed with THIS layer's chunk count: the neighbour must have the same width
  p.wqkvp = tail ? reinterpret_cast<const bf16_t*>(packed_next) + (int64_t)(4 * nc + 2) * FF_WELEMS : nullptr;
  p.bqkv = bqkv_next; p.qkv = (bf16_t*)qkv_next;
  p.attn_next = (bf16_t*)attn_next; p.lse_next = lse_next;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc; p.ln_eps = ln_eps;
  p.prof = IB_AB_PROF(g_ffn_prof);
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (att) hipLaunchKernelGGL((ffn_chain_fwd_kernel<true, true, true>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (tail) hipLaunchKernelGGL((ffn_chain_fwd_kernel<true, true, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else if (out) hipLaunchKernelGGL((ffn_chain_fwd_kernel<true, false, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL((ffn_chain_fwd_kernel<false, false, false>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
}  // namespace

extern "C" int ib_ffn_chain_fwd(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                                const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                                const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1,
                                void* x1_out, float* mean1, float* rstd1, const void* packed_next, const float* bqkv_next,
                                void* qkv_next, int64_t M, int64_t d, int64_t ffn, float ln_eps, ib_stream_t stream) {
  return ffn_chain_fwd_launch(x1, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask, attn, bo, gamma1, beta1, s1, x1_out,
                              mean1, rstd1, packed_next, bqkv_next, qkv_next, nullptr, nullptr, 0, M, d, ffn, ln_eps, stream);
}
// The frozen-weight forward of a layer's token-local half in ONE launch (DDIM sampler at batches beyond the row-panel
// kernels; TransformerBaseline.py:12-19,29-36 forward only): y = LN2(x1 + W2 relu(W1 x1 + b1) + b2) with
// x1 = LN1(x + attn Wo^T + bo), and -- qkv_next != NULL -- the NEXT layer's in-projection of y.  Nothing is saved.
extern "C" int ib_ffn_chain_fwd_infer(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                                      const float* beta, void* y, const void* attn, const float* bo, const float* gamma1,
                                      const float* beta1, const void* packed_next, const float* bqkv_next, void* qkv_next,
                                      int64_t M, int64_t d, int64_t ffn, float ln_eps, ib_stream_t stream) {
  FfnFwdParams p{};
  int P = 0, nc = 0;
  const int nwg = ffn_geometry(M, d, ffn, &P, &nc);
  if (!nwg) return IB_E_UNSUPPORTED;
  if (!x || !packed || !b1 || !b2 || !gamma || !beta || !y || !attn || !bo || !gamma1 || !beta1) return IB_E_ARG;
  if (!ff_al16({x, packed, b1, b2, gamma, beta, y, attn, bo, gamma1, beta1})) return IB_E_ARG;
  const bool tail = qkv_next != nullptr;
  if (tail && (!packed_next || !bqkv_next || !ff_al16({packed_next, bqkv_next, qkv_next}))) return IB_E_ARG;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  p.x1 = (const bf16_t*)x; p.w1p = pk; p.w2p = pk + (int64_t)nc * FF_WELEMS;
  p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.y = (bf16_t*)y;
  p.attn = (const bf16_t*)attn; p.wop = pk + (int64_t)4 * nc * FF_WELEMS; p.bo = bo; p.gamma1 = gamma1; p.beta1 = beta1;
  p.wqkvp = tail ? reinterpret_cast<const bf16_t*>(packed_next) + (int64_t)(4 * nc + 2) * FF_WELEMS : nullptr;
  p.bqkv = bqkv_next; p.qkv = (bf16_t*)qkv_next;
  p.M = (int)M; p.P = P; p.FF = (int)ffn; p.nchunk = nc; p.ln_eps = ln_eps;
  p.prof = nullptr;
  IB_PATH(IB_PATH_FFN_CHAIN);
  if (tail) hipLaunchKernelGGL((ffn_chain_fwd_kernel<true, true, false, true>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  else hipLaunchKernelGGL((ffn_chain_fwd_kernel<true, false, false, true>), dim3(nwg), dim3(FF_THREADS), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
// The same launch over panels of exactly one window of T frames (16 <= T <= 64; the geometry of a layer whose attention rides
// inside its launches -- its backward uses the same panels) and, with attn_next != NULL, the NEXT layer's temporal
// self-attention behind the QKV tail (eight heads of 64): attn_next [M, 512], lse_next [M / T, 8, T]
// (TransformerBaseline.py:12-13,29).
extern "C" int ib_ffn_chain_fwd_attn(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                                     const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                                     const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1,
                                     void* x1_out, float* mean1, float* rstd1, const void* packed_next,
                                     const float* bqkv_next, void* qkv_next, void* attn_next, float* lse_next, int64_t T,
                                     int64_t M, int64_t d, int64_t ffn, float ln_eps, ib_stream_t stream) {
  if (T <= 0) return IB_E_ARG;
  return ffn_chain_fwd_launch(x, packed, b1, b2, gamma, beta, f1, s2, y, mean, rstd, mask, attn, bo, gamma1, beta1, s1, x1_out,
                              mean1, rstd1, packed_next, bqkv_next, qkv_next, attn_next, lse_next, T, M, d, ffn, ln_eps, stream);
}
