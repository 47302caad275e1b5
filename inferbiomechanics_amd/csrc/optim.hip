// Fused optimizer step over ONE flat fp32 parameter buffer (one launch per step instead of
// torch.optim's per-tensor foreach kernels): torch.optim.{SGD,Adam,RMSprop,Adagrad,Adadelta,Adamax}
// with only lr set, as src/cli/train.py:183-194 constructs them.  The DDP mean (1/world) is folded
// in as grad_scale; an optional bf16 shadow of the parameters is refreshed in the same pass so the
// bf16 GEMMs never need a separate cast kernel.
#include "ib_common.h"

namespace {

struct OptArgs {
  float* p; const float* g; float* s1; float* s2; bf16_t* shadow;
  int64_t n; float lr; float gscale; int step; int32_t* step_dev; int32_t* ticket; int opt;
};

// Gradient SOURCES (ib_optim_step_sources): ranges of the flat buffer whose gradient is still a set of partial sums when
// the optimizer runs -- the split-M slabs of a weight-gradient GEMM, or per-workgroup partial rows (column sums).  The
// optimizer sums them itself, in the same fixed order as ib_step_reduce, so the reduction launch, its kernel boundary
// and the round trip of the reduced gradient through HBM disappear (single-GPU steps only: an all-reduce needs the
// reduced gradient in memory).
constexpr int OPT_TICKET_SUBS = 32, OPT_TICKET_LINE = 32;     // sub-counters, int32 words per 128-byte line
constexpr int OPT_MAXSRC = 64;   // 4 transformer layers x (4 slab sets + 4 bias sums + 4 LayerNorm sums) + the projections
struct GradSrc {
  int64_t start, len;          // flat element range (start % 4 == 0)
  const float* base;           // slabs: [nslab][len] ; column sums: part + col0
  int64_t stride;              // slabs: elements between slabs ; column sums: row pitch of part
  int count;                   // slabs: nslab ; column sums: rows
  int kind;                    // 1 = slabs, 2 = column sums, 3 = none: the range was updated by an earlier launch of this step
  float scale;
};
struct OptSources {
  GradSrc s[OPT_MAXSRC]; int n;
  const float* loss_col; int64_t loss_ld; int loss_rows; float loss_scale; float* loss_out;   // optional scalar
};

__device__ __forceinline__ float opt_update(const OptArgs& a, float p, float g, float& s1, float& s2, float bc1,
                                            float bc2s) {
  switch (a.opt) {
    case IB_OPT_SGD: return p - a.lr * g;
    case IB_OPT_ADAM: {
      s1 = 0.9f * s1 + 0.1f * g;                 // exp_avg
      s2 = 0.999f * s2 + 0.001f * g * g;         // exp_avg_sq
      const float denom = sqrtf(s2) / bc2s + 1e-8f;
      return p - (a.lr / bc1) * (s1 / denom);
    }
    case IB_OPT_RMSPROP: {
      s1 = 0.99f * s1 + 0.01f * g * g;           // square_avg
      return p - a.lr * (g / (sqrtf(s1) + 1e-8f));
    }
    case IB_OPT_ADAGRAD: {
      s1 = s1 + g * g;
      return p - a.lr * (g / (sqrtf(s1) + 1e-10f));
    }
    case IB_OPT_ADADELTA: {
      s1 = 0.9f * s1 + 0.1f * g * g;             // square_avg
      const float delta = sqrtf(s2 + 1e-6f) / sqrtf(s1 + 1e-6f) * g;
      s2 = 0.9f * s2 + 0.1f * delta * delta;     // acc_delta
      return p - a.lr * delta;
    }
    case IB_OPT_ADAMAX: {
      s1 = 0.9f * s1 + 0.1f * g;                 // exp_avg
      s2 = fmaxf(0.999f * s2, fabsf(g) + 1e-8f); // exp_inf
      return p - (a.lr / bc1) * (s1 / s2);
    }
    default: return p;
  }
}

// slab-backed ranges are summed by whichever thread owns the element; column-sum ranges are left to the dedicated
// cooperative blocks below (a single thread summing 256 strided rows was a 100-us tail)
// (the host passes the sources sorted by start).  Two things made this path 40 us of the transformer's 111-us launch
// beyond its bytes (48 sources, 126 MB of slabs): every thread walked the source list from its head -- up to 48 dependent
// scalar loads before its first vector load -- and a slab count below 8 (2 for most transformer weights) fell into a
// rolled one-load-per-trip loop, i.e. `count` HBM round trips in sequence.  Now the wave finds its first candidate by a
// UNIFORM binary search (6 scalar steps) and every batch of up to 8 slabs is requested at once (predicated on the count), the sum still runs strictly in slab order
// (= ib_step_reduce / slab_reduce_multi, bitwise).
__device__ __forceinline__ bool source_grad4(const OptSources& S, const float* g, int64_t idx, float4& t) {
  const int64_t idx0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(idx >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(idx & 0xffffffff));
  int lo = 0, hi = S.n;                        // first source whose end lies beyond the wave's first element
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (S.s[mid].start + S.s[mid].len <= idx0) lo = mid + 1; else hi = mid;
  }
  for (int j = lo; j < S.n; ++j) {
    const GradSrc& r = S.s[j];
    if (idx < r.start) break;
    if (idx < r.start + r.len) {
      if (r.kind != 1) return false;
      const int64_t o = idx - r.start;
      t = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* q = reinterpret_cast<const float4*>(r.base + o);
      const int64_t st4 = r.stride >> 2;
      for (int k = 0; k < r.count; k += 8) {  // up to 8 slabs in flight, added strictly in sequence
        float4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k + e < r.count) v[e] = q[(int64_t)(k + e) * st4];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k + e < r.count) { t.x += v[e].x; t.y += v[e].y; t.z += v[e].z; t.w += v[e].w; }
      }
      return true;
    }
  }
  t = *reinterpret_cast<const float4*>(g + idx);
  return true;
}

__device__ __forceinline__ void apply4(const OptArgs& a, int64_t i, float4 g, float bc1, float bc2s, int nvalid) {
  // nvalid < 4 only at the ragged end of a column-sum range (the remaining lanes belong to alignment padding)
  const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr;
  float4 p = reinterpret_cast<float4*>(a.p)[i];
  float4 s1 = has1 ? reinterpret_cast<float4*>(a.s1)[i] : make_float4(0, 0, 0, 0);
  float4 s2 = has2 ? reinterpret_cast<float4*>(a.s2)[i] : make_float4(0, 0, 0, 0);
  const float4 p0 = p, s10 = s1, s20 = s2;
  p.x = opt_update(a, p.x, g.x * a.gscale, s1.x, s2.x, bc1, bc2s);
  p.y = opt_update(a, p.y, g.y * a.gscale, s1.y, s2.y, bc1, bc2s);
  p.z = opt_update(a, p.z, g.z * a.gscale, s1.z, s2.z, bc1, bc2s);
  p.w = opt_update(a, p.w, g.w * a.gscale, s1.w, s2.w, bc1, bc2s);
  if (nvalid < 4) { p.w = p0.w; s1.w = s10.w; s2.w = s20.w; }
  if (nvalid < 3) { p.z = p0.z; s1.z = s10.z; s2.z = s20.z; }
  if (nvalid < 2) { p.y = p0.y; s1.y = s10.y; s2.y = s20.y; }
  reinterpret_cast<float4*>(a.p)[i] = p;
  if (has1) reinterpret_cast<float4*>(a.s1)[i] = s1;
  if (has2) reinterpret_cast<float4*>(a.s2)[i] = s2;
  if (a.shadow) {
    bf16x4_t o;
    o[0] = (bf16_t)p.x; o[1] = (bf16_t)p.y; o[2] = (bf16_t)p.z; o[3] = (bf16_t)p.w;
    reinterpret_cast<bf16x4_t*>(a.shadow)[i] = o;
  }
}

// SRC: blocks [0, main_blocks) walk the flat buffer (skipping column-sum ranges); block main_blocks + b owns 64 columns of
// a column-sum range: 16 float4 columns x 16 row groups, LDS combine in the order of colsum_segments_kernel, update.
template <bool SRC>
__global__ __launch_bounds__(256) void optim_kernel(OptArgs a, OptSources S, int main_blocks) {
  // self-counting mode (ticket != NULL): *step_dev holds the number of COMPLETED steps; every block reads it on
  // entry, and the block that draws the last exit ticket publishes step and resets the ticket -- it exits after
  // every other block has entered (and therefore read the old value), so no block can see the new count.
  // step_dev without a ticket: step = *step_dev + a.step, nothing published -- a.step = 1 is how a launch over PART of
  // the buffer (a layer's range, issued as soon as its gradient is complete) takes part in a self-counting step whose
  // last launch publishes.
  const int step = a.step_dev ? *(volatile int32_t*)a.step_dev + (a.ticket ? 1 : a.step) : a.step;
  // bias corrections in double (torch computes them as Python floats) -- Adam / Adamax only: two double-precision pow()
  // are several hundred instructions at the head of every thread, ahead of its first load
  float bc1 = 1.f, bc2s = 1.f;
  if (a.opt == IB_OPT_ADAM || a.opt == IB_OPT_ADAMAX) {
    bc1 = (float)(1.0 - pow(0.9, (double)step));
    bc2s = (float)sqrt(1.0 - pow(0.999, (double)step));
  }
  const int64_t n4 = a.n / 4;
  const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr;
  // the column-sum blocks are long dependent chains (256 strided rows each): they take the FIRST physical block ids so
  // they start at once and overlap the streaming blocks instead of forming the launch's tail
  const int ncs = SRC ? (int)gridDim.x - main_blocks : 0;
  const int lb = SRC ? ((int)blockIdx.x < ncs ? main_blocks + (int)blockIdx.x : (int)blockIdx.x - ncs) : (int)blockIdx.x;
  if (SRC && lb >= main_blocks) {
    __shared__ float4 red[16][16];
    int rb = lb - main_blocks, j = 0;
    for (; j < S.n; ++j) {                      // which column-sum range, which 64-column chunk of it
      if (S.s[j].kind != 2) continue;
      const int nb = (int)((S.s[j].len + 63) / 64);
      if (rb < nb) break;
      rb -= nb;
    }
    if (j < S.n) {
      const GradSrc& r = S.s[j];
      const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
      const int64_t o = (int64_t)rb * 64 + 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o < r.len) {
        const float* q = r.base + o;
        if (o + 4 <= r.len) {
          // rows rg, rg + 16, ...: batches of 8 requested together, added strictly in row order (= colsum_segments_kernel)
          v = ib_rows_sum4(q, r.stride, rg, 16, r.count);
        } else {
          for (int row = rg; row < r.count; row += 16) {
            const float* w = q + (int64_t)row * r.stride;
            v.x += w[0];
            if (o + 1 < r.len) v.y += w[1];
            if (o + 2 < r.len) v.z += w[2];
          }
        }
      }
      red[rg][c4] = v;
      __syncthreads();
      if (rg == 0 && o < r.len) {
        float4 t = red[0][c4];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 w = red[k][c4]; t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w; }
        t.x *= r.scale; t.y *= r.scale; t.z *= r.scale; t.w *= r.scale;
        apply4(a, (r.start + o) >> 2, t, bc1, bc2s, (int)min((int64_t)4, r.len - o));
      }
    }
  } else {
  const int64_t gstride = (int64_t)(SRC ? main_blocks : (int)gridDim.x) * blockDim.x;
  for (int64_t i = (int64_t)lb * blockDim.x + threadIdx.x; i < n4; i += gstride) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    float4 g;
    if constexpr (SRC) {
      if (!source_grad4(S, a.g, i * 4, g)) continue;
    } else {
      g = reinterpret_cast<const float4*>(a.g)[i];
    }
    float4 s1 = has1 ? reinterpret_cast<float4*>(a.s1)[i] : make_float4(0, 0, 0, 0);
    float4 s2 = has2 ? reinterpret_cast<float4*>(a.s2)[i] : make_float4(0, 0, 0, 0);
    p.x = opt_update(a, p.x, g.x * a.gscale, s1.x, s2.x, bc1, bc2s);
    p.y = opt_update(a, p.y, g.y * a.gscale, s1.y, s2.y, bc1, bc2s);
    p.z = opt_update(a, p.z, g.z * a.gscale, s1.z, s2.z, bc1, bc2s);
    p.w = opt_update(a, p.w, g.w * a.gscale, s1.w, s2.w, bc1, bc2s);
    reinterpret_cast<float4*>(a.p)[i] = p;
    if (has1) reinterpret_cast<float4*>(a.s1)[i] = s1;
    if (has2) reinterpret_cast<float4*>(a.s2)[i] = s2;
    if (a.shadow) {
      bf16x4_t o;
      o[0] = (bf16_t)p.x; o[1] = (bf16_t)p.y; o[2] = (bf16_t)p.z; o[3] = (bf16_t)p.w;
      reinterpret_cast<bf16x4_t*>(a.shadow)[i] = o;
    }
  }
  // tail (n % 4)
  const int64_t t0 = n4 * 4;
  for (int64_t i = t0 + (int64_t)lb * blockDim.x + threadIdx.x; i < a.n; i += gstride) {
    float s1 = has1 ? a.s1[i] : 0.f, s2 = has2 ? a.s2[i] : 0.f;
    const float p = opt_update(a, a.p[i], a.g[i] * a.gscale, s1, s2, bc1, bc2s);
    a.p[i] = p;
    if (has1) a.s1[i] = s1;
    if (has2) a.s2[i] = s2;
    if (a.shadow) a.shadow[i] = (bf16_t)p;
  }
  }   // main blocks
  if constexpr (SRC) {
    if (S.loss_out && lb == 0) {       // the step's loss scalar: one column of the partial rows, fixed order
      __shared__ float lred[16];
      float v = 0.f;
      const int rg = threadIdx.x >> 4;
      if ((threadIdx.x & 15) == 0)
        for (int r = rg; r < S.loss_rows; r += 16) v += S.loss_col[(int64_t)r * S.loss_ld];
      if ((threadIdx.x & 15) == 0) lred[rg] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = lred[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += lred[k];
        *S.loss_out = t * S.loss_scale;
      }
    }
  }
  if (a.ticket) {
    // Exit tickets, two levels: 1137 blocks (the MLP denoiser's flat buffer) drawing from ONE word were 1137 returning
    // atomics on one address, serialised in L2 at ~11 ns each -- 12 of the launch's 17 us.  Block b draws from sub-counter
    // b % 32 (its own 128-byte line: different lines pipeline); the last arriver of a sub-counter resets it and draws from
    // the top word; the last of those publishes.  The longest same-address chain is grid / 32 + 32 draws.
    __syncthreads();
    if (threadIdx.x == 0) {
      const int G = (int)gridDim.x;
      const int nsub = G < OPT_TICKET_SUBS ? G : OPT_TICKET_SUBS;
      const int sub = (int)blockIdx.x % OPT_TICKET_SUBS;
      const int quota = (G - sub + OPT_TICKET_SUBS - 1) / OPT_TICKET_SUBS;
      int32_t* sc = a.ticket + OPT_TICKET_LINE * (1 + sub);
      if (atomicAdd(sc, 1) == quota - 1) {
        *sc = 0;
        if (atomicAdd(a.ticket, 1) == nsub - 1) {
          *a.step_dev = step;
          *a.ticket = 0;
        }
      }
    }
  }
}

}  // namespace

extern "C" int ib_optim_ticket_words(void) { return OPT_TICKET_LINE * (1 + OPT_TICKET_SUBS); }

extern "C" int ib_optim_step(int opt, float* p, const float* g, float* s1, float* s2, int64_t n, float lr,
                             float grad_scale, int32_t step, int32_t* step_dev, int32_t* ticket, void* shadow_bf16,
                             ib_stream_t stream) {
  if (!p || !g || n <= 0 || opt < IB_OPT_SGD || opt > IB_OPT_ADAMAX) return IB_E_ARG;
  if (ticket && !step_dev) return IB_E_ARG;
  const bool need1 = opt != IB_OPT_SGD;
  const bool need2 = (opt == IB_OPT_ADAM || opt == IB_OPT_ADADELTA || opt == IB_OPT_ADAMAX);
  if ((need1 && !s1) || (need2 && !s2)) return IB_E_ARG;
  auto al16 = [](const void* q) { return !q || (reinterpret_cast<uintptr_t>(q) % 16) == 0; };
  if (!al16(p) || !al16(g) || !al16(s1) || !al16(s2) || (shadow_bf16 && reinterpret_cast<uintptr_t>(shadow_bf16) % 8))
    return IB_E_ARG;
  OptArgs a{p, g, need1 ? s1 : nullptr, need2 ? s2 : nullptr, reinterpret_cast<bf16_t*>(shadow_bf16),
            n, lr, grad_scale, step, step_dev, ticket, opt};
  hipLaunchKernelGGL(optim_kernel<false>, dim3(ib_grid_1d(n / 4 + 1, 256)), dim3(256), 0, ib_s(stream), a, OptSources{}, 0);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_optim_step_sources(int opt, float* p, const float* g, float* s1, float* s2, int64_t n, float lr,
                                     float grad_scale, int32_t step, int32_t* step_dev, int32_t* ticket,
                                     void* shadow_bf16, int nsrc, const int64_t* start, const int64_t* len,
                                     const int32_t* kind, const void* const* base, const int64_t* stride,
                                     const int32_t* count, const float* scale, const float* loss_col, int64_t loss_ld,
                                     int64_t loss_rows, float loss_scale, float* loss_out, ib_stream_t stream) {
  if (!p || !g || n <= 0 || opt < IB_OPT_SGD || opt > IB_OPT_ADAMAX) return IB_E_ARG;
  if (ticket && !step_dev) return IB_E_ARG;
  if (nsrc < 0 || nsrc > OPT_MAXSRC || (nsrc > 0 && (!start || !len || !kind || !base || !stride || !count))) return IB_E_ARG;
  if (n % 4 != 0) return IB_E_UNSUPPORTED;
  const bool need1 = opt != IB_OPT_SGD;
  const bool need2 = (opt == IB_OPT_ADAM || opt == IB_OPT_ADADELTA || opt == IB_OPT_ADAMAX);
  if ((need1 && !s1) || (need2 && !s2)) return IB_E_ARG;
  auto al16 = [](const void* q) { return !q || (reinterpret_cast<uintptr_t>(q) % 16) == 0; };
  if (!al16(p) || !al16(g) || !al16(s1) || !al16(s2) || (shadow_bf16 && reinterpret_cast<uintptr_t>(shadow_bf16) % 8))
    return IB_E_ARG;
  OptSources S{};
  S.n = nsrc;
  for (int j = 0; j < nsrc; ++j) {
    if (start[j] < 0 || len[j] <= 0 || start[j] % 4 != 0 || start[j] + len[j] > n) return IB_E_ARG;
    if (kind[j] == 3) {                                     // skipped by every block
      if (len[j] % 4 != 0) return IB_E_ARG;
      S.s[j] = GradSrc{start[j], len[j], nullptr, 0, 0, 3, 1.f};
      continue;
    }
    if (!base[j] || count[j] <= 0) return IB_E_ARG;
    if (kind[j] == 1) {
      if (len[j] % 4 != 0 || stride[j] % 4 != 0 || !al16(base[j])) return IB_E_ARG;
    } else if (kind[j] == 2) {
      if (stride[j] % 4 != 0 || !al16(base[j])) return IB_E_ARG;
    } else return IB_E_ARG;
    S.s[j] = GradSrc{start[j], len[j], reinterpret_cast<const float*>(base[j]), stride[j], count[j], kind[j],
                     scale ? scale[j] : 1.f};
  }
  if (loss_out) {
    if (!loss_col || loss_rows <= 0) return IB_E_ARG;
    S.loss_col = loss_col; S.loss_ld = loss_ld; S.loss_rows = (int)loss_rows; S.loss_scale = loss_scale; S.loss_out = loss_out;
  }
  OptArgs a{p, g, need1 ? s1 : nullptr, need2 ? s2 : nullptr, reinterpret_cast<bf16_t*>(shadow_bf16),
            n, lr, grad_scale, step, step_dev, ticket, opt};
  const int main_blocks = ib_grid_1d(n / 4 + 1, 256);
  int cs_blocks = 0;
  for (int j = 0; j < nsrc; ++j)
    if (kind[j] == 2) cs_blocks += (int)((len[j] + 63) / 64);
  hipLaunchKernelGGL(optim_kernel<true>, dim3(main_blocks + cs_blocks), dim3(256), 0, ib_s(stream), a, S, main_blocks);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
