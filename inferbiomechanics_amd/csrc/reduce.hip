// Fixed-order reductions that finish a training step's gradients (deterministic: no float atomics anywhere):
//   slab sets   : split-M partial slabs of every weight-gradient GEMM -> dW              (ib_slab_reduce_multi)
//   column sums : per-workgroup partial rows of the chain kernel -> small gradients + loss (ib_colsum_segments)
// and both in ONE launch (ib_step_reduce): each kernel boundary of a captured step costs ~4.5 us.
#include "ib_common.h"

namespace {
inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

constexpr int SR_MAX = 8;
struct SlabMulti {
  const float* slabs[SR_MAX]; float* dw[SR_MAX]; int64_t lddw[SR_MAX];
  int nslab[SR_MAX], rows[SR_MAX], cols[SR_MAX], blk0[SR_MAX + 1];
  int n, accumulate;
};
// several wgrad slab sets -> their gradients in ONE launch (float4 path only: cols % 4 == 0, aligned)
__device__ __forceinline__ void slab_multi_body(const SlabMulti& p, int bid) {
  int e = 0;
  for (int j = 1; j < p.n; ++j)
    if (bid >= p.blk0[j]) e = j;
  const int nb = p.blk0[e + 1] - p.blk0[e];
  const int cols = p.cols[e], nslab = p.nslab[e];
  const int64_t n4 = ((int64_t)p.rows[e] * cols) >> 2, st4 = n4;
  const float4* base = reinterpret_cast<const float4*>(p.slabs[e]);
  for (int64_t e4 = (int64_t)(bid - p.blk0[e]) * 256 + threadIdx.x; e4 < n4; e4 += (int64_t)nb * 256) {
    const float4* q = base + e4;
    float4 s = ib_slab_sum4(q, st4, nslab);
    const int64_t el = e4 << 2;
    const int r = (int)(el / cols), c0 = (int)(el % cols);
    float4* o = reinterpret_cast<float4*>(p.dw[e] + (int64_t)r * p.lddw[e] + c0);
    if (p.accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    *o = s;
  }
}

// ---- multi-segment column sums: out_s[c] (+)= scale_s * sum_r part[r][col0_s + c]   (fixed order, one launch for
// every small parameter gradient of a step: LayerNorm gains / biases, linear biases, the loss scalar)
constexpr int CS_MAXSEG = 24;
struct ColsumSegs {
  const float* part; int64_t ld; int rows; int nseg; int accumulate;
  int col0[CS_MAXSEG], ncols[CS_MAXSEG];
  float* dst[CS_MAXSEG]; float* dst2[CS_MAXSEG];
  float scale[CS_MAXSEG];
  int blk0[CS_MAXSEG + 1];       // first block of each segment (64 columns per block)
  // optional per-segment partial matrices (ib_step_reduce_parts): segment j sums rowsv[j] rows of partv[j] (pitch ldv[j])
  const float* partv[CS_MAXSEG]; int64_t ldv[CS_MAXSEG]; int rowsv[CS_MAXSEG];
};
__device__ __forceinline__ void colsum_segs_body(const ColsumSegs& p, int bid, float4 (&red)[16][16]) {
  int sgi = 0;
  for (int j = 1; j < p.nseg; ++j)
    if (bid >= p.blk0[j]) sgi = j;
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;      // 16 float4 columns x 16 row groups
  const int c = (bid - p.blk0[sgi]) * 64 + 4 * c4;
  const int nc = p.ncols[sgi];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < nc) {
    const bool own = p.partv[sgi] != nullptr;
    const float* base = (own ? p.partv[sgi] : p.part) + p.col0[sgi] + c;
    const int64_t ld = own ? p.ldv[sgi] : p.ld;
    const int rows = own ? p.rowsv[sgi] : p.rows;
    if (c + 4 <= nc) {
      s = ib_rows_sum4(base, ld, rg, 16, rows);
    } else {
      for (int r = rg; r < rows; r += 16) {
        const float* q = base + (int64_t)r * ld;
        s.x += q[0];
        if (c + 1 < nc) s.y += q[1];
        if (c + 2 < nc) s.z += q[2];
      }
    }
  }
  red[rg][c4] = s;
  __syncthreads();
  if (rg == 0 && c < nc) {
    float4 t = red[0][c4];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const float sc = p.scale[sgi];
    const float o[4] = {t.x * sc, t.y * sc, t.z * sc, t.w * sc};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k < nc) {
        float* d = p.dst[sgi] + c + k;
        *d = p.accumulate ? *d + o[k] : o[k];
        if (p.dst2[sgi]) { float* d2 = p.dst2[sgi] + c + k; *d2 = p.accumulate ? *d2 + o[k] : o[k]; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(SlabMulti p) { slab_multi_body(p, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void colsum_segments_kernel(ColsumSegs p) {
  __shared__ float4 red[16][16];
  colsum_segs_body(p, (int)blockIdx.x, red);
}
// blocks [0, slab_blocks) reduce slabs, the rest sum columns
__global__ __launch_bounds__(256) void step_reduce_kernel(SlabMulti sp, ColsumSegs cp, int slab_blocks) {
  __shared__ float4 red[16][16];
  if ((int)blockIdx.x < slab_blocks) slab_multi_body(sp, (int)blockIdx.x);
  else colsum_segs_body(cp, (int)blockIdx.x - slab_blocks, red);
}

static int build_slab_multi(int n, const void* const* slabs, const int32_t* nslab, float* const* dw, const int64_t* lddw,
                            const int32_t* N, const int32_t* K, int accumulate, SlabMulti& p, int* blocks_out) {
  if (n <= 0 || n > SR_MAX || !slabs || !nslab || !dw || !lddw || !N || !K) return IB_E_ARG;
  p = SlabMulti{};
  p.n = n; p.accumulate = accumulate;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    if (!slabs[j] || !dw[j] || nslab[j] <= 0 || N[j] <= 0 || K[j] <= 0 || lddw[j] < K[j]) return IB_E_ARG;
    if (K[j] % 4 != 0 || lddw[j] % 4 != 0 || !aligned(slabs[j], 16) || !aligned(dw[j], 16)) return IB_E_UNSUPPORTED;
    p.slabs[j] = reinterpret_cast<const float*>(slabs[j]); p.dw[j] = dw[j]; p.lddw[j] = lddw[j];
    p.nslab[j] = nslab[j]; p.rows[j] = N[j]; p.cols[j] = K[j];
    p.blk0[j] = blocks;
    blocks += ib_grid_1d((int64_t)N[j] * K[j] / 4, 256, 1024);
  }
  p.blk0[n] = blocks;
  *blocks_out = blocks;
  return IB_OK;
}

static int build_colsum_segs(const float* part, int64_t ld, int64_t rows, int nseg, const int32_t* col0,
                             const int32_t* ncols, float* const* dst, float* const* dst2, const float* scale,
                             int accumulate, ColsumSegs& p, int* blocks_out) {
  if (!part || rows <= 0 || nseg <= 0 || nseg > CS_MAXSEG || !col0 || !ncols || !dst || ld % 4 != 0) return IB_E_ARG;
  if ((reinterpret_cast<uintptr_t>(part) % 16) != 0) return IB_E_ARG;
  p = ColsumSegs{};
  p.part = part; p.ld = ld; p.rows = (int)rows; p.nseg = nseg; p.accumulate = accumulate;
  int blocks = 0;
  for (int j = 0; j < nseg; ++j) {
    if (!dst[j] || ncols[j] <= 0 || col0[j] < 0 || col0[j] % 4 != 0 || col0[j] + ncols[j] > ld) return IB_E_ARG;
    p.col0[j] = col0[j]; p.ncols[j] = ncols[j]; p.dst[j] = dst[j]; p.dst2[j] = dst2 ? dst2[j] : nullptr;
    p.scale[j] = scale ? scale[j] : 1.f;
    p.blk0[j] = blocks;
    blocks += (ncols[j] + 63) / 64;
  }
  p.blk0[nseg] = blocks;
  *blocks_out = blocks;
  return IB_OK;
}
}  // namespace

extern "C" int ib_slab_reduce_multi(int n, const void* const* slabs, const int32_t* nslab, float* const* dw,
                                    const int64_t* lddw, const int32_t* N, const int32_t* K, int accumulate,
                                    ib_stream_t stream) {
  SlabMulti p;
  int blocks = 0;
  const int rc = build_slab_multi(n, slabs, nslab, dw, lddw, N, K, accumulate, p, &blocks);
  if (rc != IB_OK) return rc;
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(blocks), dim3(256), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_colsum_segments(const float* part, int64_t ld, int64_t rows, int nseg, const int32_t* col0,
                                  const int32_t* ncols, float* const* dst, float* const* dst2, const float* scale,
                                  int accumulate, ib_stream_t stream) {
  ColsumSegs p;
  int blocks = 0;
  const int rc = build_colsum_segs(part, ld, rows, nseg, col0, ncols, dst, dst2, scale, accumulate, p, &blocks);
  if (rc != IB_OK) return rc;
  hipLaunchKernelGGL(colsum_segments_kernel, dim3(blocks), dim3(256), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_step_reduce(int n, const void* const* slabs, const int32_t* nslab, float* const* dw,
                              const int64_t* lddw, const int32_t* N, const int32_t* K, const float* part, int64_t ld,
                              int64_t rows, int nseg, const int32_t* col0, const int32_t* ncols, float* const* dst,
                              float* const* dst2, const float* scale, int accumulate, ib_stream_t stream) {
  SlabMulti sp;
  ColsumSegs cp;
  int sb = 0, cb = 0;
  int rc = build_slab_multi(n, slabs, nslab, dw, lddw, N, K, accumulate, sp, &sb);
  if (rc != IB_OK) return rc;
  rc = build_colsum_segs(part, ld, rows, nseg, col0, ncols, dst, dst2, scale, accumulate, cp, &cb);
  if (rc != IB_OK) return rc;
  hipLaunchKernelGGL(step_reduce_kernel, dim3(sb + cb), dim3(256), 0, ib_s(stream), sp, cp, sb);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// ib_step_reduce with a partial matrix PER SEGMENT (a transformer layer's gradients under data parallelism: the split-M
// slabs of its four weight gradients + the bias / LayerNorm partial sums, each left by a different launch, finished by ONE
// launch right before the layer's bucket is all-reduced).  nseg <= 24; every part[j] 16-byte aligned, ld[j] % 4 == 0.
extern "C" int ib_step_reduce_parts(int n, const void* const* slabs, const int32_t* nslab, float* const* dw,
                                    const int64_t* lddw, const int32_t* N, const int32_t* K, int nseg,
                                    const float* const* part, const int64_t* ld, const int32_t* rows, const int32_t* col0,
                                    const int32_t* ncols, float* const* dst, const float* scale, int accumulate,
                                    ib_stream_t stream) {
  if (nseg < 0 || nseg > CS_MAXSEG || (nseg > 0 && (!part || !ld || !rows || !col0 || !ncols || !dst))) return IB_E_ARG;
  SlabMulti sp{};
  int sb = 0;
  if (n > 0) {
    const int rc = build_slab_multi(n, slabs, nslab, dw, lddw, N, K, accumulate, sp, &sb);
    if (rc != IB_OK) return rc;
  }
  ColsumSegs cp{};
  cp.nseg = nseg; cp.accumulate = accumulate;
  int cb = 0;
  for (int j = 0; j < nseg; ++j) {
    if (!part[j] || !dst[j] || rows[j] <= 0 || ncols[j] <= 0 || col0[j] < 0 || col0[j] % 4 != 0 || ld[j] % 4 != 0 ||
        col0[j] + ncols[j] > ld[j] || (reinterpret_cast<uintptr_t>(part[j]) % 16) != 0)
      return IB_E_ARG;
    cp.partv[j] = part[j]; cp.ldv[j] = ld[j]; cp.rowsv[j] = rows[j];
    cp.col0[j] = col0[j]; cp.ncols[j] = ncols[j]; cp.dst[j] = dst[j]; cp.dst2[j] = nullptr;
    cp.scale[j] = scale ? scale[j] : 1.f;
    cp.blk0[j] = cb;
    cb += (ncols[j] + 63) / 64;
  }
  cp.blk0[nseg] = cb;
  if (sb + cb == 0) return IB_OK;
  hipLaunchKernelGGL(step_reduce_kernel, dim3(sb + cb), dim3(256), 0, ib_s(stream), sp, cp, sb);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
