// Groundlink's convolution stack (src/models/Groundlink.py:41-48) as GEMMs over an explicit
// im2col: Conv1d(C_in -> C_out, k = 7, padding = 3, padding_mode = "replicate") on a window of F frames is
//     y[(n,f)][o] = b[o] + sum_{c,j} w[o][c][j] * x[(n, clamp(f + j - 3, 0, F-1))][c]
// = col[N*F, C_in*k] . w.view(C_out, C_in*k)^T with col[(n,f)][c*k + j] = the clamped-frame gather.  The GEMM, bias, ELU,
// dgrad and wgrad are the kernels of gemm.hip; this file holds the gather (im2col), its transpose (col2im, with the ELU
// derivative of the layer below fused) and the counter-based dropout of the fully connected part.  HBM-bound,
// element-wise; windows are short (F = 10 .. 50 frames), so no tiling of the frame axis is needed.
#include "ib_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int64_t ldcol, int N,
                                                      int F, int C, int k) {
  const int64_t n_el = (int64_t)N * F * ldcol;
  const int h = k / 2, K = C * k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / ldcol;
    const int q = (int)(i % ldcol);
    T v = ib_from_f32<T>(0.f);
    if (q < K) {
      const int c = q / k, j = q % k;
      const int n = (int)(row / F), f = (int)(row % F);
      const int fs = min(max(f + j - h, 0), F - 1);
      v = x[((int64_t)n * F + fs) * C + c];
    }
    col[i] = v;
  }
}

// bf16, row pitch a multiple of 8: one thread gathers the 8 values of a 16-byte piece (8 two-byte loads that hit L1/L2 --
// x is read 7 times over -- and ONE 16-byte store), all index arithmetic in 32 bits.  The element-per-thread form above
// with its 64-bit divisions ran at 1.2 TB/s (39 us for [12800, 1792]); it remains the fallback for fp32 / odd pitches.
__global__ __launch_bounds__(256) void im2col_vec8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int ldcol,
                                                           int M, int F, int C, int k) {
  const int ppr = ldcol >> 3, total = M * ppr;
  const int h = k / 2, K = C * k;
  for (int pi = blockIdx.x * blockDim.x + threadIdx.x; pi < total; pi += gridDim.x * blockDim.x) {
    const int row = pi / ppr, p = pi - row * ppr;
    const int n = row / F, f = row - n * F;
    int q = 8 * p, c = q / k, j = q - c * k;
    const bf16_t* xw = x + (int64_t)n * F * C;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t v = (bf16_t)0.f;
      if (q + e < K) v = xw[min(max(f + j - h, 0), F - 1) * C + c];
      o[e] = v;
      if (++j == k) { j = 0; ++c; }
    }
    __builtin_memcpy(__builtin_assume_aligned(col + (int64_t)row * ldcol + 8 * p, 16), &o, 16);
  }
}

// dx[(n,f')][c] = act'(aux) * sum over (f, j) with clamp(f + j - h) == f' of dcol[(n,f)][c*k + j], in a fixed order:
// the k exact hits j = 0..k-1 (f = f' - j + h), then -- on the first / last frame -- the clamped (padded) taps.
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, int64_t ldcol, const T* __restrict__ aux,
                                                      int act, T* __restrict__ dx, int N, int F, int C, int k) {
  const int n_el = N * F * C;                            // < 2^31 by the launch conditions: 32-bit divisions
  const int h = k / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += gridDim.x * blockDim.x) {
    const int row = i / C, c = i - row * C;
    const int n = row / F, fp = row - n * F;
    const T* base = dcol + (int64_t)n * F * ldcol + (int64_t)c * k;
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
      const int f = fp - j + h;
      if (f >= 0 && f < F) s += ib_to_f32(base[(int64_t)f * ldcol + j]);
    }
    if (fp == 0) {                       // taps that fell before the window were read from frame 0
      for (int f = 0; f < F && f < h; ++f)
        for (int j = 0; j + f < h; ++j) s += ib_to_f32(base[(int64_t)f * ldcol + j]);
    }
    if (fp == F - 1) {                   // taps beyond the window were read from the last frame
      for (int f = max(0, F - h); f < F; ++f)
        for (int j = k - 1; f + j - h > F - 1; --j) s += ib_to_f32(base[(int64_t)f * ldcol + j]);
    }
    if (aux) s *= ib_act_bwd(act, ib_to_f32(aux[i]));
    dx[i] = ib_from_f32<T>(s);
  }
}


template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p,
                                                       uint32_t seed, int step, const int32_t* __restrict__ step_dev) {
  const uint32_t st = (uint32_t)(step_dev ? *step_dev : step);
  const uint32_t key = ib_mix32(seed ^ ib_mix32(st + 0x9e3779b9u));
  const float keep = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);          // drop when the 32-bit draw is below p * 2^32
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = ib_mix32((uint32_t)i ^ key) ^ ib_mix32((uint32_t)(i >> 32) + key);
    y[i] = r < thr ? ib_from_f32<T>(0.f) : ib_from_f32<T>(ib_to_f32(x[i]) * keep);
  }
}

// bf16, n % 8 == 0, 16-byte aligned: 8 elements per thread through one 16-byte load and store (same mask as above)
__global__ __launch_bounds__(256) void dropout_vec8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8,
                                                            float p, uint32_t seed, int step,
                                                            const int32_t* __restrict__ step_dev) {
  const uint32_t st = (uint32_t)(step_dev ? *step_dev : step);
  const uint32_t key = ib_mix32(seed ^ ib_mix32(st + 0x9e3779b9u));
  const float keep = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += (int64_t)gridDim.x * blockDim.x) {
    bf16x8_t v;
    __builtin_memcpy(&v, __builtin_assume_aligned(x + 8 * g, 16), 16);
    const uint32_t hi = ib_mix32((uint32_t)((8 * g) >> 32) + key);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t r = ib_mix32((uint32_t)(8 * g + e) ^ key) ^ hi;       // (8g + e) >> 32 is the same for all 8
      v[e] = r < thr ? (bf16_t)0.f : (bf16_t)((float)v[e] * keep);
    }
    __builtin_memcpy(__builtin_assume_aligned(y + 8 * g, 16), &v, 16);
  }
}

}  // namespace

extern "C" int ib_im2col_replicate(const void* x, void* col, int64_t ldcol, int64_t N, int64_t F, int64_t C, int k, int dtype,
                                   ib_stream_t stream) {
  if (!x || !col || N <= 0 || F <= 0 || C <= 0 || k <= 0 || (k & 1) == 0 || ldcol < C * k) return IB_E_ARG;
  if (N * F * ldcol >= (int64_t)1 << 31) return IB_E_UNSUPPORTED;
  if (dtype == IB_BF16 && ldcol % 8 == 0 && (reinterpret_cast<uintptr_t>(col) % 16) == 0) {
    hipLaunchKernelGGL(im2col_vec8_kernel, dim3(ib_grid_1d(N * F * (ldcol / 8), 256, 256 * 16)), dim3(256), 0, ib_s(stream),
                       (const bf16_t*)x, (bf16_t*)col, (int)ldcol, (int)(N * F), (int)F, (int)C, k);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  const int grid = ib_grid_1d(N * F * ldcol, 256);
  if (dtype == IB_F32)
    hipLaunchKernelGGL((im2col_kernel<float>), dim3(grid), dim3(256), 0, ib_s(stream), (const float*)x, (float*)col, ldcol,
                       (int)N, (int)F, (int)C, k);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL((im2col_kernel<bf16_t>), dim3(grid), dim3(256), 0, ib_s(stream), (const bf16_t*)x, (bf16_t*)col, ldcol,
                       (int)N, (int)F, (int)C, k);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_col2im_replicate(const void* dcol, int64_t ldcol, const void* aux, int act, void* dx, int64_t N, int64_t F,
                                   int64_t C, int k, int dtype, ib_stream_t stream) {
  if (!dcol || !dx || N <= 0 || F <= 0 || C <= 0 || k <= 0 || (k & 1) == 0 || ldcol < C * k) return IB_E_ARG;
  if (act < IB_ACT_NONE || act > IB_ACT_ELU) return IB_E_ARG;
  if (act == IB_ACT_NONE) aux = nullptr;
  if (N * F * C >= (int64_t)1 << 31 || N * F * ldcol >= (int64_t)1 << 31) return IB_E_UNSUPPORTED;
  const int grid = ib_grid_1d(N * F * C, 256, 256 * 16);
  if (dtype == IB_F32)
    hipLaunchKernelGGL((col2im_kernel<float>), dim3(grid), dim3(256), 0, ib_s(stream), (const float*)dcol, ldcol,
                       (const float*)aux, act, (float*)dx, (int)N, (int)F, (int)C, k);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL((col2im_kernel<bf16_t>), dim3(grid), dim3(256), 0, ib_s(stream), (const bf16_t*)dcol, ldcol,
                       (const bf16_t*)aux, act, (bf16_t*)dx, (int)N, (int)F, (int)C, k);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_dropout(const void* x, void* y, int64_t n, float p, uint32_t seed, int32_t step, const int32_t* step_dev,
                          int dtype, ib_stream_t stream) {
  if (!x || !y || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return IB_E_ARG;
  if (dtype == IB_BF16 && n % 8 == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0) {
    hipLaunchKernelGGL(dropout_vec8_kernel, dim3(ib_grid_1d(n / 8, 256)), dim3(256), 0, ib_s(stream), (const bf16_t*)x,
                       (bf16_t*)y, n / 8, p, seed, step, step_dev);
    IB_CHECK_LAUNCH();
    return IB_OK;
  }
  const int grid = ib_grid_1d(n, 256);
  if (dtype == IB_F32)
    hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid), dim3(256), 0, ib_s(stream), (const float*)x, (float*)y, n, p, seed,
                       step, step_dev);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid), dim3(256), 0, ib_s(stream), (const bf16_t*)x, (bf16_t*)y, n, p,
                       seed, step, step_dev);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
