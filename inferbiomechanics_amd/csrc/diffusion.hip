// [BUILD-DEFINED] diffusion wrapper kernels (no reference counterpart):
// DDPM q_sample, DDIM (eta = 0) update, table gathers.  The schedule / coefficient / timestep-embedding
// tables are computed in float64 on the host and cast ONCE to fp32 (bit-exactness target of §8c); these
// kernels only index them, so noise-schedule and timestep indexing stay bit-exact.
#include "ib_common.h"

namespace {

template <typename T>
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, T* __restrict__ out,
                                   int64_t B, int64_t dim, int64_t table_rows) {
  const int64_t n = B * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / dim, d = i % dim;
    int64_t r = idx[b];
    r = r < 0 ? 0 : (r >= table_rows ? table_rows - 1 : r);
    out[i] = ib_from_f32<T>(table[r * dim + d]);
  }
}

// 8 elements per thread (16-byte bf16 / 2 x 16-byte fp32 accesses) when `per` is a multiple of 8
template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&v)[V]) {
  if constexpr (V == 1) { v[0] = ib_to_f32(p[0]); }
  else if constexpr (sizeof(T) == 2) {
    bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  } else {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
  if constexpr (V == 1) { p[0] = ib_from_f32<T>(v[0]); }
  else if constexpr (sizeof(T) == 2) {
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8_t*>(p) = o;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// rows = B*T tokens, cols = D features; 4 consecutive columns per thread (8 B bf16 / 16 B fp32).  x_t may have a
// padded leading dimension (the trainer keeps D = 300 activations at ld = 304 so every row starts 16-byte aligned).
template <typename T, int V>
__global__ void q_sample_kernel(const T* __restrict__ x0, const T* __restrict__ eps, const int64_t* __restrict__ t,
                                const float* __restrict__ sqrt_ab, const float* __restrict__ sqrt_1mab, T* __restrict__ xt,
                                int64_t ld_xt, int64_t rows, int64_t rows_per_window, int64_t cols, int64_t table_rows) {
  const int64_t cv = cols / V;
  const int64_t n = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv, c = (i % cv) * V;
    int64_t k = t[r / rows_per_window];
    k = k < 0 ? 0 : (k >= table_rows ? table_rows - 1 : k);
    const float a = sqrt_ab[k], s = sqrt_1mab[k];
    if constexpr (V == 4) {
      float x[4], e[4];
      if constexpr (sizeof(T) == 2) {
        bf16x4_t tx = *reinterpret_cast<const bf16x4_t*>(x0 + r * cols + c), te = *reinterpret_cast<const bf16x4_t*>(eps + r * cols + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = (float)tx[j]; e[j] = (float)te[j]; }
        bf16x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(a * x[j] + s * e[j]);
        *reinterpret_cast<bf16x4_t*>(xt + r * ld_xt + c) = o;
      } else {
        const float4 tx = *reinterpret_cast<const float4*>(x0 + r * cols + c), te = *reinterpret_cast<const float4*>(eps + r * cols + c);
        *reinterpret_cast<float4*>(xt + r * ld_xt + c) =
            make_float4(a * tx.x + s * te.x, a * tx.y + s * te.y, a * tx.z + s * te.z, a * tx.w + s * te.w);
      }
    } else {
      xt[r * ld_xt + c] = ib_from_f32<T>(a * ib_to_f32(x0[r * cols + c]) + s * ib_to_f32(eps[r * cols + c]));
    }
  }
}

template <typename T, int V>
__global__ void ddim_step_kernel(T* __restrict__ x, const T* __restrict__ eps, const float* __restrict__ coef,
                                 const int64_t* __restrict__ timesteps, int64_t num_steps, int step,
                                 const int32_t* __restrict__ step_dev, int64_t* __restrict__ t_out, int64_t B, int64_t n) {
  int s = step_dev ? *step_dev : step;
  s = s < 0 ? 0 : (s >= num_steps ? (int)num_steps - 1 : s);
  const float cx = coef[2 * s], ce = coef[2 * s + 1];
  const int64_t nv = n / V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float a[V], e[V], o[V];
    ldv<T, V>(x + i * V, a);
    ldv<T, V>(eps + i * V, e);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = cx * a[k] + ce * e[k];
    stv<T, V>(x + i * V, o);
  }
  if (t_out && blockIdx.x == 0) {
    const int64_t tn = (s + 1 < num_steps) ? timesteps[s + 1] : 0;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) t_out[b] = tn;
  }
}

__global__ void counter_add_kernel(int32_t* c, int32_t d) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += d;
}
__global__ void fill_i64_kernel(int64_t* dst, int64_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

}  // namespace

extern "C" int ib_gather_rows(const float* table, const int64_t* idx, void* out, int64_t B, int64_t dim,
                              int64_t table_rows, int dtype_out, ib_stream_t stream) {
  if (!table || !idx || !out || B <= 0 || dim <= 0 || table_rows <= 0) return IB_E_ARG;
  const int grid = ib_grid_1d(B * dim, 256);
  if (dtype_out == IB_F32)
    hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(grid), dim3(256), 0, ib_s(stream), table, idx, (float*)out, B, dim, table_rows);
  else if (dtype_out == IB_BF16)
    hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, ib_s(stream), table, idx, (bf16_t*)out, B, dim, table_rows);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_q_sample(const void* x0, const void* eps, const int64_t* t, const float* sqrt_ab, const float* sqrt_1mab,
                           void* x_t, int64_t ld_xt, int64_t B, int64_t T, int64_t D, int64_t table_rows, int dtype,
                           ib_stream_t stream) {
  if (!x0 || !eps || !t || !sqrt_ab || !sqrt_1mab || !x_t || B <= 0 || T <= 0 || D <= 0 || table_rows <= 0 || ld_xt < D)
    return IB_E_ARG;
  const int es = dtype == IB_BF16 ? 2 : 4;
  auto al = [&](const void* q) { return (reinterpret_cast<uintptr_t>(q) % (4 * es)) == 0; };
  const bool v4 = (D % 4 == 0) && (ld_xt % 4 == 0) && al(x0) && al(eps) && al(x_t);
  const int64_t rows = B * T;
  const int grid = ib_grid_1d(rows * D / (v4 ? 4 : 1), 256);
  hipStream_t s = ib_s(stream);
  if (dtype == IB_F32) {
    if (v4) hipLaunchKernelGGL((q_sample_kernel<float, 4>), dim3(grid), dim3(256), 0, s, (const float*)x0, (const float*)eps, t, sqrt_ab, sqrt_1mab, (float*)x_t, ld_xt, rows, T, D, table_rows);
    else hipLaunchKernelGGL((q_sample_kernel<float, 1>), dim3(grid), dim3(256), 0, s, (const float*)x0, (const float*)eps, t, sqrt_ab, sqrt_1mab, (float*)x_t, ld_xt, rows, T, D, table_rows);
  } else if (dtype == IB_BF16) {
    if (v4) hipLaunchKernelGGL((q_sample_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x0, (const bf16_t*)eps, t, sqrt_ab, sqrt_1mab, (bf16_t*)x_t, ld_xt, rows, T, D, table_rows);
    else hipLaunchKernelGGL((q_sample_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x0, (const bf16_t*)eps, t, sqrt_ab, sqrt_1mab, (bf16_t*)x_t, ld_xt, rows, T, D, table_rows);
  } else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_ddim_step(void* x, const void* eps, const float* coef, const int64_t* timesteps, int64_t num_steps,
                            int32_t step, const int32_t* step_dev, int64_t* t_out, int64_t B, int64_t n, int dtype,
                            ib_stream_t stream) {
  if (!x || !eps || !coef || num_steps <= 0 || n <= 0) return IB_E_ARG;
  if (t_out && (!timesteps || B <= 0)) return IB_E_ARG;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) % 16) == 0; };
  const bool v8 = (n % 8 == 0) && al16(x) && al16(eps);
  const int grid = ib_grid_1d(n / (v8 ? 8 : 1), 256);
  hipStream_t s = ib_s(stream);
  if (dtype == IB_F32) {
    if (v8) hipLaunchKernelGGL((ddim_step_kernel<float, 8>), dim3(grid), dim3(256), 0, s, (float*)x, (const float*)eps, coef, timesteps, num_steps, step, step_dev, t_out, B, n);
    else hipLaunchKernelGGL((ddim_step_kernel<float, 1>), dim3(grid), dim3(256), 0, s, (float*)x, (const float*)eps, coef, timesteps, num_steps, step, step_dev, t_out, B, n);
  } else if (dtype == IB_BF16) {
    if (v8) hipLaunchKernelGGL((ddim_step_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)eps, coef, timesteps, num_steps, step, step_dev, t_out, B, n);
    else hipLaunchKernelGGL((ddim_step_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)eps, coef, timesteps, num_steps, step, step_dev, t_out, B, n);
  } else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_counter_add(int32_t* counter, int32_t delta, ib_stream_t stream) {
  if (!counter) return IB_E_ARG;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ib_s(stream), counter, delta);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_fill_i64(int64_t* dst, int64_t value, int64_t n, ib_stream_t stream) {
  if (!dst || n <= 0) return IB_E_ARG;
  hipLaunchKernelGGL(fill_i64_kernel, dim3(ib_grid_1d(n, 256)), dim3(256), 0, ib_s(stream), dst, value, n);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// ---- gradient of a row gather (nn.Embedding backward, TransformerBaseline.py:41-48): dtable[r, :] = sum over the
// positions i with idx[i] == r of dout[i, :], added in position order (deterministic; tables of a few dozen rows)
namespace {
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                              float* __restrict__ dtable, int64_t n, int64_t dim,
                                                              int64_t table_rows) {
  const int64_t total = table_rows * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / dim, d = e % dim;
    float s = 0.f;
    for (int64_t i = 0; i < n; ++i) {
      int64_t k = idx[i];
      k = k < 0 ? 0 : (k >= table_rows ? table_rows - 1 : k);
      if (k == r) s += dout[i * dim + d];
    }
    dtable[e] = s;
  }
}
}  // namespace

extern "C" int ib_gather_rows_bwd(const float* dout, const int64_t* idx, float* dtable, int64_t n, int64_t dim,
                                  int64_t table_rows, ib_stream_t stream) {
  if (!dout || !idx || !dtable || n <= 0 || dim <= 0 || table_rows <= 0) return IB_E_ARG;
  if (n * table_rows * dim > ((int64_t)1 << 32)) return IB_E_UNSUPPORTED;      // meant for per-frame tables
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(ib_grid_1d(table_rows * dim, 256)), dim3(256), 0, ib_s(stream), dout, idx,
                     dtable, n, dim, table_rows);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// ---- on-device window cache (replaces AddBiomechanicsDataset.__getitem__ `:161-285` + the
// DataLoader collate + the model's torch.concat `FeedForwardRegressionBaseline.py:97-108` for cached windows):
// one packed fp32 row per window = [model input, frame-major F x 147 | labels, key-major: cop F' x 6, force F' x 6,
// torque F' x 6, wrench F' x 12], every block zero-padded to 4 values so each starts on a 16-byte boundary of the row
// ('last_frame' labels: 6 | 6 | 6 | 12 values in 8 | 8 | 8 | 12 slots).  One launch gathers a batch of rows into the model's input tensor (compute dtype) and the
// four contiguous label tensors the loss kernel takes.
namespace {
struct GatherWin {
  const float* table; int64_t row_elems, rows; const int64_t* idx; int64_t B;
  void* x_out; int64_t x_elems, x_pad; int x_bf16;
  float* lab[4]; int64_t lab_elems[4], lab_pad[4];
};
__global__ __launch_bounds__(256) void gather_windows_kernel(GatherWin p) {
  const int64_t per = p.row_elems / 4;                     // float4 pieces per window row (row_elems % 4 == 0)
  const int64_t n = p.B * per;
  const int64_t xq = p.x_pad / 4;
  const bool xvec = (p.x_elems & 3) == 0;               // 1470 inputs per window at the reference defaults: not a multiple of 4
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per, q = i % per;
    int64_t r = p.idx[b];
    r = r < 0 ? 0 : (r >= p.rows ? p.rows - 1 : r);
    const float4 v = *reinterpret_cast<const float4*>(p.table + r * p.row_elems + 4 * q);
    if (q < xq) {
      if (!xvec) {                       // rows of the model input are then only element-aligned: scalar stores
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t c = 4 * q + k;
          if (c < p.x_elems) {
            if (p.x_bf16) reinterpret_cast<bf16_t*>(p.x_out)[b * p.x_elems + c] = (bf16_t)e[k];
            else reinterpret_cast<float*>(p.x_out)[b * p.x_elems + c] = e[k];
          }
        }
      } else if (p.x_bf16) {
        bf16x4_t o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<bf16_t*>(p.x_out) + b * p.x_elems + 4 * q) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.x_out) + b * p.x_elems + 4 * q) = v;
      }
    } else {
      int64_t d = 4 * (q - xq);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (d < p.lab_pad[k]) {
          float* dst = p.lab[k] + b * p.lab_elems[k] + d;
          if (p.lab_elems[k] == p.lab_pad[k]) {
            *reinterpret_cast<float4*>(dst) = v;
          } else {                       // a label block that is not a whole number of 16-byte pieces: element stores
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (d + c < p.lab_elems[k]) dst[c] = e[c];
          }
          break;
        }
        d -= p.lab_pad[k];
      }
    }
  }
}
}  // namespace

extern "C" int ib_gather_windows(const float* table, int64_t row_elems, int64_t rows, const int64_t* idx, int64_t B,
                                 void* x_out, int64_t x_elems, int dtype_x, float* const* lab_out,
                                 const int64_t* lab_elems, ib_stream_t stream) {
  if (!table || !idx || !x_out || !lab_out || !lab_elems || rows <= 0 || B <= 0 || x_elems <= 0) return IB_E_ARG;
  if (dtype_x != IB_F32 && dtype_x != IB_BF16) return IB_E_DTYPE;
  const int64_t x_pad = (x_elems + 3) / 4 * 4;             // the input segment of a packed row is padded to 16 bytes
  int64_t total = x_pad;
  GatherWin p{};
  for (int k = 0; k < 4; ++k) {
    if (!lab_out[k] || lab_elems[k] <= 0 || (reinterpret_cast<uintptr_t>(lab_out[k]) % 16)) return IB_E_ARG;
    p.lab[k] = lab_out[k]; p.lab_elems[k] = lab_elems[k]; p.lab_pad[k] = (lab_elems[k] + 3) / 4 * 4;
    total += p.lab_pad[k];
  }
  if (total != row_elems || (reinterpret_cast<uintptr_t>(table) % 16) ||
      (reinterpret_cast<uintptr_t>(x_out) % (dtype_x == IB_BF16 ? 8 : 16)))
    return IB_E_ARG;
  p.table = table; p.row_elems = row_elems; p.rows = rows; p.idx = idx; p.B = B;
  p.x_out = x_out; p.x_elems = x_elems; p.x_pad = x_pad; p.x_bf16 = dtype_x == IB_BF16;
  hipLaunchKernelGGL(gather_windows_kernel, dim3(ib_grid_1d(B * (row_elems / 4), 256)), dim3(256), 0, ib_s(stream), p);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
