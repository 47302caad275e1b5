// Loss kernels for gfx950.
//  * ib_regression_loss: the whole of RegressionLossEvaluator.__call__ steps 1-2.2
//    (src/loss/RegressionLossEvaluator.py:184-263) -- 4 per-component MSE vectors, the CoP mask from the
//    LABEL force norm (> threshold, strict), the component-selected scalar loss, the six last-frame
//    norm metrics and d loss/d outputs -- in two launches (per-block partials, then one finalising
//    block that sums partials in a fixed order).  The reference issues ~40 small kernels and 7 .item()
//    host syncs for the same arithmetic.
//  * ib_mse_loss: the diffusion eps-prediction loss (build-defined).
#include "ib_common.h"

namespace {

constexpr int NPART = 40;  // 30 component sums + 8 metric sums (+2 pad)
// partial layout: [0..5] force, [6..11] cop, [12..17] moment, [18..29] wrench,
//                 [30] force-norm sum, [31] moment-norm sum, [32] cop-norm sum, [33] wrench6-norm sum,
//                 [34] wrench-moment-left norm sum, [35] wrench-moment-right norm sum, [36] com-acc norm sum

// 64-lane sum without the LDS crossbar: four DPP adds give every lane its 16-lane row total, the four row totals are
// read out as scalars and added in a fixed order.  (37 butterflies of ds_bpermute shuffles were most of this kernel.)
template <int CTRL>
__device__ __forceinline__ float rl_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rl_wave_sum(float v) {
  v += rl_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += rl_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += rl_dpp<0x141>(v);    // row_half_mirror
  v += rl_dpp<0x140>(v);    // row_mirror
  const int i = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
  return ((r0 + r1) + r2) + r3;
}

__device__ __forceinline__ void rl_finalize(const float* tot, const float* __restrict__ comp_w, float* __restrict__ result,
                                            int B, int F);

template <typename T>
__global__ __launch_bounds__(256) void regression_loss_partial_kernel(
    const T* __restrict__ o_cop, const T* __restrict__ o_force, const T* __restrict__ o_torque,
    const T* __restrict__ o_wrench, int64_t bs_cop, int64_t bs_force, int64_t bs_torque, int64_t bs_wrench,
    int fo_cop, int fo_force, int fo_torque, int fo_wrench,      // elements between frames of an output / gradient key
    const float* __restrict__ l_cop, const float* __restrict__ l_force,
    const float* __restrict__ l_torque, const float* __restrict__ l_wrench, const float* __restrict__ comp_w,
    float threshold, T* __restrict__ g_cop, T* __restrict__ g_force, T* __restrict__ g_torque, T* __restrict__ g_wrench,
    int64_t gs_cop, int64_t gs_force, int64_t gs_torque, int64_t gs_wrench, int fg_cop, int fg_force, int fg_torque,
    int fg_wrench, float* __restrict__ partial, int B, int F, float* __restrict__ result_inline) {
  __shared__ float red[4][NPART];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[NPART];
#pragma unroll
  for (int i = 0; i < NPART; ++i) acc[i] = 0.f;
  const int rows = B * F;
  const float gscale = 2.f / (float)rows;
  float cw[30];                                     // component weights: requested once, ahead of the rows
#pragma unroll
  for (int c = 0; c < 30; ++c) cw[c] = comp_w[c];
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
    const int b = row / F, f = row % F;
    const bool last = (f == F - 1);
    // every operand of the row is requested before the first is used (48 output / label values): written section by section
    // -- load, square, store the gradient, next key -- the kernel was four dependent memory round trips (10 us for 40 rows)
    float of[6], lf[6], oc[6], lc[6], ot[6], lt[6], ow[12], lw[12];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      of[c] = ib_to_f32(o_force[(int64_t)b * bs_force + f * fo_force + c]);
      lf[c] = l_force[(int64_t)row * 6 + c];
      oc[c] = ib_to_f32(o_cop[(int64_t)b * bs_cop + f * fo_cop + c]);
      lc[c] = l_cop[(int64_t)row * 6 + c];
      ot[c] = ib_to_f32(o_torque[(int64_t)b * bs_torque + f * fo_torque + c]);
      lt[c] = l_torque[(int64_t)row * 6 + c];
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      ow[c] = ib_to_f32(o_wrench[(int64_t)b * bs_wrench + f * fo_wrench + c]);
      lw[c] = l_wrench[(int64_t)row * 12 + c];
    }
    float d6[6];
    // ---- force (also feeds the CoP mask and the COM-acc metric)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      d6[c] = of[c] - lf[c];
      acc[c] += d6[c] * d6[c];
      if (g_force) g_force[(int64_t)b * gs_force + f * fg_force + c] = ib_from_f32<T>(cw[c] * gscale * d6[c]);
    }
    float mask[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float n = sqrtf(lf[3 * v] * lf[3 * v] + lf[3 * v + 1] * lf[3 * v + 1] + lf[3 * v + 2] * lf[3 * v + 2]);
      mask[v] = n > threshold ? 1.f : 0.f;  // strict '>': RegressionLossEvaluator.py:101
    }
    if (last) {
      acc[30] += sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]) + sqrtf(d6[3] * d6[3] + d6[4] * d6[4] + d6[5] * d6[5]);
      const float cx = (of[0] + of[3]) - (lf[0] + lf[3]);
      const float cy = (of[1] + of[4]) - (lf[1] + lf[4]);
      const float cz = (of[2] + of[5]) - (lf[2] + lf[5]);
      acc[36] += sqrtf(cx * cx + cy * cy + cz * cz);
    }
    // ---- CoP (masked on both sides: RegressionLossEvaluator.py:210-214)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float m = mask[c / 3];
      d6[c] = oc[c] * m - lc[c] * m;
      acc[6 + c] += d6[c] * d6[c];
      if (g_cop) g_cop[(int64_t)b * gs_cop + f * fg_cop + c] = ib_from_f32<T>(cw[6 + c] * gscale * d6[c] * m);
    }
    if (last)
      acc[32] += sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]) + sqrtf(d6[3] * d6[3] + d6[4] * d6[4] + d6[5] * d6[5]);
    // ---- moment
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      d6[c] = ot[c] - lt[c];
      acc[12 + c] += d6[c] * d6[c];
      if (g_torque) g_torque[(int64_t)b * gs_torque + f * fg_torque + c] = ib_from_f32<T>(cw[12 + c] * gscale * d6[c]);
    }
    if (last)
      acc[31] += sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]) + sqrtf(d6[3] * d6[3] + d6[4] * d6[4] + d6[5] * d6[5]);
    // ---- wrench (12 = [moment3, force3] x 2 bodies)
    float d12[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      d12[c] = ow[c] - lw[c];
      acc[18 + c] += d12[c] * d12[c];
      if (g_wrench) g_wrench[(int64_t)b * gs_wrench + f * fg_wrench + c] = ib_from_f32<T>(cw[18 + c] * gscale * d12[c]);
    }
    if (last) {
      float n0 = 0.f, n1 = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) { n0 += d12[c] * d12[c]; n1 += d12[6 + c] * d12[6 + c]; }
      acc[33] += sqrtf(n0) + sqrtf(n1);
      acc[34] += sqrtf(d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2]);
      acc[35] += sqrtf(d12[6] * d12[6] + d12[7] * d12[7] + d12[8] * d12[8]);
    }
  }
#pragma unroll
  for (int i = 0; i < NPART; ++i) {
    const float s = rl_wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NPART) {
    const int i = threadIdx.x;
    const float v = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    partial[(int64_t)blockIdx.x * NPART + i] = v;
    if (result_inline) red[0][i] = 0.f + v;            // what the final kernel's one-part sum would hold
  }
  // a single block (the reference's batch sizes: B x F <= 256 pairs) finishes the loss itself: the second launch was 5 us
  // of launch latency for 37 additions
  if (result_inline) {
    __syncthreads();
    if (threadIdx.x == 0) rl_finalize(red[0], comp_w, result_inline, B, F);
  }
}

__global__ void regression_loss_final_kernel(const float* __restrict__ partial, int nparts,
                                             const float* __restrict__ comp_w, float* __restrict__ result, int B, int F) {
  __shared__ float tot[NPART];
  if (threadIdx.x < NPART) {
    float s = 0.f;
    for (int b0 = 0; b0 < nparts; b0 += 8) {           // up to 8 partial rows requested together, added in block order
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (b0 + e < nparts) v[e] = partial[(int64_t)(b0 + e) * NPART + threadIdx.x];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (b0 + e < nparts) s += v[e];
    }
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) rl_finalize(tot, comp_w, result, B, F);
}

// the loss value, the 30 per-component means and the metric means from the NPART column totals (one thread)
__device__ __forceinline__ void rl_finalize(const float* tot, const float* __restrict__ comp_w, float* __restrict__ result,
                                            int B, int F) {
  {
    const float inv = 1.f / (float)(B * F);
    float loss = 0.f;
    for (int c = 0; c < 30; ++c) {
      const float v = tot[c] * inv;
      result[1 + c] = v;
      loss += comp_w[c] * v;
    }
    result[0] = loss;
    const float invb2 = 1.f / (float)(2 * B), invb = 1.f / (float)B;
    result[31] = tot[30] * invb2;                              // force
    result[32] = tot[31] * invb2;                              // moment
    result[33] = tot[32] * invb2;                              // cop
    result[34] = tot[33] * invb2;                              // wrench (vec 6)
    result[35] = (tot[34] * invb + tot[35] * invb) * 0.5f;     // wrench moment
    result[36] = tot[36] * invb;                               // com acc
  }
}

// pred / dpred are [rows, cols] with leading dimensions (the trainer pads D = 300 rows to ld = 304); target is
// contiguous.  4 consecutive columns per thread.
template <typename T, int V>
__global__ __launch_bounds__(256) void mse_partial_kernel(const T* __restrict__ pred, int64_t ld_pred,
                                                          const T* __restrict__ target, T* __restrict__ dpred,
                                                          int64_t ld_dpred, float* __restrict__ partial, int64_t rows,
                                                          int64_t cols, float gscale) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t cv = cols / V;
  const int64_t n = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv, c = (i % cv) * V;
    float a[V], b[V], d[V];
    if constexpr (V == 4) {
      if constexpr (sizeof(T) == 2) {
        const bf16x4_t ta = *reinterpret_cast<const bf16x4_t*>(pred + r * ld_pred + c), tb = *reinterpret_cast<const bf16x4_t*>(target + r * cols + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (float)ta[e]; b[e] = (float)tb[e]; }
      } else {
        const float4 ta = *reinterpret_cast<const float4*>(pred + r * ld_pred + c), tb = *reinterpret_cast<const float4*>(target + r * cols + c);
        a[0] = ta.x; a[1] = ta.y; a[2] = ta.z; a[3] = ta.w; b[0] = tb.x; b[1] = tb.y; b[2] = tb.z; b[3] = tb.w;
      }
    } else {
      a[0] = ib_to_f32(pred[r * ld_pred + c]); b[0] = ib_to_f32(target[r * cols + c]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { d[e] = a[e] - b[e]; s += d[e] * d[e]; d[e] *= gscale; }
    if (dpred) {
      if constexpr (V == 4) {
        if constexpr (sizeof(T) == 2) {
          bf16x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)d[e];
          *reinterpret_cast<bf16x4_t*>(dpred + r * ld_dpred + c) = o;
        } else {
          *reinterpret_cast<float4*>(dpred + r * ld_dpred + c) = make_float4(d[0], d[1], d[2], d[3]);
        }
      } else {
        dpred[r * ld_dpred + c] = ib_from_f32<T>(d[0]);
      }
    }
  }
  s = ib_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void mse_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ result, float inv_n) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) result[0] = red[0] * inv_n;
}

int rl_parts(int64_t rows) { return ib_grid_1d(rows, 256, 256); }
int mse_parts(int64_t n) { return ib_grid_1d(n, 256 * 8, 1024); }

}  // namespace

extern "C" size_t ib_regression_loss_workspace(int64_t B, int64_t F) {
  return (size_t)rl_parts(B * F) * NPART * sizeof(float);
}

extern "C" int ib_regression_loss_strided(const void* o_cop, const void* o_force, const void* o_torque, const void* o_wrench,
                                          const int64_t* o_bs, const int64_t* o_fs, const float* l_cop,
                                          const float* l_force, const float* l_torque, const float* l_wrench,
                                          const float* comp_w, float threshold, float* result, void* g_cop, void* g_force,
                                          void* g_torque, void* g_wrench, const int64_t* g_bs, const int64_t* g_fs,
                                          void* workspace, size_t workspace_bytes, int64_t B, int64_t F, int dtype,
                                          ib_stream_t stream) {
  if (!o_cop || !o_force || !o_torque || !o_wrench || !l_cop || !l_force || !l_torque || !l_wrench || !comp_w || !result)
    return IB_E_ARG;
  if (B <= 0 || F <= 0 || !o_bs) return IB_E_ARG;
  const bool has_g = g_cop || g_force || g_torque || g_wrench;
  if (has_g && (!g_cop || !g_force || !g_torque || !g_wrench || !g_bs)) return IB_E_ARG;
  const int64_t zero4[4] = {0, 0, 0, 0}, dense[4] = {6, 6, 6, 12};
  const int64_t* gb = has_g ? g_bs : zero4;
  const int64_t* of = o_fs ? o_fs : dense;
  const int64_t* gf = g_fs ? g_fs : dense;
  for (int k = 0; k < 4; ++k)
    if (of[k] < dense[k] || gf[k] < dense[k]) return IB_E_ARG;
  const int parts = rl_parts(B * F);
  if (!workspace || workspace_bytes < (size_t)parts * NPART * sizeof(float)) return IB_E_WORKSPACE;
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t s = ib_s(stream);
  float* inline_result = parts == 1 ? result : nullptr;       // one block: it writes the result itself, no second launch
  if (dtype == IB_F32) {
    hipLaunchKernelGGL((regression_loss_partial_kernel<float>), dim3(parts), dim3(256), 0, s, (const float*)o_cop,
                       (const float*)o_force, (const float*)o_torque, (const float*)o_wrench, o_bs[0], o_bs[1], o_bs[2],
                       o_bs[3], (int)of[0], (int)of[1], (int)of[2], (int)of[3], l_cop, l_force, l_torque, l_wrench, comp_w,
                       threshold, (float*)g_cop, (float*)g_force, (float*)g_torque, (float*)g_wrench, gb[0], gb[1], gb[2],
                       gb[3], (int)gf[0], (int)gf[1], (int)gf[2], (int)gf[3], partial, (int)B, (int)F, inline_result);
  } else if (dtype == IB_BF16) {
    hipLaunchKernelGGL((regression_loss_partial_kernel<bf16_t>), dim3(parts), dim3(256), 0, s, (const bf16_t*)o_cop,
                       (const bf16_t*)o_force, (const bf16_t*)o_torque, (const bf16_t*)o_wrench, o_bs[0], o_bs[1],
                       o_bs[2], o_bs[3], (int)of[0], (int)of[1], (int)of[2], (int)of[3], l_cop, l_force, l_torque,
                       l_wrench, comp_w, threshold, (bf16_t*)g_cop, (bf16_t*)g_force, (bf16_t*)g_torque,
                       (bf16_t*)g_wrench, gb[0], gb[1], gb[2], gb[3], (int)gf[0], (int)gf[1], (int)gf[2], (int)gf[3],
                       partial, (int)B, (int)F, inline_result);
  } else {
    return IB_E_DTYPE;
  }
  IB_CHECK_LAUNCH();
  if (!inline_result) {
    hipLaunchKernelGGL(regression_loss_final_kernel, dim3(1), dim3(64), 0, s, partial, parts, comp_w, result, (int)B, (int)F);
    IB_CHECK_LAUNCH();
  }
  return IB_OK;
}

extern "C" int ib_regression_loss(const void* o_cop, const void* o_force, const void* o_torque, const void* o_wrench,
                                  const int64_t* o_bs, const float* l_cop, const float* l_force, const float* l_torque,
                                  const float* l_wrench, const float* comp_w, float threshold, float* result, void* g_cop,
                                  void* g_force, void* g_torque, void* g_wrench, const int64_t* g_bs, void* workspace,
                                  size_t workspace_bytes, int64_t B, int64_t F, int dtype, ib_stream_t stream) {
  return ib_regression_loss_strided(o_cop, o_force, o_torque, o_wrench, o_bs, nullptr, l_cop, l_force, l_torque, l_wrench,
                                    comp_w, threshold, result, g_cop, g_force, g_torque, g_wrench, g_bs, nullptr, workspace,
                                    workspace_bytes, B, F, dtype, stream);
}

extern "C" size_t ib_mse_loss_workspace(int64_t n) { return (size_t)mse_parts(n) * sizeof(float); }

extern "C" int ib_mse_loss_partial(const void* pred, int64_t ld_pred, const void* target, void* dpred, int64_t ld_dpred,
                                   void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols, int dtype,
                                   ib_stream_t stream) {
  if (!pred || !target || rows <= 0 || cols <= 0 || ld_pred < cols || (dpred && ld_dpred < cols)) return IB_E_ARG;
  const int64_t n = rows * cols;
  const int parts = mse_parts(n);
  if (!workspace || workspace_bytes < (size_t)parts * sizeof(float)) return IB_E_WORKSPACE;
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t s = ib_s(stream);
  const float gscale = 2.f / (float)n;
  const int es = dtype == IB_BF16 ? 2 : 4;
  auto al = [&](const void* q) { return !q || (reinterpret_cast<uintptr_t>(q) % (4 * es)) == 0; };
  const bool v4 = (cols % 4 == 0) && (ld_pred % 4 == 0) && (!dpred || ld_dpred % 4 == 0) && al(pred) && al(target) && al(dpred);
  if (dtype == IB_F32) {
    if (v4) hipLaunchKernelGGL((mse_partial_kernel<float, 4>), dim3(parts), dim3(256), 0, s, (const float*)pred, ld_pred, (const float*)target, (float*)dpred, ld_dpred, partial, rows, cols, gscale);
    else hipLaunchKernelGGL((mse_partial_kernel<float, 1>), dim3(parts), dim3(256), 0, s, (const float*)pred, ld_pred, (const float*)target, (float*)dpred, ld_dpred, partial, rows, cols, gscale);
  } else if (dtype == IB_BF16) {
    if (v4) hipLaunchKernelGGL((mse_partial_kernel<bf16_t, 4>), dim3(parts), dim3(256), 0, s, (const bf16_t*)pred, ld_pred, (const bf16_t*)target, (bf16_t*)dpred, ld_dpred, partial, rows, cols, gscale);
    else hipLaunchKernelGGL((mse_partial_kernel<bf16_t, 1>), dim3(parts), dim3(256), 0, s, (const bf16_t*)pred, ld_pred, (const bf16_t*)target, (bf16_t*)dpred, ld_dpred, partial, rows, cols, gscale);
  } else
    return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_mse_loss_finalize(const void* workspace, size_t workspace_bytes, float* result, int64_t n,
                                    ib_stream_t stream) {
  if (!workspace || !result || n <= 0) return IB_E_ARG;
  const int parts = mse_parts(n);
  if (workspace_bytes < (size_t)parts * sizeof(float)) return IB_E_WORKSPACE;
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, ib_s(stream), reinterpret_cast<const float*>(workspace),
                     parts, result, 1.f / (float)n);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_mse_loss(const void* pred, const void* target, void* dpred, float* result, void* workspace,
                           size_t workspace_bytes, int64_t n, int dtype, ib_stream_t stream) {
  if (!result) return IB_E_ARG;
  const int rc = ib_mse_loss_partial(pred, n, target, dpred, n, workspace, workspace_bytes, 1, n, dtype, stream);
  if (rc != IB_OK) return rc;
  return ib_mse_loss_finalize(workspace, workspace_bytes, result, n, stream);
}

// ---- the four static helpers of the reference evaluator as entry points of their own (API surface:
// src/loss/RegressionLossEvaluator.py:73-158; the fused kernel above is what the training step uses).  Tiny problems
// ([B, F, 6 | 12]): one workgroup per output column / one thread per 3-vector, fixed summation order.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void sqdiff_mean_kernel(const T* __restrict__ o, const T* __restrict__ l,
                                                          float* __restrict__ out, int rows, int C) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float d = ib_to_f32(o[(int64_t)r * C + c]) - ib_to_f32(l[(int64_t)r * C + c]);
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = red[0] / (float)rows;
}
// d out[c] / d o[r, c] = 2 (o - l) / rows
template <typename T>
__global__ __launch_bounds__(256) void sqdiff_mean_bwd_kernel(const T* __restrict__ o, const T* __restrict__ l,
                                                              const float* __restrict__ dout, T* __restrict__ d_o,
                                                              int64_t n, int C, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d_o[i] = ib_from_f32<T>(dout[i % C] * scale * (ib_to_f32(o[i]) - ib_to_f32(l[i])));
}
template <typename T>
__global__ __launch_bounds__(256) void mask_by_threes_kernel(const T* __restrict__ t, float* __restrict__ mask, int64_t nvec,
                                                             float threshold) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = ib_to_f32(t[3 * i]), b = ib_to_f32(t[3 * i + 1]), c = ib_to_f32(t[3 * i + 2]);
    const float m = sqrtf(a * a + b * b + c * c) > threshold ? 1.f : 0.f;
    mask[3 * i] = m; mask[3 * i + 1] = m; mask[3 * i + 2] = m;
  }
}
// mean over (window, chunk) of || (o - l)[window, LAST frame, chunk of vec] ||; fold: the two halves of the last dimension
// are added first (left + right force, get_com_acc_error :143-158)
template <typename T>
__global__ __launch_bounds__(256) void mean_norm_error_kernel(const T* __restrict__ o, const T* __restrict__ l,
                                                              float* __restrict__ out, int B, int F, int C, int vec, int fold) {
  __shared__ float red[256];
  const int Ce = fold ? C / 2 : C, chunks = Ce / vec;
  float s = 0.f;
  for (int i = threadIdx.x; i < B * chunks; i += 256) {
    const int b = i / chunks, k = i % chunks;
    const int64_t base = ((int64_t)b * F + (F - 1)) * C + k * vec;
    float q = 0.f;
    for (int e = 0; e < vec; ++e) {
      float d = ib_to_f32(o[base + e]) - ib_to_f32(l[base + e]);
      if (fold) d = (ib_to_f32(o[base + e]) + ib_to_f32(o[base + e + Ce])) - (ib_to_f32(l[base + e]) + ib_to_f32(l[base + e + Ce]));
      q += d * d;
    }
    s += sqrtf(q);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)(B * chunks);
}
}  // namespace

extern "C" int ib_sqdiff_mean(const void* o, const void* l, float* out, int64_t rows, int64_t C, int dtype,
                              ib_stream_t stream) {
  if (!o || !l || !out || rows <= 0 || C <= 0 || rows >= (1 << 30) || C > 65535) return IB_E_ARG;
  if (dtype == IB_F32)
    hipLaunchKernelGGL(sqdiff_mean_kernel<float>, dim3((int)C), dim3(256), 0, ib_s(stream), (const float*)o, (const float*)l, out, (int)rows, (int)C);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL(sqdiff_mean_kernel<bf16_t>, dim3((int)C), dim3(256), 0, ib_s(stream), (const bf16_t*)o, (const bf16_t*)l, out, (int)rows, (int)C);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
extern "C" int ib_sqdiff_mean_bwd(const void* o, const void* l, const float* dout, void* d_o, int64_t rows, int64_t C,
                                  int dtype, ib_stream_t stream) {
  if (!o || !l || !dout || !d_o || rows <= 0 || C <= 0) return IB_E_ARG;
  const int64_t n = rows * C;
  const float scale = 2.f / (float)rows;
  if (dtype == IB_F32)
    hipLaunchKernelGGL(sqdiff_mean_bwd_kernel<float>, dim3(ib_grid_1d(n, 256)), dim3(256), 0, ib_s(stream), (const float*)o, (const float*)l, dout, (float*)d_o, n, (int)C, scale);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL(sqdiff_mean_bwd_kernel<bf16_t>, dim3(ib_grid_1d(n, 256)), dim3(256), 0, ib_s(stream), (const bf16_t*)o, (const bf16_t*)l, dout, (bf16_t*)d_o, n, (int)C, scale);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
extern "C" int ib_mask_by_threes(const void* t, float* mask, int64_t n, float threshold, int dtype, ib_stream_t stream) {
  if (!t || !mask || n <= 0 || n % 3) return IB_E_ARG;
  if (dtype == IB_F32)
    hipLaunchKernelGGL(mask_by_threes_kernel<float>, dim3(ib_grid_1d(n / 3, 256)), dim3(256), 0, ib_s(stream), (const float*)t, mask, n / 3, threshold);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL(mask_by_threes_kernel<bf16_t>, dim3(ib_grid_1d(n / 3, 256)), dim3(256), 0, ib_s(stream), (const bf16_t*)t, mask, n / 3, threshold);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
extern "C" int ib_mean_norm_error(const void* o, const void* l, float* out, int64_t B, int64_t F, int64_t C, int vec_size,
                                  int fold_halves, int dtype, ib_stream_t stream) {
  if (!o || !l || !out || B <= 0 || F <= 0 || C <= 0 || vec_size <= 0 || B * C >= (1 << 30)) return IB_E_ARG;
  if (fold_halves ? (C % 2 || (C / 2) % vec_size) : (C % vec_size)) return IB_E_ARG;
  if (dtype == IB_F32)
    hipLaunchKernelGGL(mean_norm_error_kernel<float>, dim3(1), dim3(256), 0, ib_s(stream), (const float*)o, (const float*)l, out, (int)B, (int)F, (int)C, vec_size, fold_halves);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL(mean_norm_error_kernel<bf16_t>, dim3(1), dim3(256), 0, ib_s(stream), (const bf16_t*)o, (const bf16_t*)l, out, (int)B, (int)F, (int)C, vec_size, fold_halves);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
