// nn.BatchNorm1d over a [B, C] matrix (the optional layer the reference's feedforward model puts in front of every
// Linear: src/models/FeedForwardRegressionBaseline.py:71-72, CLI flag --batchnorm, src/cli/train.py:47) for gfx950.
//
// Per feature (column) statistics over the batch (rows).  A workgroup owns 64 consecutive columns (one wave-instruction
// reads 64 consecutive elements of a row: coalesced) and splits the rows over its 4 waves; the partial sums of the 4
// waves are combined through LDS in a FIXED order, so results are bitwise reproducible.  Two passes for the variance
// (sum, then sum of squared deviations): the matrix is a few hundred rows and L2-resident, a second read is cheaper than
// the cancellation of the one-pass form.  Everything accumulates in fp32.
//
//   training: mean / biased variance of the batch normalise; running_mean <- (1 - m) running_mean + m mean,
//             running_var <- (1 - m) running_var + m var * B / (B - 1)  (torch's unbiased update), num_batches_tracked += 1
//   eval:     running statistics normalise; nothing is updated
// Backward (training): dgamma = sum dy xhat, dbeta = sum dy, dx = gamma rstd / B (B dy - dbeta - xhat dgamma);
// (eval): dx = gamma rstd dy.  The derivative of the activation BELOW the layer (the previous block's act, whose output --
// possibly through dropout, an elementwise mask that commutes -- is this layer's input) can be multiplied in the same pass.
#include "ib_common.h"

namespace {

constexpr int BN_COLS = 64, BN_WAVES = 4;

// per-column total of v over the 4 waves, fixed order; every thread of the column gets it
__device__ __forceinline__ float bn_combine(float v, float (*red)[BN_COLS], int wave, int col) {
  __syncthreads();                         // earlier readers of `red` are done
  red[wave][col] = v;
  __syncthreads();
  return ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
}

template <typename T>
__global__ __launch_bounds__(BN_COLS* BN_WAVES) void batchnorm_fwd_kernel(
    const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
    T* __restrict__ y, int64_t ldy, float* __restrict__ save_mean, float* __restrict__ save_rstd, int B, int C,
    float momentum, float eps, int training) {
  __shared__ float red[BN_WAVES][BN_COLS];
  const int col = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * BN_COLS + col;
  const bool live = c < C;
  float mean, rstd;
  if (training) {
    float s = 0.f;
    if (live)
      for (int b = wave; b < B; b += BN_WAVES) s += ib_to_f32(x[(int64_t)b * ldx + c]);
    mean = bn_combine(s, red, wave, col) / (float)B;
    float q = 0.f;
    if (live)
      for (int b = wave; b < B; b += BN_WAVES) {
        const float d = ib_to_f32(x[(int64_t)b * ldx + c]) - mean;
        q += d * d;
      }
    const float ss = bn_combine(q, red, wave, col);
    const float var = ss / (float)B;
    rstd = 1.f / sqrtf(var + eps);
    if (live && wave == 0) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (ss / (float)(B - 1));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  } else {
    mean = live ? running_mean[c] : 0.f;
    rstd = live ? 1.f / sqrtf(running_var[c] + eps) : 0.f;
  }
  if (!live) return;
  if (wave == 0) { save_mean[c] = mean; save_rstd[c] = rstd; }
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  for (int b = wave; b < B; b += BN_WAVES)
    y[(int64_t)b * ldy + c] = ib_from_f32<T>((ib_to_f32(x[(int64_t)b * ldx + c]) - mean) * rstd * g + bt);
}

template <typename T>
__global__ __launch_bounds__(BN_COLS* BN_WAVES) void batchnorm_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_rstd, T* __restrict__ dx, int64_t lddx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int act_below, const T* __restrict__ aux,
    int64_t ldaux, int B, int C, int training) {
  __shared__ float red[BN_WAVES][BN_COLS];
  const int col = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * BN_COLS + col;
  const bool live = c < C;
  const float mean = live ? save_mean[c] : 0.f, rstd = live ? save_rstd[c] : 0.f;
  float sg = 0.f, sb = 0.f;
  if (live)
    for (int b = wave; b < B; b += BN_WAVES) {
      const float d = ib_to_f32(dy[(int64_t)b * lddy + c]);
      sb += d;
      sg += d * ((ib_to_f32(x[(int64_t)b * ldx + c]) - mean) * rstd);
    }
  const float tg = bn_combine(sg, red, wave, col);
  const float tb = bn_combine(sb, red, wave, col);
  if (!live) return;
  if (wave == 0) {
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + tg : tg;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + tb : tb;
  }
  if (!dx) return;
  const float g = gamma ? gamma[c] : 1.f;
  const float k = g * rstd, invB = 1.f / (float)B;
  for (int b = wave; b < B; b += BN_WAVES) {
    const float d = ib_to_f32(dy[(int64_t)b * lddy + c]);
    float v;
    if (training) {
      const float xh = (ib_to_f32(x[(int64_t)b * ldx + c]) - mean) * rstd;
      v = k * (d - invB * (tb + xh * tg));
    } else {
      v = k * d;
    }
    if (act_below != IB_ACT_NONE) v *= ib_act_bwd(act_below, ib_to_f32(aux[(int64_t)b * ldaux + c]));
    dx[(int64_t)b * lddx + c] = ib_from_f32<T>(v);
  }
}

}  // namespace

extern "C" int ib_batchnorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int64_t* num_batches_tracked, void* y, int64_t ldy, float* save_mean,
                                float* save_rstd, int64_t B, int64_t C, float momentum, float eps, int training, int dtype,
                                ib_stream_t stream) {
  if (!x || !y || !running_mean || !running_var || !save_mean || !save_rstd || B <= 0 || C <= 0 || ldx < C || ldy < C)
    return IB_E_ARG;
  if (training && B < 2) return IB_E_ARG;          // torch: "Expected more than 1 value per channel when training"
  const dim3 grid((unsigned)((C + BN_COLS - 1) / BN_COLS)), block(BN_COLS * BN_WAVES);
  hipStream_t s = ib_s(stream);
  long long* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  if (dtype == IB_F32)
    hipLaunchKernelGGL((batchnorm_fwd_kernel<float>), grid, block, 0, s, (const float*)x, ldx, gamma, beta, running_mean,
                       running_var, nbt, (float*)y, ldy, save_mean, save_rstd, (int)B, (int)C, momentum, eps, training);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL((batchnorm_fwd_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, ldx, gamma, beta, running_mean,
                       running_var, nbt, (bf16_t*)y, ldy, save_mean, save_rstd, (int)B, (int)C, momentum, eps, training);
  else
    return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_batchnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                                const float* save_mean, const float* save_rstd, void* dx, int64_t lddx, float* dgamma,
                                float* dbeta, int accumulate, int act_below, const void* aux, int64_t ldaux, int64_t B,
                                int64_t C, int training, int dtype, ib_stream_t stream) {
  if (!dy || !x || !save_mean || !save_rstd || B <= 0 || C <= 0 || lddy < C || ldx < C || (dx && lddx < C)) return IB_E_ARG;
  if (act_below != IB_ACT_NONE && (!aux || ldaux < C)) return IB_E_ARG;
  const dim3 grid((unsigned)((C + BN_COLS - 1) / BN_COLS)), block(BN_COLS * BN_WAVES);
  hipStream_t s = ib_s(stream);
  if (dtype == IB_F32)
    hipLaunchKernelGGL((batchnorm_bwd_kernel<float>), grid, block, 0, s, (const float*)dy, lddy, (const float*)x, ldx, gamma,
                       save_mean, save_rstd, (float*)dx, lddx, dgamma, dbeta, accumulate, act_below, (const float*)aux, ldaux,
                       (int)B, (int)C, training);
  else if (dtype == IB_BF16)
    hipLaunchKernelGGL((batchnorm_bwd_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx,
                       gamma, save_mean, save_rstd, (bf16_t*)dx, lddx, dgamma, dbeta, accumulate, act_below,
                       (const bf16_t*)aux, ldaux, (int)B, (int)C, training);
  else
    return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
