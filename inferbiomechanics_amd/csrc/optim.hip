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

__global__ __launch_bounds__(256) void optim_kernel(OptArgs a) {
  // self-counting mode (ticket != NULL): *step_dev holds the number of COMPLETED steps; every block reads it on
  // entry, and the block that draws the last exit ticket publishes step and resets the ticket -- it exits after
  // every other block has entered (and therefore read the old value), so no block can see the new count.
  const int step = a.step_dev ? (a.ticket ? *(volatile int32_t*)a.step_dev + 1 : *a.step_dev) : a.step;
  // bias corrections in double (torch computes them as Python floats)
  const float bc1 = (float)(1.0 - pow(0.9, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow(0.999, (double)step));
  const int64_t n4 = a.n / 4;
  const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    float4 g = reinterpret_cast<const float4*>(a.g)[i];
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
  for (int64_t i = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    float s1 = has1 ? a.s1[i] : 0.f, s2 = has2 ? a.s2[i] : 0.f;
    const float p = opt_update(a, a.p[i], a.g[i] * a.gscale, s1, s2, bc1, bc2s);
    a.p[i] = p;
    if (has1) a.s1[i] = s1;
    if (has2) a.s2[i] = s2;
    if (a.shadow) a.shadow[i] = (bf16_t)p;
  }
  if (a.ticket) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = atomicAdd(a.ticket, 1);
      if (t == (int)gridDim.x - 1) {
        *a.step_dev = step;
        *a.ticket = 0;
      }
    }
  }
}

}  // namespace

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
  hipLaunchKernelGGL(optim_kernel, dim3(ib_grid_1d(n / 4 + 1, 256)), dim3(256), 0, ib_s(stream), a);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
