// [BUILD-DEFINED] the diffusion batch made ON the device (no reference counterpart: the reference has no diffusion
// path).  One launch per training step gathers x0 windows out of an HBM-resident table and draws the
// step's timesteps t ~ U{0..S-1} and noise eps ~ N(0,1) from a counter-based generator, so `main.py train` feeds the
// fused step without host random numbers, host tensors or PCIe traffic.
//
// Generator: Philox4x32-10 (Salmon et al., SC'11).  counter = (block, step, stream, domain), key = 64-bit seed; `step`
// is read from the trainer's device-resident step counter (+ offset) so a REPLAYED hipGraph draws fresh numbers; `stream`
// = the data-parallel rank.  The 32-bit words and the timestep indices are bit-exact against oracle/ref_cpu.py
// (philox4x32 / draw_timesteps, pinned by the Random123 known answers); normals are Box-Muller over those words with the
// hardware log2 / sin / cos (v_log_f32, v_sin_f32, v_cos_f32: about 1e-6 absolute; a noise sample needs no more).
//
// HBM-bound streaming kernel: per element 2 B (bf16) read of x0 + 2 x 2 B written; 8 elements (16 B of bf16) per lane
// per access, consecutive lanes on consecutive 16-byte pieces.
#include "ib_common.h"

namespace {

constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
constexpr uint32_t kDomainEps = 0u, kDomainT = 1u;

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(kM0, c.x), lo0 = kM0 * c.x;
    const uint32_t hi1 = __umulhi(kM1, c.z), lo1 = kM1 * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += kW0; k1 += kW1;
  }
  return c;
}

// (w, w') -> two N(0,1): u1 = (w + 1) / 2^32 in (0, 1], u2 = w' / 2^32 in [0, 1)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  // -2 ln u1 = -2 ln2 * log2(u1); u1 from the top 24 bits + 1 so the float conversion is exact and never 0
  const float u1 = (float)((a >> 8) + 1u) * 0x1.0p-24f;
  const float u2 = (float)(b >> 8) * 0x1.0p-24f;                  // revolutions: v_sin / v_cos take x / (2 pi)
  const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

struct Draw {
  const void* table; int64_t table_rows, row_pitch;     // x0 table [rows, row_pitch] (NULL: x0 is not touched)
  const int64_t* idx;                                   // [B] window indices into the table
  void* x0; void* eps; int64_t* t;                      // outputs: [B, per], [B, per], [B]   (eps / t may be NULL)
  int64_t B, per;                                       // per = T * D values per window
  uint32_t k0, k1, stream; int32_t step; const int32_t* step_dev;
  int32_t num_train_steps;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void diffusion_draw_kernel(Draw p) {
  const uint32_t step = (uint32_t)(p.step + (p.step_dev ? *p.step_dev : 0));
  const int64_t groups = (p.per + 7) / 8;               // 8 values = two Philox blocks per lane per trip
  const int64_t n = p.B * groups;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p.t && gtid < p.B) {
    const U4 w = philox4x32_10(U4{(uint32_t)gtid, step, p.stream, kDomainT}, p.k0, p.k1);
    p.t[gtid] = (int64_t)__umulhi(w.x, (uint32_t)p.num_train_steps);
  }
  for (int64_t i = gtid; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / groups, g = i % groups;
    const int64_t e0 = b * p.per + 8 * g;               // first element (of the whole batch) this lane produces
    const int valid = (int)((p.per - 8 * g) < 8 ? (p.per - 8 * g) : 8);
    if (p.table) {
      int64_t r = p.idx[b];
      r = r < 0 ? 0 : (r >= p.table_rows ? p.table_rows - 1 : r);
      const T* src = reinterpret_cast<const T*>(p.table) + r * p.row_pitch + 8 * g;   // row_pitch % 8 == 0: aligned
      T* dst = reinterpret_cast<T*>(p.x0) + e0;
      if constexpr (VEC) {
        if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x8_t*>(dst) = *reinterpret_cast<const bf16x8_t*>(src);
        else {
          const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
          *reinterpret_cast<float4*>(dst) = a; *reinterpret_cast<float4*>(dst + 4) = c;
        }
      } else {
        for (int e = 0; e < valid; ++e) dst[e] = src[e];
      }
    }
    if (p.eps) {
      // the noise of element e (index inside the whole [B, per] batch, windows back to back) comes from block e / 4:
      // with per % 8 == 0 a lane owns blocks 2q and 2q+1 of its eight elements; the generic path looks each element up
      float z[8];
      if constexpr (VEC) {
        const uint32_t q = (uint32_t)(e0 >> 2);
        const U4 w0 = philox4x32_10(U4{q, step, p.stream, kDomainEps}, p.k0, p.k1);
        const U4 w1 = philox4x32_10(U4{q + 1u, step, p.stream, kDomainEps}, p.k0, p.k1);
        box_muller(w0.x, w0.y, z[0], z[1]); box_muller(w0.z, w0.w, z[2], z[3]);
        box_muller(w1.x, w1.y, z[4], z[5]); box_muller(w1.z, w1.w, z[6], z[7]);
        T* dst = reinterpret_cast<T*>(p.eps) + e0;
        if constexpr (sizeof(T) == 2) {
          bf16x8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)z[e];
          *reinterpret_cast<bf16x8_t*>(dst) = o;
        } else {
          *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(z[4], z[5], z[6], z[7]);
        }
      } else {
        T* dst = reinterpret_cast<T*>(p.eps) + e0;
        for (int e = 0; e < valid; ++e) {
          const int64_t el = e0 + e;
          const U4 w = philox4x32_10(U4{(uint32_t)(el >> 2), step, p.stream, kDomainEps}, p.k0, p.k1);
          float za, zb;
          if ((el & 2) == 0) box_muller(w.x, w.y, za, zb); else box_muller(w.z, w.w, za, zb);
          dst[e] = ib_from_f32<T>((el & 1) ? zb : za);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void philox_words_kernel(uint32_t* out, int64_t blocks, uint32_t k0, uint32_t k1,
                                                           uint32_t step, uint32_t stream, uint32_t domain) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < blocks; i += (int64_t)gridDim.x * blockDim.x) {
    const U4 w = philox4x32_10(U4{(uint32_t)i, step, stream, domain}, k0, k1);
    reinterpret_cast<uint4*>(out)[i] = make_uint4(w.x, w.y, w.z, w.w);
  }
}

}  // namespace

extern "C" int ib_diffusion_draw(const void* table, int64_t table_rows, int64_t row_pitch, const int64_t* idx,
                                 void* x0_out, void* eps_out, int64_t* t_out, int64_t B, int64_t per,
                                 int32_t num_train_steps, uint64_t seed, int32_t step, const int32_t* step_dev,
                                 uint32_t stream_id, int dtype, ib_stream_t stream) {
  if (B <= 0 || per <= 0 || (!eps_out && !t_out && !table)) return IB_E_ARG;
  if (table && (!idx || !x0_out || table_rows <= 0 || row_pitch < per || row_pitch % 8 != 0)) return IB_E_ARG;
  if (t_out && num_train_steps <= 0) return IB_E_ARG;
  if (dtype != IB_F32 && dtype != IB_BF16) return IB_E_DTYPE;
  if (B * ((per + 3) / 4 + 1) >= (int64_t)1 << 32) return IB_E_UNSUPPORTED;     // block index is one 32-bit counter word
  const uintptr_t align = reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(x0_out) |
                          reinterpret_cast<uintptr_t>(eps_out);
  const bool vec = per % 8 == 0 && align % 16 == 0;
  Draw p{table, table_rows, row_pitch, idx, x0_out, eps_out, t_out, B, per,
         (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), stream_id, step, step_dev, num_train_steps};
  const dim3 grid(ib_grid_1d(B * ((per + 7) / 8), 256, 256 * 16)), block(256);
  // the timestep draw rides in the first B lanes of the grid: the grid must hold them
  if (t_out && (int64_t)grid.x * 256 < B) return IB_E_UNSUPPORTED;
  if (dtype == IB_BF16) {
    if (vec) hipLaunchKernelGGL((diffusion_draw_kernel<bf16_t, true>), grid, block, 0, ib_s(stream), p);
    else hipLaunchKernelGGL((diffusion_draw_kernel<bf16_t, false>), grid, block, 0, ib_s(stream), p);
  } else {
    if (vec) hipLaunchKernelGGL((diffusion_draw_kernel<float, true>), grid, block, 0, ib_s(stream), p);
    else hipLaunchKernelGGL((diffusion_draw_kernel<float, false>), grid, block, 0, ib_s(stream), p);
  }
  IB_CHECK_LAUNCH();
  return IB_OK;
}

extern "C" int ib_philox_words(uint32_t* out, int64_t blocks, uint64_t seed, uint32_t step, uint32_t stream_id,
                               uint32_t domain, ib_stream_t stream) {
  if (!out || blocks <= 0 || blocks >= (int64_t)1 << 32 || (reinterpret_cast<uintptr_t>(out) % 16)) return IB_E_ARG;
  hipLaunchKernelGGL(philox_words_kernel, dim3(ib_grid_1d(blocks, 256)), dim3(256), 0, ib_s(stream), out, blocks,
                     (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), step, stream_id, domain);
  IB_CHECK_LAUNCH();
  return IB_OK;
}
