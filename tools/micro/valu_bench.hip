// VALU issue-rate micro-benchmark for gfx950 (tools/micro: measurement only, not part of the library).
// One workgroup per CU; W waves per SIMD (256 * W threads); each wave runs a long unrolled chain of independent
// instructions of one kind; prints cycles per wave-instruction per SIMD (s_memtime) for W = 1, 2.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_bench.hip -o gpurun_out/valu_bench && gpurun_out/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(2))) float f2;

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  float a[8]; f2 b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = f2{a[i], a[i] + 0.5f}; }
  const float c = 1.0001f, d = 0.0001f;
  const f2 c2 = f2{c, c}, d2 = f2{d, d};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c2), "v"(d2));
        if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
        if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(d2));
        if constexpr (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if constexpr (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if constexpr (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if constexpr (KIND == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if constexpr (KIND == 8) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if constexpr (KIND == 9) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
        if constexpr (KIND == 10) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a[i]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i][0] + b[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  const int iters = 2000;
  for (int waves = 4; waves <= 16; waves *= 2) {           // 1, 2, 4 waves per SIMD
    hipMemset(cyc, 0, 256 * 16 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_wave = mx / (iters * 32.0);
    printf("%-22s waves/SIMD %d: %.2f cyc per wave-instruction, %.2f cyc per instruction per SIMD\n", name, waves / 4,
           per_wave, per_wave / (waves / 4));
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<3>("v_pk_add_f32"); run<4>("v_exp_f32");
  run<5>("v_rcp_f32"); run<6>("v_mul_f32"); run<7>("v_cvt_pk_bf16_f32"); run<8>("v_add_f32_dpp quad_perm");
  run<9>("v_lshlrev_b32"); run<10>("v_and_b32 literal");
  return 0;
}
