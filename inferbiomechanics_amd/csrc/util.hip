// Version / error strings, hipGraph capture helpers, HIP-event timing, instruction self-tests.
#include "ib_common.h"

extern "C" int ib_version(void) { return 100; }

thread_local int g_ib_last_path = IB_PATH_NONE;
extern "C" int ib_debug_last_path(void) { const int v = g_ib_last_path; g_ib_last_path = IB_PATH_NONE; return v; }   // read and clear

extern "C" const char* ib_error_string(int code) {
  switch (code) {
    case IB_OK: return "ok";
    case IB_E_ARG: return "invalid argument (null pointer, non-positive size or leading dimension too small)";
    case IB_E_DTYPE: return "unsupported dtype";
    case IB_E_LAUNCH: return "HIP kernel launch failed";
    case IB_E_WORKSPACE: return "workspace missing or too small";
    case IB_E_UNSUPPORTED: return "shape not supported by this kernel";
    default: return "unknown error";
  }
}

extern "C" int ib_graph_begin(ib_stream_t stream) {
  return hipStreamBeginCapture(ib_s(stream), hipStreamCaptureModeThreadLocal) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}
extern "C" int ib_graph_end(ib_stream_t stream, void** graph_exec_out) {
  if (!graph_exec_out) return IB_E_ARG;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture(ib_s(stream), &g) != hipSuccess || !g) return IB_E_LAUNCH;
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return IB_E_LAUNCH;
  *graph_exec_out = ge;
  return IB_OK;
}
extern "C" int ib_graph_launch(void* graph_exec, ib_stream_t stream) {
  if (!graph_exec) return IB_E_ARG;
  return hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), ib_s(stream)) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}
extern "C" int ib_graph_destroy(void* graph_exec) {
  if (!graph_exec) return IB_E_ARG;
  return hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}

// A stream of the library's own (non-blocking, current device).  The host side runs its side branches, its capture stream
// and its trainer stream on these instead of torch.cuda.Stream(): torch hands those out round-robin from a pool of 32 per
// device that c10d's communication stream comes from too -- the 33rd stream of a process IS an earlier one, and a side branch
// that aliases c10d's stream pulls it into a capture (tests/test_ddp_rccl_gpu.py failed one run in ten that way).
extern "C" int ib_stream_create(void** stream_out) {
  if (!stream_out) return IB_E_ARG;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || !s) return IB_E_LAUNCH;
  *stream_out = s;
  return IB_OK;
}
extern "C" int ib_stream_destroy(void* stream) {
  if (!stream) return IB_E_ARG;
  return hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}

extern "C" int ib_event_create(void** ev_out) {
  if (!ev_out) return IB_E_ARG;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return IB_E_LAUNCH;
  *ev_out = e;
  return IB_OK;
}
extern "C" int ib_event_record(void* ev, ib_stream_t stream) {
  if (!ev) return IB_E_ARG;
  return hipEventRecord(reinterpret_cast<hipEvent_t>(ev), ib_s(stream)) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}
extern "C" int ib_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
  if (!ev_start || !ev_stop || !ms_out) return IB_E_ARG;
  if (hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev_stop)) != hipSuccess) return IB_E_LAUNCH;
  return hipEventElapsedTime(ms_out, reinterpret_cast<hipEvent_t>(ev_start), reinterpret_cast<hipEvent_t>(ev_stop)) ==
                 hipSuccess ? IB_OK : IB_E_LAUNCH;
}
extern "C" int ib_event_destroy(void* ev) {
  if (!ev) return IB_E_ARG;
  return hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? IB_OK : IB_E_LAUNCH;
}

// Probe of ds_read_b64_tr_b16 (the transposing LDS read the bf16 k-strided GEMM operands rely on):
// input  in[64 rows][16 cols] bf16 (row-major, a [k][row] image of 64 k x 16 rows),
// output out[lane][4]: what lane gets from ONE read whose address follows gemm.hip::read_frag_bf16
// with kbase = 0, rbase = 0, i.e. expected out[lane][q] = in[8*(lane/16) + q][lane%16].
namespace {
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
__global__ void selftest_tr16_kernel(const short* __restrict__ in, short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short img[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) img[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int kr = 8 * (lane >> 4) + q;
  const short* a0 = img + kr * 16 + 4 * pp;
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(a0));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
}  // namespace

extern "C" int ib_selftest_tr16(const void* in_bf16_64x16, void* out_bf16_64x4, ib_stream_t stream) {
  if (!in_bf16_64x16 || !out_bf16_64x4) return IB_E_ARG;
  hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, ib_s(stream), (const short*)in_bf16_64x16,
                     (short*)out_bf16_64x4);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// TIMING-ONLY (tools/timeline.py): one thread writes the 100 MHz wall clock into *slot when the stream reaches it
namespace {
__global__ void stamp_kernel(long long* slot) { *slot = wall_clock64(); }
}  // namespace
extern "C" int ib_debug_stamp(void* slot, ib_stream_t stream) {
  if (!slot) return IB_E_ARG;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ib_s(stream), reinterpret_cast<long long*>(slot));
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// Input slots: a device array of pointers that captured kernels dereference at their start (ib_mlp_chain_train /
// ib_mlp_chain_prep `in_slots`).  The pointers travel as KERNEL ARGUMENTS of this one-thread launch, so any number of
// steps may be queued ahead without a host-side buffer to keep alive.
namespace {
struct Ptrs4 { const void* p[4]; };
__global__ void set_ptrs_kernel(const void** slots, Ptrs4 v, int n) {
  for (int i = 0; i < n; ++i) slots[i] = v.p[i];
}
}  // namespace
extern "C" int ib_set_ptrs(void* slots, int n, const void* const* ptrs, ib_stream_t stream) {
  if (!slots || !ptrs || n <= 0 || n > 4) return IB_E_ARG;
  Ptrs4 v{};
  for (int i = 0; i < n; ++i) v.p[i] = ptrs[i];
  hipLaunchKernelGGL(set_ptrs_kernel, dim3(1), dim3(1), 0, ib_s(stream), reinterpret_cast<const void**>(slots), v, n);
  IB_CHECK_LAUNCH();
  return IB_OK;
}

// y[i] *= *scale  (the upstream gradient of a scalar loss arriving in a loss plugin's backward: autograd hands it over as a
// DEVICE scalar, the kernel that wrote d loss / d outputs ran in the forward).  In place, 16-byte accesses where aligned.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(T* __restrict__ y, const float* __restrict__ scale, int64_t n) {
  const float s = *scale;
  constexpr int V = 16 / sizeof(T);
  const int64_t nv = n / V;
  typedef __attribute__((ext_vector_type(V))) T vec_t;
  vec_t* yv = reinterpret_cast<vec_t*>(y);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    vec_t v = yv[i];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = ib_from_f32<T>(ib_to_f32(v[e]) * s);
    yv[i] = v;
  }
  for (int64_t i = nv * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = ib_from_f32<T>(ib_to_f32(y[i]) * s);
}
}  // namespace
extern "C" int ib_scale_by_device_scalar(void* y, const float* scale, int64_t n, int dtype, ib_stream_t stream) {
  if (!y || !scale || n < 0) return IB_E_ARG;
  if (n == 0) return IB_OK;
  if ((reinterpret_cast<uintptr_t>(y) & 15) != 0) return IB_E_ARG;
  const int grid = ib_grid_1d(n, 256 * 8);
  if (dtype == IB_F32) hipLaunchKernelGGL((scale_by_scalar_kernel<float>), dim3(grid), dim3(256), 0, ib_s(stream), (float*)y, scale, n);
  else if (dtype == IB_BF16) hipLaunchKernelGGL((scale_by_scalar_kernel<bf16_t>), dim3(grid), dim3(256), 0, ib_s(stream), (bf16_t*)y, scale, n);
  else return IB_E_DTYPE;
  IB_CHECK_LAUNCH();
  return IB_OK;
}
