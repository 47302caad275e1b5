// internal interface of gemm_nt.hip (the large-M bf16 NT GEMM); called from the Linear entry points of gemm.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// C[M,N] = fwd_act(A[M,K] B[N,K]^T + bias)            (bwd_act == IB_ACT_NONE)
// C[M,N] = (A B^T) * bwd_act'(aux) + addend           (otherwise; bias must be NULL)
// IB_E_UNSUPPORTED when the problem does not qualify (alignment, sizes, dtype): nothing was launched.
int ib_gemm_nt_try(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, int fwd_act,
                   const void* aux, int64_t ldaux, int bwd_act, const void* addend, int64_t ldadd, int64_t M, int64_t N,
                   int64_t K, hipStream_t s);

// split-K form: fp32 partial slabs [splits][M][N] (few output tiles, long reduction); splits = 0: does not qualify
int ib_gemm_nt_splitk_splits(int64_t M, int64_t N, int64_t K);
int ib_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* slab, int splits, int64_t M, int64_t N,
                      int64_t K, hipStream_t s);

// gemm_tn.hip: weight gradients of long reductions, dW_j[N_j,K_j] = dz_j^T x_j as fp32 split slabs, up to 6 problems per
// launch.  ib_gemm_tn_splits: slab count the kernel will use (0 = the problem does not qualify); ib_gemm_tn_multi returns
// IB_E_UNSUPPORTED (nothing launched) unless every problem qualifies.
int ib_gemm_tn_splits(int64_t M, int64_t N, int64_t K, int group);
int ib_gemm_tn_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x, const int64_t* ldx,
                     void* const* workspace, const size_t* workspace_bytes, float* const* dbias_part, int32_t* nslab_out,
                     const int64_t* M, const int64_t* N, const int64_t* K, hipStream_t s, const void* rider = nullptr,
                     int rider_te = 0);
// gemm_tn256.hip: the same contract with 256 x 256 output tiles and ONE split count for the whole group (every N, K a
// multiple of 256, equal M); ib_gemm_tn_multi tries it first when no rider is attached.  IB_E_UNSUPPORTED = nothing launched.
int ib_gemm_tn256_splits(int n, const int64_t* M, const int64_t* N, const int64_t* K);
int ib_gemm_tn256_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x, const int64_t* ldx,
                        void* const* workspace, const size_t* workspace_bytes, float* const* dbias_part, int32_t* nslab_out,
                        const int64_t* M, const int64_t* N, const int64_t* K, hipStream_t s);
// rider: optional TimeBwdParams (time_bwd.h) whose workgroups are appended to the launch (temb = rider_te: 128 or 32)

// gemm_f32_small.hip: fp32 forward / dgrad of batches of a few rows (16 x 16 tiles, the four waves split the reduction).
// IB_E_UNSUPPORTED = nothing launched.
int ib_f32_small_fwd_try(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int act, float* y,
                         int64_t ldy, float* z, int64_t ldz, int64_t M, int64_t N, int64_t K, hipStream_t s);
int ib_f32_small_dgrad_try(const float* dz, int64_t lddz, const float* w, int64_t ldw, int act, const float* aux,
                           int64_t ldaux, const float* addend, int64_t ldadd, float* dx, int64_t lddx, int64_t M, int64_t N,
                           int64_t K, hipStream_t s);
int ib_f32_small_wgrad_bias_try(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, int64_t lddw, float* dbias,
                                int accumulate, int64_t M, int64_t N, int64_t K, hipStream_t s);
