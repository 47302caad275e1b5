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
