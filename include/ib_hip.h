/* ib_hip.h -- C-ABI of the MI355X (gfx950) hot-path library `libib_hip.so`.
 *
 * The reference (jbejjani2022/InferBiomechanics) has NO FFI / native interface: its hot path is
 * stock PyTorch ops called from Python.  Each entry point below therefore
 * cites the reference call site whose torch op(s) it replaces (paths relative to the reference
 * root).  Conventions ("[BUILD-DEFINED] C-ABI extension"):
 *   - plain pointers + sizes only, no torch types; every pointer is a DEVICE pointer that the
 *     caller (PyTorch) owns and that is only borrowed for the duration of the call;
 *   - every call enqueues on `stream` (a hipStream_t; NULL = default stream) and returns without
 *     synchronising; workspace is supplied by the caller; no global mutable state -> re-entrant;
 *   - returns 0 on success, a negative IB_E_* code on error; never throws;
 *   - `dtype` selects the STORAGE type of activations / weights operands: IB_F32 (parity mode,
 *     exact-f32 MFMA) or IB_BF16 (throughput mode, bf16 storage, fp32 accumulate).  Bias, LayerNorm
 *     affine, statistics, gradients of parameters, optimizer state are always fp32.
 *   - all matrices are row-major with an explicit leading dimension (elements, not bytes).
 */
#ifndef IB_HIP_H
#define IB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ib_stream_t; /* hipStream_t */

enum { IB_F32 = 0, IB_BF16 = 1 };
/* ACTIVATION_FUNCS: src/models/FeedForwardRegressionBaseline.py:7-11 (+ silu, build-defined) */
enum { IB_ACT_NONE = 0, IB_ACT_RELU = 1, IB_ACT_TANH = 2, IB_ACT_SIGMOID = 3, IB_ACT_SILU = 4,
       IB_ACT_ELU = 5 /* torch.nn.ELU(alpha=1), Groundlink.py:48,57; derivative operand = the layer OUTPUT */ };
/* --opt-type choices: src/cli/train.py:183-194 */
enum { IB_OPT_SGD = 0, IB_OPT_ADAM = 1, IB_OPT_RMSPROP = 2, IB_OPT_ADAGRAD = 3, IB_OPT_ADADELTA = 4,
       IB_OPT_ADAMAX = 5 };
enum { IB_OK = 0, IB_E_ARG = -1, IB_E_DTYPE = -2, IB_E_LAUNCH = -3, IB_E_WORKSPACE = -4,
       IB_E_UNSUPPORTED = -5 };

int ib_version(void);
const char* ib_error_string(int code);

/* ---- Linear family: nn.Linear fwd/bwd (FeedForwardRegressionBaseline.py:73,113;
 *      TransformerBaseline.py:12-18,29,34) --------------------------------------------------- */

/* y[M,N] = act( x[M,K] . w[N,K]^T + bias[N] + add_div[m / seg, :] + add_mod[m % seg, :] )
 * bias / add_div / add_mod / z may be NULL.  z (if given) receives the pre-activation (needed for
 * silu backward).  add_div broadcasts one row per window (the diffusion time embedding), add_mod one
 * row per frame (the projected per-frame embedding, TransformerBaseline.py:119-126). */
int ib_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                  const void* add_div, int64_t ld_add_div, const void* add_mod, int64_t ld_add_mod,
                  int64_t seg, int act, void* y, int64_t ldy, void* z, int64_t ldz,
                  int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream);

/* dx[M,K] = ( dz[M,N] . w[N,K] ) * act'(aux[M,K]) + addend[M,K]   (autograd of the layer BELOW and
 * the residual-path gradient of the post-norm block fused in the epilogue).  aux = the lower layer's
 * OUTPUT for relu/tanh/sigmoid, its PRE-activation for silu; ignored for IB_ACT_NONE.  addend may be
 * NULL. */
int ib_linear_dgrad(const void* dz, int64_t lddz, const void* w, int64_t ldw, int act_below,
                    const void* aux, int64_t ldaux, const void* addend, int64_t ldadd, void* dx, int64_t lddx,
                    int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream);

/* The same product with the weight handed over TRANSPOSED, wt[K,N] (a bf16 copy the caller refreshes once per step with
 * ib_transpose_multi): both operands are then contiguous along the reduction and large-M problems (the transformer
 * denoiser's training shapes) take the 256 x 128 LDS-DMA kernel.  Returns IB_E_UNSUPPORTED -- nothing launched -- when the
 * problem does not qualify (small M, unaligned operands, fp32, silu): call ib_linear_dgrad then. */
int ib_linear_dgrad_wt(const void* dz, int64_t lddz, const void* wt, int64_t ldwt, int act_below, const void* aux,
                       int64_t ldaux, const void* addend, int64_t ldadd, void* dx, int64_t lddx, int64_t M, int64_t N,
                       int64_t K, int dtype, ib_stream_t stream);
/* dst_i[c][r] = src_i[r][c] for n <= 32 bf16 matrices in one launch (host arrays of pointers / sizes). */
int ib_debug_set_nt_prof(void* stamps);   /* TIMING-ONLY: [workgroups][16] int64 wall-clock stamps of the NT GEMM (NULL = off) */
int ib_transpose_multi(int n, const void* const* src, const int64_t* lds, void* const* dst, const int64_t* ldd,
                       const int64_t* rows, const int64_t* cols, int dtype, ib_stream_t stream);

/* The same product for FEW rows and a long reduction (the time-embedding MLP's hidden layer: one row per window of the
 * batch; autograd of `nn.Linear` + activation as in ib_linear_dgrad), plus the bias gradient of the layer below:
 * dbias[K] (fp32, may be NULL) (+)= column sums of the stored dx.  One workgroup per 16 output columns sees every row, so
 * the sums need no second launch.  bf16, M <= 256, N <= 1024, N % 128 == 0, K % 16 == 0; IB_E_UNSUPPORTED otherwise
 * (callers fall back to ib_linear_dgrad + ib_segment_colsum). */
int ib_linear_dgrad_skinny(const void* dz, int64_t lddz, const void* w, int64_t ldw, int act_below, const void* aux,
                           int64_t ldaux, void* dx, int64_t lddx, float* dbias, int accumulate,
                           int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream);

/* dw[N,K] (fp32) (+)= dz[M,N]^T . x[M,K]   split over M; partial slabs go to `workspace`
 * (deterministic: slabs are summed in a fixed order by a second kernel, no float atomics). */
size_t ib_linear_wgrad_workspace(int64_t M, int64_t N, int64_t K);
int ib_linear_wgrad(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw, int64_t lddw,
                    int accumulate, void* workspace, size_t workspace_bytes,
                    int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream);
/* Weight AND bias gradient of a Linear layer from one launch, for short reductions (M <= 1024 rows: the reference's
 * regression models at their batch sizes; autograd of nn.Linear, FeedForwardRegressionBaseline.py:68-77): dw[N,K] (+)=
 * dz^T x as in ib_linear_wgrad, dbias[N] (+)= column sums of dz (rows added in order).  bf16, 4-byte aligned operand rows;
 * IB_E_UNSUPPORTED otherwise (callers fall back to ib_linear_wgrad + ib_segment_colsum). */
int ib_linear_wgrad_bias(const void* dz, int64_t lddz, const void* x, int64_t ldx, float* dw, int64_t lddw, float* dbias,
                         int accumulate, int64_t M, int64_t N, int64_t K, int dtype, ib_stream_t stream);

/* y = LayerNorm(res + x W^T + bias) * gamma + beta for small token counts (the DDIM sampler): a K-split GEMM into fp32
 * slabs whose reduction kernel is the LayerNorm (TransformerBaseline.py:29-31 / 34-36: Linear -> add -> norm).  bf16,
 * N % 64 == 0, N <= 1024 (N in {64,128,256,512,1024}), K % 32 == 0, 16-byte aligned operand rows; IB_E_UNSUPPORTED
 * otherwise (use ib_linear_fwd + ib_layernorm_fwd).  a_out / mean / rstd: optional (what a backward pass needs). */
size_t ib_linear_ln_fwd_workspace(int64_t M, int64_t N, int64_t K);
int ib_linear_ln_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const void* res,
                     int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, void* a_out,
                     float* mean, float* rstd, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
                     int64_t K, float eps, int dtype, ib_stream_t stream);
/* The same operation for d = 512 projections (N == K == 512) as ONE launch over panels of ceil(M / 256) <= 32 rows
 * (csrc/linln_panel.hip; the sampler's attention out-projection + residual + LayerNorm1, TransformerBaseline.py:12-13,
 * 29-31): the fp32 GEMM result stays in LDS and is normalised in place -- no slabs, no second launch.  `w_packed` = the
 * weight's [512 x 512] fragment-major image (ib_ffn_chain_pack: the out-projection's image sits at element offset
 * 4 * (ffn / 512) * 512 * 512 of a layer's packed buffer).  bias / res optional.  ib_linear_ln_panel_workgroups returns the
 * workgroup count (0: shape not supported -- M > 8192 or N, K != 512) and the rows per workgroup. */
int ib_linear_ln_panel_workgroups(int64_t M, int64_t N, int64_t K, int32_t* rows_out);
int ib_linear_ln_panel_fwd(const void* x, int64_t ldx, const void* w_packed, const float* bias, const void* res,
                           int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, int64_t M, int64_t N,
                           int64_t K, float eps, ib_stream_t stream);
/* y = x W^T + bias for K == 512 and N a multiple of 512 (a frozen-weight layer's in-projection, TransformerBaseline.py:12-13)
 * at up to 8192 rows as a launch over (panel of rows, 512-column chunk) workgroups, each streaming one packed [512 x 512]
 * image (`w_packed`: N / 512 consecutive images, ib_ffn_chain_pack puts the in-projection's at element offset
 * (4 * (ffn / 512) + 2) * 512 * 512); chunks are cut into 128-column blocks when there are few panels.
 * ib_linear_panel_workgroups: the workgroup count, 0 = unsupported. */
int ib_linear_panel_workgroups(int64_t M, int64_t N, int64_t K);
int ib_linear_panel_fwd(const void* x, int64_t ldx, const void* w_packed, const float* bias, void* y, int64_t ldy, int64_t M,
                        int64_t N, int64_t K, ib_stream_t stream);
/* The feed-forward sublayer of a frozen-weight forward, y = LayerNorm2(x1 + W2 ReLU(W1 x1 + b1) + b2)
 * (TransformerBaseline.py:15-19,33-36), d == 512, ffn a multiple of 512, at most 32768 rows
 * (csrc/linln_panel.hip): a panel of rows is shared by the ffn / 512 workgroups of its hidden chunks, each leaves an fp32
 * partial product in `workspace` (ib_ffn_infer_workspace bytes); the slab reduction of ib_linear_ln_fwd (bias + residual +
 * LayerNorm, partials added in chunk order) finishes it: two launches, no [M, ffn] activation in HBM.  x1 / y: contiguous
 * [M, 512] bf16; `packed`: the layer's image from ib_ffn_chain_pack.  ib_ffn_infer_workgroups returns the workgroup count
 * of the first launch (0: unsupported), the rows per panel and the panel count. */
size_t ib_ffn_infer_workspace(int64_t M, int64_t d, int64_t ffn);
int ib_ffn_infer_workgroups(int64_t M, int64_t d, int64_t ffn, int32_t* rows_out, int32_t* panels_out);
int ib_ffn_infer_fwd(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                     const float* beta, void* y, void* workspace, size_t workspace_bytes, int64_t M, int64_t d, int64_t ffn,
                     float eps, ib_stream_t stream);
/* Deferred form for a step that computes several weight gradients: ib_linear_wgrad_slabs writes only the split-M
 * partial slabs ([*nslab_out][N][K] fp32, workspace of ib_linear_wgrad_slabs_workspace bytes); ONE
 * ib_slab_reduce_multi launch (n <= 8 gradients, K % 4 == 0, host arrays) then sums every slab set into its dw. */
size_t ib_linear_wgrad_slabs_workspace(int64_t M, int64_t N, int64_t K);
int ib_linear_wgrad_slabs(const void* dz, int64_t lddz, const void* x, int64_t ldx, void* workspace,
                          size_t workspace_bytes, int* nslab_out, int64_t M, int64_t N, int64_t K, int dtype,
                          ib_stream_t stream);
/* n <= 6 independent weight-gradient problems in ONE launch (bf16 ring kernel only: IB_E_UNSUPPORTED otherwise, the
 * caller then issues them one by one); host arrays */
int ib_linear_wgrad_slabs_multi(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                int32_t* nslab_out, const int64_t* M, const int64_t* N, const int64_t* K, int dtype,
                                ib_stream_t stream);
/* The same launch that ALSO leaves the bias gradients' split-M partial sums: dbias_part[j] (may be NULL per problem) is an
 * fp32 [32][N_j] array whose rows 0 .. nslab_out[j]-1 receive sum over the slice's rows of dz (one extra MFMA per row tile
 * against an all-ones fragment in the first column tile's workgroups); the caller's final reduction (ib_colsum_segments /
 * ib_optim_step_sources) adds the rows up in a fixed order.  Replaces ib_segment_colsum over the whole dz matrix per layer. */
int ib_linear_wgrad_slabs_multi_bias(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                     const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                     float* const* dbias_part, int32_t* nslab_out, const int64_t* M, const int64_t* N,
                                     const int64_t* K, int dtype, ib_stream_t stream);
int ib_slab_reduce_multi(int n, const void* const* slabs, const int32_t* nslab, float* const* dw,
                         const int64_t* lddw, const int32_t* N, const int32_t* K, int accumulate,
                         ib_stream_t stream);

/* out[s, n] (fp32) = sum over rows m of segment s of x[m, n]; mode 0: s = m / seg (M/seg segments,
 * bias grads / per-window time-embedding grads), mode 1: s = m % seg (seg segments, per-frame
 * embedding grads). */
int ib_segment_colsum(const void* x, int64_t ldx, float* out, int64_t ldo, void* out_bf16, int64_t ld_bf16,
                      int64_t M, int64_t N, int64_t seg, int mode, int accumulate, int dtype, ib_stream_t stream);
/* out_bf16 (optional): a bf16 copy of the sums, i.e. the operand of the GEMM that consumes them (saves a cast launch) */

/* ---- LayerNorm (+ fused residual add / pre-activation): nn.LayerNorm, TransformerBaseline.py:21-22,31,36.
 * v = act(x + add_div[m / seg, :]) (+ res);  y = (v - mean)/sqrt(var + eps) * gamma + beta;  mean/rstd [M] saved
 * (fp32).  add_div (optional, may be NULL) broadcasts one row per window BEFORE the activation (the diffusion
 * time embedding; having it here instead of in the GEMM epilogue takes the time-MLP off the critical path). */
int ib_layernorm_fwd(const void* x, const void* res, int act, const float* gamma, const float* beta,
                     void* y, float* mean, float* rstd, const void* add_div, int64_t ld_add_div, int64_t seg,
                     int64_t M, int64_t N, float eps, int dtype, ib_stream_t stream);
/* dx = grad wrt x (through act), dres (optional, may alias nothing) = grad wrt res (= grad wrt v);
 * dgamma/dbeta partials are written to `partial` [(2*nparts), N] fp32, then summed into dgamma/dbeta. */
/* dgamma == dbeta == NULL defers the fixed-order reduction of the parameter-gradient partials (they stay in
 * `workspace`); ib_layernorm_bwd_reduce finishes it later, e.g. on a forked stream off the critical path. */
size_t ib_layernorm_bwd_workspace(int64_t M, int64_t N);
int ib_layernorm_bwd_reduce(const void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta,
                            int accumulate, int64_t M, int64_t N, ib_stream_t stream);
int ib_layernorm_bwd(const void* dy, const void* x, const void* res, int act, const float* gamma,
                     const float* mean, const float* rstd, void* dx, void* dres, float* dgamma,
                     float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                     const void* add_div, int64_t ld_add_div, int64_t seg,
                     int64_t M, int64_t N, int dtype, ib_stream_t stream);

/* ---- temporal self-attention: nn.MultiheadAttention core, TransformerBaseline.py:12-13,29.
 * qkv [B, T, 3*H*dh] packed as the in-proj produces it (q | k | v, head h at columns h*dh..),
 * out [B, T, H*dh];  softmax(q k^T / sqrt(dh)) v per (window, head); no mask.  lse [B,H,T] fp32. */
int ib_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H,
                     int64_t dh, int dtype, ib_stream_t stream);
int ib_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                     int64_t B, int64_t T, int64_t H, int64_t dh, int dtype, ib_stream_t stream);
/* the same with nn.MultiheadAttention(dropout = p)'s dropout on the softmax probabilities (TransformerBaseline.py:12-13;
 * train mode only -- the caller passes p = 0 in eval mode): out = (softmax(..) x mask / (1 - p)) v.  The mask of
 * (window, head, query, key) is a counter-based hash of (seed, step or *step_dev, indices): the backward regenerates the
 * forward's draw from the same (seed, step), nothing T x T is stored.  p in [0, 1); p = 0 is ib_attention_fwd / _bwd.
 * ib_attention_drop_mask writes the multipliers (0 or 1 / (1 - p)) of one draw as fp32 [B, H, T, T] (tests: a float64
 * restatement of a train-mode layer needs the masks the kernels used). */
int ib_attention_fwd_drop(const void* qkv, void* out, float* lse, int64_t B, int64_t T, int64_t H, int64_t dh,
                          float p, uint32_t seed, int32_t step, const int32_t* step_dev, int dtype,
                          ib_stream_t stream);
int ib_attention_bwd_drop(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                          int64_t B, int64_t T, int64_t H, int64_t dh, float p, uint32_t seed, int32_t step,
                          const int32_t* step_dev, int dtype, ib_stream_t stream);
int ib_attention_drop_mask(float* mask, int64_t B, int64_t T, int64_t H, float p, uint32_t seed, int32_t step,
                           const int32_t* step_dev, ib_stream_t stream);

/* ---- input packing: torch.concat x10 + reshape, FeedForwardRegressionBaseline.py:97-108.
 * out[b, f, off_k + c] = cast(in_k[b, f, c]); inputs fp32 contiguous [B*F, width_k]. */
int ib_concat_keys(const float* const* inputs, const int32_t* widths, int32_t nkeys, void* out,
                   int64_t rows, int dtype_out, ib_stream_t stream);
int ib_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, ib_stream_t stream);
/* 2-D strided copy/cast (used to split the model output / refresh bf16 weight shadows) */
int ib_cast2d(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype,
              int64_t rows, int64_t cols, ib_stream_t stream);

/* ---- loss plugin: RegressionLossEvaluator.__call__ steps 1-2.2, src/loss/RegressionLossEvaluator.py:184-263.
 * Each tensor is [B, F, C] with element (b,f,c) at ptr[b*bs + f*C + c] (bs = batch stride; the model
 * output views of FeedForwardRegressionBaseline.py:116-121 have bs = 30*F).  C = 6,6,6,12 for
 * cop, force, torque(moment), wrench; o_bs / g_bs are HOST arrays of the 4 batch strides in that order.  comp_w[30] = 0/1 selection of --predict-*-components in the
 * order force[6], cop[6], moment[6], wrench[12] (RegressionLossEvaluator.py:217-220).
 * result[64] fp32: [0] loss, [1..6] force vec, [7..12] cop vec, [13..18] moment vec, [19..30] wrench
 * vec, [31..36] metrics force, moment, cop, wrench, wrench_moment, com_acc (last frame only, :136).
 * grads (optional, same layout as the outputs, dtype `dtype`) = d loss / d output. */
size_t ib_regression_loss_workspace(int64_t B, int64_t F);
int ib_regression_loss(const void* o_cop, const void* o_force, const void* o_torque, const void* o_wrench,
                       const int64_t* o_bs, const float* l_cop, const float* l_force, const float* l_torque,
                       const float* l_wrench, const float* comp_w, float threshold, float* result,
                       void* g_cop, void* g_force, void* g_torque, void* g_wrench, const int64_t* g_bs,
                       void* workspace, size_t workspace_bytes, int64_t B, int64_t F, int dtype,
                       ib_stream_t stream);

/* the same with explicit FRAME strides (elements between frames of an output / gradient key; NULL = dense 6,6,6,12):
 * Groundlink's output is [B,F,30] with the four keys interleaved per frame (Groundlink.py:151-156 slices the last dim),
 * i.e. frame stride 30 for every key. */
int ib_regression_loss_strided(const void* o_cop, const void* o_force, const void* o_torque, const void* o_wrench,
                               const int64_t* o_bs, const int64_t* o_fs, const float* l_cop, const float* l_force,
                               const float* l_torque, const float* l_wrench, const float* comp_w, float threshold,
                               float* result, void* g_cop, void* g_force, void* g_torque, void* g_wrench,
                               const int64_t* g_bs, const int64_t* g_fs, void* workspace, size_t workspace_bytes,
                               int64_t B, int64_t F, int dtype, ib_stream_t stream);

/* The evaluator's four static helpers as entry points of their own (src/loss/RegressionLossEvaluator.py:73-158; the
 * training step uses the fused ib_regression_loss above).  o / l: contiguous [B, F, C] (rows = B*F), fp32 or bf16.
 *   ib_sqdiff_mean:      out[c] = mean over rows of (o - l)^2                 get_squared_diff_mean_vector :73-83
 *   ib_sqdiff_mean_bwd:  d_o[r, c] = dout[c] * 2 (o - l) / rows               (its autograd)
 *   ib_mask_by_threes:   mask[.., 3k..3k+2] = (||t[.., 3k..3k+2]|| > threshold) as fp32, n = element count   :85-108
 *   ib_mean_norm_error:  mean over (window, chunk of vec_size) of ||(o - l)[window, LAST frame, chunk]|| :119-141;
 *                        fold_halves: the two halves of the last dimension are added first (get_com_acc_error :143-158) */
int ib_sqdiff_mean(const void* o, const void* l, float* out, int64_t rows, int64_t C, int dtype, ib_stream_t stream);
int ib_sqdiff_mean_bwd(const void* o, const void* l, const float* dout, void* d_o, int64_t rows, int64_t C, int dtype,
                       ib_stream_t stream);
int ib_mask_by_threes(const void* t, float* mask, int64_t n, float threshold, int dtype, ib_stream_t stream);
int ib_mean_norm_error(const void* o, const void* l, float* out, int64_t B, int64_t F, int64_t C, int vec_size,
                       int fold_halves, int dtype, ib_stream_t stream);

/* diffusion eps-prediction loss [BUILD-DEFINED; no reference counterpart, SURVEY.md §0.1]:
 * loss = mean((pred - target)^2) -> result[0]; dpred = 2 (pred - target) / n. */
size_t ib_mse_loss_workspace(int64_t n);
int ib_mse_loss(const void* pred, const void* target, void* dpred, float* result, void* workspace,
                size_t workspace_bytes, int64_t n, int dtype, ib_stream_t stream);
/* the same in two halves, so the scalar reduction can run off the critical path (a forked stream): the first
 * writes dpred and the per-block partial sums into `workspace`, the second reduces them into result[0]. */
int ib_mse_loss_partial(const void* pred, int64_t ld_pred, const void* target, void* dpred, int64_t ld_dpred,
                        void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols, int dtype,
                        ib_stream_t stream);   /* pred / dpred: [rows, cols] with leading dimensions; target contiguous */
int ib_mse_loss_finalize(const void* workspace, size_t workspace_bytes, float* result, int64_t n, ib_stream_t stream);

/* ---- optimizer step: torch.optim.{SGD,Adam,RMSprop,Adagrad,Adadelta,Adamax}(lr) defaults,
 * src/cli/train.py:183-197,284.  One launch over a FLAT fp32 parameter buffer.  g is multiplied by
 * grad_scale first (DDP mean = 1/world, train.py:175).  `step_dev` (optional, int32 device scalar)
 * holds the step count and is used for the bias corrections, so a captured hipGraph replays correctly: without a
 * ticket the step is *step_dev + step (step = 0: the counter was advanced with ib_counter_add BEFORE this call;
 * step = 1: a launch over PART of the buffer inside a self-counting step, see `ticket`); if NULL, `step` (1-based) is
 * used.  shadow (optional) receives bf16 copies of p. */
int ib_optim_step(int opt, float* p, const float* g, float* s1, float* s2, int64_t n, float lr,
                  float grad_scale, int32_t step, int32_t* step_dev, int32_t* ticket, void* shadow_bf16,
                  ib_stream_t stream);
/* ib_optim_step whose gradient, for up to 28 ranges [start, start+len) of the flat buffer, is still a set of partial sums:
 * kind 1 = split-M slabs (base = [count][len] fp32, `stride` elements apart), kind 2 = column sums over `count` rows of a
 * row-major partial array (base = first column, `stride` = row pitch), times scale, kind 3 = nothing to do (the range was
 * updated by an earlier launch of this step over that part of the buffer; base / count ignored).  The optimizer sums them itself in the
 * fixed order of ib_step_reduce; elsewhere it reads g.  Optionally also writes *loss_out = loss_scale * sum of one column.
 * Single-GPU steps only (an all-reduce needs the reduced gradient in memory).  Host arrays; n % 4 == 0. */
int ib_optim_step_sources(int opt, float* p, const float* g, float* s1, float* s2, int64_t n, float lr,
                          float grad_scale, int32_t step, int32_t* step_dev, int32_t* ticket, void* shadow_bf16,
                          int nsrc, const int64_t* start, const int64_t* len, const int32_t* kind,
                          const void* const* base, const int64_t* stride, const int32_t* count, const float* scale,
                          const float* loss_col, int64_t loss_ld, int64_t loss_rows, float loss_scale, float* loss_out,
                          ib_stream_t stream);
/* ticket (optional): self-counting mode -- *step_dev then holds the number of COMPLETED steps, the kernel uses
 * *step_dev + 1 and its last-exiting block publishes it (no separate counter launch).  The buffer holds
 * ib_optim_ticket_words() zero-initialised int32 words (a top word + 32 sub-counters, one 128-byte line each: the exit
 * tickets are drawn in two levels so no single address serialises the grid); the kernel leaves it zeroed. */
int ib_optim_ticket_words(void);

/* ---- fused token-local half of the post-norm encoder layer (csrc/ffn_chain.hip), bf16, d == 512, ffn a multiple of 512
 * (<= 4096).  Replaces, per layer and direction, the feed-forward sublayer's two nn.Linear GEMMs + residual + nn.LayerNorm
 * (TransformerBaseline.py:15-19,33-36) and -- with the attention epilogue -- the attention out-projection + residual +
 * LayerNorm1 in front of it (:12-13,29-31), and their autograd.  One launch over panels of <= 64 token rows
 * (ib_ffn_chain_workgroups), the hidden width walked in chunks of 512 columns, weights streamed from a packed image:
 *   ib_ffn_chain_pack   packed[l] (ib_ffn_chain_packed_elems bf16 elements each) <- w1[l] = feedforward.0.weight [ffn, d],
 *                       w2[l] = feedforward.2.weight [d, ffn], wo[l] = out_proj.weight [d, d], wqkv[l] = in_proj_weight [3 d, d]
 *                       (wo / wqkv and their entries may be NULL): the forward and transposed fragment-major images of each,
 *                       all layers in ONE launch;
 *   ib_ffn_chain_fwd    y = LN2(x1 + relu(x1 W1^T + b1) W2^T + b2); also stores f1 = relu(..) [M, ffn] (weight-gradient
 *                       operand), s2 = the LayerNorm2 input [M, d], mean / rstd [M], the ReLU bits (`mask`,
 *                       ib_ffn_chain_mask_bytes).  attn != NULL (attention epilogue): the first argument is the LAYER input x
 *                       and x1 = LN1(x + attn Wo^T + bo) is computed here: x1_out [M, d], s1 = its LayerNorm input, mean1 / rstd1;
 *   ib_ffn_chain_bwd    ds2 = LN2-backward(dy) [M, d] (= d f2 and the residual addend), dz1 = (ds2 W2) * relu'(.) [M, ffn],
 *                       dx1 = dz1 W1 + ds2 [M, d]; partial fp32 [2 (4) x workgroups, d]: dgamma2, dbeta2 (, dgamma1, dbeta1) of
 *                       every panel (summed in workgroup order by the optimizer / ib_step_reduce).  s1 != NULL: dx1 is not
 *                       stored; ds1 = LN1-backward(dx1) [M, d] (the out-projection's weight-gradient operand and the layer
 *                       input's residual addend) and dattn = ds1 Wo [M, d] are.  The weight / bias gradients stay GEMMs. */
int ib_ffn_chain_supported(int64_t d, int64_t ffn);
size_t ib_ffn_chain_packed_elems(int64_t d, int64_t ffn);
int ib_ffn_chain_workgroups(int64_t M, int64_t d, int64_t ffn, int* rows_per_wg);
size_t ib_ffn_chain_mask_bytes(int64_t M, int64_t d, int64_t ffn);
int ib_ffn_chain_pack(const void* const* w1, const int64_t* ld1, const void* const* w2, const int64_t* ld2,
                      const void* const* wo, const int64_t* ldo, const void* const* wqkv, const int64_t* ldq,
                      void* const* packed, int layers, int64_t d, int64_t ffn, ib_stream_t stream);
/* qkv_next != NULL (needs the attention epilogue): the NEXT layer's in-projection rides behind LayerNorm2 --
 * qkv_next [M, 3 d] = y . Wqkv_next^T + bqkv_next, from `packed_next` (that layer's packed image, wqkv given to the pack). */
int ib_ffn_chain_fwd(const void* x1, const void* packed, const float* b1, const float* b2, const float* gamma,
                     const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                     const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1, void* x1_out,
                     float* mean1, float* rstd1, const void* packed_next, const float* bqkv_next, void* qkv_next,
                     int64_t M, int64_t d, int64_t ffn, float ln_eps, ib_stream_t stream);
/* dqkv_next != NULL (needs the attention epilogue): dy is not read (may be NULL) but computed in front of LayerNorm2's
 * backward = dqkv_next [M, 3 d] . Wqkv_next + ds1_next [M, d]: the next layer's in-projection dgrad + its residual addend. */
int ib_ffn_chain_bwd(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                     const void* packed, const void* mask, void* ds2, void* dz1, void* dx1, float* partial,
                     const void* s1, const float* mean1, const float* rstd1, const float* gamma1, void* ds1, void* dattn,
                     const void* packed_next, const void* dqkv_next, const void* ds1_next, int64_t M, int64_t d, int64_t ffn,
                     ib_stream_t stream);

/* ---- the temporal self-attention INSIDE the two launches above (round 5): panels of exactly one window of T frames
 * (16 <= T <= 64, M % T == 0; ib_ffn_chain_attn_workgroups = M / T, 0 = unsupported), d = 512 = eight heads of 64.  Wave w of a
 * panel's workgroup owns the 64 in-projection columns of head w, so softmax(Q K^T / 8) V of a (window, head) and its
 * backward (nn.MultiheadAttention's core, TransformerBaseline.py:12-13,29) are wave-private work between the GEMM phases:
 *   ib_ffn_chain_fwd_attn   ib_ffn_chain_fwd (all of its arguments, same meaning) over one-window panels and, with
 *                           attn_next != NULL (needs the QKV tail), the NEXT layer's attention behind the tail:
 *                           attn_next [M, 512], lse_next [M / T, 8, T] (row log-sum-exp, fp32);
 *   ib_ffn_chain_bwd_attn   the attention epilogue form of ib_ffn_chain_bwd continued through THIS layer's attention
 *                           backward (qkv [M, 1536] and lse as the forward left them) and in-projection dgrad:
 *                           dqkv [M, 1536] (the in-projection's weight-gradient operand) and dx [M, 512] = dqkv Wqkv + ds1,
 *                           the gradient w.r.t. the layer input.  dattn is not stored.  `mask` / `partial` sized for
 *                           ib_ffn_chain_attn_workgroups panels (ib_ffn_chain_attn_mask_bytes). */
/* frozen-weight forward (DDIM sampler beyond 8192 rows): ib_ffn_chain_fwd's attention-epilogue form (+ optional QKV tail)
 * with nothing saved for a backward -- only y [M, 512] (and qkv_next [M, 1536]) leave the workgroup */
int ib_ffn_chain_fwd_infer(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                           const float* beta, void* y, const void* attn, const float* bo, const float* gamma1,
                           const float* beta1, const void* packed_next, const float* bqkv_next, void* qkv_next, int64_t M,
                           int64_t d, int64_t ffn, float ln_eps, ib_stream_t stream);
int ib_ffn_chain_attn_workgroups(int64_t M, int64_t d, int64_t ffn, int64_t T);
size_t ib_ffn_chain_attn_mask_bytes(int64_t M, int64_t d, int64_t ffn, int64_t T);
int ib_ffn_chain_fwd_attn(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                          const float* beta, void* f1, void* s2, void* y, float* mean, float* rstd, void* mask,
                          const void* attn, const float* bo, const float* gamma1, const float* beta1, void* s1, void* x1_out,
                          float* mean1, float* rstd1, const void* packed_next, const float* bqkv_next, void* qkv_next,
                          void* attn_next, float* lse_next, int64_t T, int64_t M, int64_t d, int64_t ffn, float ln_eps,
                          ib_stream_t stream);
int ib_ffn_chain_bwd_attn(const void* dy, const void* s2, const float* mean, const float* rstd, const float* gamma,
                          const void* packed, const void* mask, void* ds2, void* dz1, float* partial, const void* s1,
                          const float* mean1, const float* rstd1, const float* gamma1, void* ds1, const void* qkv,
                          const float* lse, void* dqkv, void* dx, int64_t T, int64_t M, int64_t d, int64_t ffn,
                          ib_stream_t stream);

/* ---- diffusion wrapper [BUILD-DEFINED]: DDPM q_sample, DDIM eta=0 update, table gathers ----- */
/* ---- tiny matrix products: C[M,N] (+)= sum_k A(m,k) B(k,n), A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn], each
 * operand with its own storage dtype (IB_F32 / IB_BF16), fp32 accumulation in a fixed order.  The frame-embedding
 * projection of the transformer denoiser and its two gradients (nn.Embedding(window, 30) concatenated on the feature
 * dim, TransformerBaseline.py:41-48,119-126, then the input Linear): [50,30] x [30,512] problems on an unaligned column
 * slice of in_proj.weight.  M*N <= 2^22, K <= 2^16, M*N*K <= 2^28; IB_E_UNSUPPORTED beyond. */
int ib_tiny_matmul(const void* A, int a_dtype, int64_t sam, int64_t sak, const void* B, int b_dtype, int64_t sbk,
                   int64_t sbn, void* C, int c_dtype, int64_t ldc, int accumulate, int64_t M, int64_t N, int64_t K,
                   ib_stream_t stream);

/* out[b, :] = table[idx[b], :]  (timestep-embedding rows; table computed in float64 on the host,
 * cast once).  idx int64. */
int ib_gather_rows(const float* table, const int64_t* idx, void* out, int64_t B, int64_t dim,
                   int64_t table_rows, int dtype_out, ib_stream_t stream);
/* gradient of ib_gather_rows w.r.t. the table (nn.Embedding backward, TransformerBaseline.py:41-48): dtable[r, :] = sum
 * of dout[i, :] over the positions with idx[i] == r, in position order; all fp32. */
int ib_gather_rows_bwd(const float* dout, const int64_t* idx, float* dtable, int64_t n, int64_t dim, int64_t table_rows,
                       ib_stream_t stream);
/* x_t[b,f,:] = sqrt_ab[t[b]] * x0[b,f,:] + sqrt_1mab[t[b]] * eps[b,f,:]; x0 / eps contiguous [B,T,D];
 * x_t is [B*T, D] with leading dimension ld_xt >= D (a 16-byte-aligned row pitch for D = 300). */
int ib_q_sample(const void* x0, const void* eps, const int64_t* t, const float* sqrt_ab,
                const float* sqrt_1mab, void* x_t, int64_t ld_xt, int64_t B, int64_t T, int64_t D,
                int64_t table_rows, int dtype, ib_stream_t stream);
/* x <- coef[s][0] * x + coef[s][1] * eps with s = *step_dev (or `step` if step_dev NULL).
 * Also writes t_out[b] = timesteps[s+1] (the NEXT step's timestep, if t_out given) so a captured
 * single-step graph can be replayed; ib_counter_add advances the counter. */
int ib_ddim_step(void* x, const void* eps, const float* coef, const int64_t* timesteps,
                 int64_t num_steps, int32_t step, const int32_t* step_dev, int64_t* t_out, int64_t B,
                 int64_t n, int dtype, ib_stream_t stream);
/* The diffusion batch made on the device (csrc/noise.hip; replaces host torch.randint / torch.randn + H2D copies in the
 * training loop, cli/train.py -- the reference has no diffusion path).  For window b < B:
 *   x0_out[b, :per] = table[idx[b], :per]           (table NULL: x0 untouched; row_pitch % 8 == 0, row_pitch >= per)
 *   t_out[b]        = floor(word0(b) * num_train_steps / 2^32)                          (t_out NULL: not drawn)
 *   eps_out[b, :]   = N(0,1) by Box-Muller over Philox4x32-10 words                       (eps_out NULL: not drawn)
 * Philox counter = (block, step + *step_dev, stream_id, domain), key = seed; element e of the [B, per] batch uses block
 * e / 4, domain 0; timesteps use block b, domain 1.  Words and timesteps are bit-exact against oracle/ref_cpu.py
 * (philox4x32, draw_timesteps); the normals use hardware log2 / sin / cos (compared at 2e-5 absolute).  step_dev (device
 * int32, may be NULL) lets a replayed hipGraph draw fresh numbers every step.  dtype = storage type of table / x0 / eps. */
int ib_diffusion_draw(const void* table, int64_t table_rows, int64_t row_pitch, const int64_t* idx, void* x0_out,
                      void* eps_out, int64_t* t_out, int64_t B, int64_t per, int32_t num_train_steps, uint64_t seed,
                      int32_t step, const int32_t* step_dev, uint32_t stream_id, int dtype, ib_stream_t stream);
/* out[4 * i .. 4 * i + 3] = Philox4x32-10 words of counter (i, step, stream_id, domain), key = seed, i < blocks: the raw
 * stream behind ib_diffusion_draw (parity tests; bit-exact against oracle/ref_cpu.py::draw_words). */
int ib_philox_words(uint32_t* out, int64_t blocks, uint64_t seed, uint32_t step, uint32_t stream_id, uint32_t domain,
                    ib_stream_t stream);
/* on-device window cache: table = packed fp32 rows [rows, row_elems], one per window:
 * [model input (x_elems, frame-major) | labels key-major: cop, force, torque, wrench], every block zero-padded to a
 * multiple of 4 values (row_elems = pad4(x_elems) + sum pad4(lab_elems[k])); gathers idx[B] rows into the
 * model input (fp32 or bf16) and the four contiguous fp32 label tensors in ONE launch.  Replaces
 * AddBiomechanicsDataset.__getitem__ (:161-285) + collate + the model's torch.concat (FeedForwardRegressionBaseline.py
 * :97-108) for windows that are already packed.  lab_out / lab_elems: host arrays of 4 (values per window of each label
 * tensor, e.g. 6 | 6 | 6 | 12 for 'last_frame'; blocks that are whole 16-byte pieces are copied as such). */
int ib_gather_windows(const float* table, int64_t row_elems, int64_t rows, const int64_t* idx, int64_t B,
                      void* x_out, int64_t x_elems, int dtype_x, float* const* lab_out, const int64_t* lab_elems,
                      ib_stream_t stream);
/* ---- Groundlink (src/models/Groundlink.py:41-48): Conv1d(k, padding=k/2, replicate) over the
 * frames of a window as a GEMM over an explicit im2col.  x: [N*F, C] channels-last rows; col: [N*F, ldcol >= C*k] with
 * col[(n,f)][c*k + j] = x[(n, clamp(f + j - k/2, 0, F-1))][c]  (column order == weight.view(C_out, C_in*k)), columns beyond
 * C*k are zero-filled.  col2im is its transpose (fixed summation order) times the activation derivative of the layer
 * below (act' evaluated on `aux`, that layer's OUTPUT; aux NULL or act NONE = plain). */
int ib_im2col_replicate(const void* x, void* col, int64_t ldcol, int64_t N, int64_t F, int64_t C, int k, int dtype,
                        ib_stream_t stream);
int ib_col2im_replicate(const void* dcol, int64_t ldcol, const void* aux, int act, void* dx, int64_t N, int64_t F,
                        int64_t C, int k, int dtype, ib_stream_t stream);
/* inverted dropout y = x * m / (1 - p), m ~ Bernoulli(1 - p) from a counter-based hash of (seed, step, element): the same
 * call on the gradient reproduces the mask (torch.nn.Dropout, Groundlink.py:54,59; the mask STREAM differs from torch's
 * generator, the distribution does not).  step_dev (device int32, may be NULL -> `step`) makes it graph-replayable. */
int ib_dropout(const void* x, void* y, int64_t n, float p, uint32_t seed, int32_t step, const int32_t* step_dev, int dtype,
               ib_stream_t stream);
int ib_counter_add(int32_t* counter, int32_t delta, ib_stream_t stream);

/* ---- nn.BatchNorm1d over [B, C] rows (the optional layer in front of every Linear of the feedforward model,
 * src/models/FeedForwardRegressionBaseline.py:71-72; flag --batchnorm, src/cli/train.py:47).  torch defaults: eps 1e-5,
 * momentum 0.1, affine, track_running_stats.  training != 0: batch statistics normalise, the running statistics and
 * num_batches_tracked (int64 device scalar, may be NULL) are updated in place (unbiased variance), B >= 2; training == 0:
 * the running statistics normalise.  save_mean / save_rstd [C] receive what the backward needs.  Fixed summation order.
 * Backward: dgamma / dbeta [C] (+)=, dx (may be NULL) = d loss / d x, optionally multiplied by the derivative of the
 * activation below (act_below / aux as in ib_linear_dgrad). */
int ib_batchnorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, void* y, int64_t ldy, float* save_mean,
                     float* save_rstd, int64_t B, int64_t C, float momentum, float eps, int training, int dtype,
                     ib_stream_t stream);
int ib_batchnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* save_mean,
                     const float* save_rstd, void* dx, int64_t lddx, float* dgamma, float* dbeta, int accumulate,
                     int act_below, const void* aux, int64_t ldaux, int64_t B, int64_t C, int training, int dtype,
                     ib_stream_t stream);
/* y[i] *= *scale (fp32 device scalar), in place; y 16-byte aligned.  The loss plugins' autograd bridge: the kernels write
 * d loss / d outputs in the forward launch, `loss.backward()` (src/cli/train.py:281) later hands the upstream gradient of
 * the scalar loss over as a device scalar. */
int ib_scale_by_device_scalar(void* y, const float* scale, int64_t n, int dtype, ib_stream_t stream);
int ib_fill_i64(int64_t* dst, int64_t value, int64_t n, ib_stream_t stream);

/* ---- fused training chain of the token-wise MLP denoiser (BASELINE.json configs[1]; bf16 only).
 * Replaces, in ONE launch, what the per-op plan issues for SURVEY.md §8a rows "forward" + "loss" + the dgrad half of
 * "backward" (the torch ops behind cli/train.py:240-281 for this model family): q_sample, every block
 * (Linear + time embedding + SiLU + LayerNorm), the head, MSE loss with dL/dpred, and the backward chain through the
 * head, LayerNorm/SiLU and the block weights.  Token rows never mix, so a workgroup owns <= 64 tokens and keeps their
 * activations in LDS.  The weight gradients stay ib_linear_wgrad launches over the operands this leaves in HBM.
 *   hidden width H in {128, 256, 512} for every block, L <= 4 blocks, D % 4 == 0 and D <= 512 at H = 512 (narrower
 *   sets for H = 128 / 256: ib_mlp_chain_supported says).
 *   packed: the bf16 weights in MFMA-fragment order (ib_mlp_chain_pack; re-pack after every optimizer step).
 *   bias[L+1], gamma[L], beta[L], u[L], h[L], dz[L]: HOST arrays of device pointers.
 *   u_i = W_i h_{i-1} + b_i + e[window] (bf16), h_i = LN(silu(u_i)), dz_i = dL/du_i, all [M, H] contiguous.
 *   partial: [ib_mlp_chain_workgroups(M), ld_part >= ib_mlp_chain_partial_width()] fp32 per-workgroup column sums:
 *     block i at columns 3iH: dgamma_i | dbeta_i | dbias_i (H each); head-bias sums at 3LH; the squared-error sum
 *     at width - 4.  ib_colsum_segments() turns them into the parameter gradients and the loss in one launch.
 *   de_lp: optional bf16 [B, >= L*H]: when each workgroup's panel is exactly one window (rows_per_workgroup == T)
 *     the dbias sums are also that window's time-embedding gradient row; NULL otherwise (use ib_segment_colsum). */
int ib_mlp_chain_supported(int64_t D, int64_t H, int L);
size_t ib_mlp_chain_packed_elems(int64_t D, int64_t H, int L);
int ib_mlp_chain_workgroups(int64_t M, int* rows_per_workgroup);
int ib_mlp_chain_pack(const void* const* w, const int64_t* ldw, void* packed, int64_t D, int64_t H, int L,
                      ib_stream_t stream);        /* w[L+1]: blocks.i.linear.weight (bf16), then head.weight */
int64_t ib_mlp_chain_partial_width(int64_t D, int64_t H, int L);
int ib_mlp_chain_train(const void* x0, const void* eps, const int64_t* t, const float* sqrt_ab,
                       const float* sqrt_1mab, int64_t table_rows, const void* e, int64_t ld_e,
                       const void* packed, const float* const* bias, const float* const* gamma,
                       const float* const* beta, void* xt, int64_t ld_xt, void* const* u, void* const* h,
                       void* const* dz, void* dpred, int64_t ld_dpred, float* partial, int64_t ld_part,
                       void* de_lp, int64_t ld_de, const void* const* in_slots, int64_t M, int64_t T, int64_t D,
                       int64_t H, int L, float ln_eps, ib_stream_t stream);
/* in_slots (chain_train / chain_prep): optional DEVICE array {x0, eps, t} the kernels dereference at their start instead
 * of the pointer arguments -- a captured graph then consumes each step's batch where it lies (ib_set_ptrs before the
 * graph launch) instead of through a staging copy. */
int ib_set_ptrs(void* slots, int n, const void* const* ptrs, ib_stream_t stream);   /* n <= 4, ptrs = host array */
/* dst_s[c] (+)= scale_s * sum_r part[r][col0_s + c], c < ncols_s, for nseg <= 24 segments in ONE launch (fixed
 * summation order); dst2 (array or NULL; entries may be NULL) receives a second copy.  Host arrays. */
int ib_colsum_segments(const float* part, int64_t ld, int64_t rows, int nseg, const int32_t* col0,
                       const int32_t* ncols, float* const* dst, float* const* dst2, const float* scale,
                       int accumulate, ib_stream_t stream);
/* fused forward of the time-embedding MLP (models: time_mlp.0 / time_mlp.2; bf16): e = W2 silu(W1 sinus(t) + b1) + b2.
 * table: fp32 [table_rows, temb] sinusoid rows; w1 [hidden, temb], w2 [out, hidden] row-major bf16 (as stored).
 * Also writes what the backward needs: s [B, temb] (gathered rows), zu [B, hidden] (pre-activation), u = silu(zu). */
int ib_time_mlp_fwd_supported(int64_t temb, int64_t hidden, int64_t out);
int ib_time_mlp_fwd(const float* table, int64_t table_rows, const int64_t* t, const void* w1, int64_t ldw1,
                    const float* b1, const void* w2, int64_t ldw2, const float* b2, void* s, void* zu, void* u,
                    void* e, int64_t ld_e, int64_t B, int64_t temb, int64_t hidden, int64_t out,
                    ib_stream_t stream);
/* Backward of the time-embedding MLP's hidden layer in ONE launch of independent workgroups (bf16):
 *   dzu = (de w2) * silu'(zu)   [B, hidden];   dW1 = dzu^T s   [hidden, temb];   db1 = column sums of dzu (as stored in bf16)
 * de [B, out] = d loss / d e (the chain kernel's per-window sums), w2 [out, hidden], zu / s: what ib_time_mlp_fwd saved.
 * The results are left as ib_time_mlp_bwd_slab_count(B) fp32 partial slabs (one per 64 windows, summed in slab order by
 * ib_optim_step_sources / ib_step_reduce like every split weight gradient): dw1_slabs [slabs][hidden][temb],
 * db1_slabs [slabs][hidden].  Replaces ib_linear_dgrad_skinny + ib_linear_wgrad_slabs of time_mlp.0 on a forked stream. */
int ib_time_mlp_bwd_supported(int64_t temb, int64_t hidden, int64_t out);
int ib_time_mlp_bwd_slab_count(int64_t B);
int ib_time_mlp_bwd(const void* de, int64_t ld_de, const void* w2, int64_t ldw2, const void* zu, int64_t ldzu,
                    const void* s, int64_t lds, float* dw1_slabs, float* db1_slabs, int64_t B, int64_t temb,
                    int64_t hidden, int64_t out, ib_stream_t stream);
/* ib_linear_wgrad_slabs_multi + ib_time_mlp_bwd as ONE launch: the time-MLP backward's workgroups ride in the grouped
 * weight-gradient launch (the MLP denoiser's group leaves 52 of 256 CUs idle).  IB_E_UNSUPPORTED = nothing launched. */
int ib_linear_wgrad_slabs_multi_tb(int n, const void* const* dz, const int64_t* lddz, const void* const* x,
                                   const int64_t* ldx, void* const* workspace, const size_t* workspace_bytes,
                                   int32_t* nslab_out, const int64_t* M, const int64_t* N, const int64_t* K, int dtype,
                                   const void* de, int64_t ld_de, const void* w2, int64_t ldw2, const void* zu, int64_t ldzu,
                                   const void* s, int64_t lds, float* dw1_slabs, float* db1_slabs, int64_t B, int64_t temb,
                                   int64_t hidden, int64_t out, ib_stream_t stream);
/* ib_time_mlp_fwd + ib_mlp_chain_pack as ONE launch (both are independent and tiny; saves a kernel boundary) */
int ib_mlp_chain_prep(const float* table, int64_t table_rows, const int64_t* t, const void* w1, int64_t ldw1,
                      const float* b1, const void* w2, int64_t ldw2, const float* b2, void* s, void* zu, void* u,
                      void* e, int64_t ld_e, int64_t B, int64_t temb, int64_t hidden, int64_t out,
                      const void* const* w, const int64_t* ldw, void* packed, int64_t D, int64_t H, int L,
                      const void* const* in_slots, ib_stream_t stream);
/* ib_slab_reduce_multi + ib_colsum_segments in ONE launch (the reductions that finish a training step's gradients) */
int ib_step_reduce(int n, const void* const* slabs, const int32_t* nslab, float* const* dw, const int64_t* lddw,
                   const int32_t* N, const int32_t* K, const float* part, int64_t ld, int64_t rows, int nseg,
                   const int32_t* col0, const int32_t* ncols, float* const* dst, float* const* dst2,
                   const float* scale, int accumulate, ib_stream_t stream);
/* the same with a partial matrix per segment (each left by a different launch): part[j] [rows[j]][ld[j]] fp32 */
int ib_step_reduce_parts(int n, const void* const* slabs, const int32_t* nslab, float* const* dw, const int64_t* lddw,
                         const int32_t* N, const int32_t* K, int nseg, const float* const* part, const int64_t* ld,
                         const int32_t* rows, const int32_t* col0, const int32_t* ncols, float* const* dst,
                         const float* scale, int accumulate, ib_stream_t stream);
int ib_debug_stamp(void* slot, ib_stream_t stream);   /* timing-only: *slot = 100 MHz wall clock when the stream gets here */
int ib_debug_set_gemm_prof(void* stamps);    /* timing-only: [workgroups][8] stamps of the ring GEMM kernel, NULL = off */
int ib_debug_set_chain_prof(void* stamps);   /* timing-only: [workgroups][16] int64 wall-clock stamps, NULL = off */
int ib_debug_set_ffn_prof(void* stamps);     /* timing-only: [workgroups][64] stamps of ib_ffn_chain_fwd (measurement builds) */
int ib_sum_partials(const float* partial, int64_t parts, float scale, float* out, ib_stream_t stream);

/* ---- hipGraph capture of a launch sequence (the captured denoise / train step) */
int ib_graph_begin(ib_stream_t stream);
int ib_graph_end(ib_stream_t stream, void** graph_exec_out);
int ib_graph_launch(void* graph_exec, ib_stream_t stream);
int ib_graph_destroy(void* graph_exec);

/* ---- streams of the library's own (non-blocking, current device).  The reference runs everything on the default stream
 * (SURVEY.md 8b "Threading / processes"); the build's side branches / capture stream / trainer stream must not come from
 * torch's round-robin pool of 32 streams, which c10d's communication stream shares (engine.py, plans.Branch) */
int ib_stream_create(void** stream_out);
int ib_stream_destroy(void* stream);

/* ---- HIP-event timing on the launch stream (bench.py's roofline leg) ------------------------ */
int ib_event_create(void** ev_out);
int ib_event_record(void* ev, ib_stream_t stream);
int ib_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */
int ib_event_destroy(void* ev);

/* ---- device self-tests of instruction-layout assumptions (tests only) ----------------------- */
/* timing-only: bit mask of GEMM phases to skip (results are WRONG when non-zero; tools/kbench.py only) */
int ib_debug_set_ablate(int mask);
/* Which kernel family the LAST dispatching entry point of this thread's process launched (tests/test_dispatch_gpu.py holds
 * every BASELINE shape to its family, so that a threshold edit cannot silently move a benchmarked shape to another kernel).
 * Not part of the reference-facing surface; a plain global, read (and cleared) right after the call it describes. */
enum {
  IB_PATH_NONE = 0,
  IB_PATH_NT = 1,           /* gemm_nt.hip: 256 x 128 LDS-DMA kernel (forward / dgrad with a transposed weight) */
  IB_PATH_TN = 2,           /* gemm_tn.hip: 256 x 128 weight-gradient kernel, grouped */
  IB_PATH_RING = 3,         /* gemm.hip: 128 x 128 LDS-DMA ring kernel */
  IB_PATH_GENERIC = 4,      /* gemm.hip: register-staged kernel (fp32 exact MFMA, or unaligned / ragged bf16 operands) */
  IB_PATH_SMALLM = 5,       /* gemm.hip: 64 x 16 tiles for batches of a few hundred rows (forward / dgrad) */
  IB_PATH_SKINNY = 6,       /* gemm.hip: few-row dgrad (one workgroup per 16 output columns) */
  IB_PATH_WGRAD_SMALL = 7,  /* gemm.hip: direct weight gradient of a short reduction */
  IB_PATH_RING_MULTI = 8,   /* gemm.hip: grouped ring weight-gradient launch */
  IB_PATH_LINLN = 9,        /* gemm.hip: K-split Linear + residual + LayerNorm (sampler) */
  IB_PATH_CHAIN2 = 10,      /* chain.hip: fused MLP-denoiser chain, row-wise epilogues */
  IB_PATH_CHAIN1 = 11,      /* chain.hip: the round-2 chain kernel (IB_CHAIN_V1=1) */
  IB_PATH_TN256 = 12,       /* gemm_tn256.hip: 256 x 256 weight-gradient kernel, grouped, one split count per group */
  IB_PATH_NT_SPLITK = 13,   /* gemm_nt.hip in split-K form (fp32 slabs) under the sampler's Linear + LayerNorm */
  IB_PATH_FFN_CHAIN = 14,   /* ffn_chain.hip: fused feed-forward sublayer (Linear + ReLU + Linear + residual + LayerNorm) */
  IB_PATH_LINLN_PANEL = 15, /* linln_panel.hip: Linear + residual + LayerNorm over row panels, one launch (sampler) */
  IB_PATH_FFN_INFER = 16,   /* linln_panel.hip: feed-forward sublayer, panels shared by their hidden chunks (sampler) */
  IB_PATH_LIN_PANEL = 17    /* linln_panel.hip: Linear over (row panel, 512-column chunk) workgroups (sampler in-projection) */
};
int ib_debug_last_path(void);
int ib_selftest_tr16(const void* in_bf16_64x16, void* out_bf16_64x4, ib_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IB_HIP_H */
