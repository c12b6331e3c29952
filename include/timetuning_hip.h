/*
 * timetuning_hip.h - C ABI of libtimetuning_hip.so (gfx950 / MI355X).
 *
 * The reference (SMSD75/Timetuning) is pure Python on top of ATen; it has no FFI of its own.
 * The entry points below are the op sites of its training hot path (SURVEY.md section 2.4,
 * k1-k19) restated as a C ABI: what a maintainer would bind with ctypes in place of the
 * ATen calls.  Each declaration cites the reference lines it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (row-major, contiguous unless a leading
 *     dimension is passed) unless the parameter says otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls enqueue work
 *     and return, they never synchronise, allocate or free (ABI 7: also true of the K-split
 *     workspace of the persistent GEMMs, which is the caller's - tt_linear_ksplit_workspace_*);
 *     every call can be captured into a hipGraph;
 *   - return value: 0 on success, a negative TT_E* code otherwise; tt_last_error() returns a
 *     thread-local message for the last failing call;
 *   - workspaces are caller-owned; the matching *_workspace_bytes() gives the size.
 */
#ifndef TIMETUNING_HIP_H
#define TIMETUNING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_OK 0
#define TT_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define TT_ELAUNCH (-2)  /* HIP launch error */
#define TT_EUNSUPPORTED (-3)

typedef void* tt_stream_t;

const char* tt_last_error(void);
int tt_abi_version(void);   /* 8 = this header (7: before `precision` became an argument - tt_set_gemm_precision, a process-wide switch, is gone; 6: before the caller-owned K-split workspace and the range flag; 5: before the transpose-free weight gradient, the batched operand refresh and the distributed Sinkhorn steps; 4: before the fp16-pair entry points; 3: before tt_vit_params.patch_wp; 2: before the coarse entry points) */
/* Tuning knobs of the dispatchers (TT_PLANES_VARIANT, TT_P8_ORDER, TT_P8_NO_HALF, TT_P8_CLOCK_PRINT, TT_Q8_ORDER, TT_PAIRS_NO8,
 * TT_PAIRS8_NO_KEPT) are read ONCE from the environment; this setter changes one afterwards - for the A/B tools and tests only. */
int tt_set_tuning_knob(const char* name, int value);
/* Fills name (<= cap bytes) with the gcnArchName of the current device; returns CU count or <0. */
int tt_device_info(char* name, int cap);

/* ---- ABI 7: the K-split workspace of the persistent GEMM kernels (gemm_pairs8.hip / gemm_planes8.hip behind tt_linear_fwd_pairs,
 * tt_linear_fwd_planes, tt_linear_bwd_data_pairs, tt_linear_bwd_data_planes).  The output tiles beyond the last whole round over the CUs
 * are split along K over several workgroups, which exchange fp32 partials and (tile, wave) arrival counters through this buffer.
 *   tt_linear_ksplit_workspace_bytes   its size for the current device (counters + partials; 64 MB + 8 KB on MI355X);
 *   tt_linear_ksplit_workspace_init    zeroes the counter block (a hipMemsetAsync on `stream`): ONCE after allocation - every launch
 *                                      leaves the counters zero again - and again after a launch that was aborted.
 * One buffer serves any sequence of launches on ONE stream (launches on different streams may overlap: give each stream its own).
 * workspace == NULL (or too small): such a call never splits K - the left-over tiles are cut into half tiles or dealt round-robin;
 * same arithmetic per output element, a different summation order across the K-tiles of those tiles (results may differ in the last bit
 * from a call with a workspace). */
size_t tt_linear_ksplit_workspace_bytes(void);
int tt_linear_ksplit_workspace_init(void* workspace, size_t workspace_bytes, tt_stream_t stream);

/* ---- k4,k6,k7,k8,k9: nn.Linear forward  (dino_vision_transformer.py:94-103,115-117,122,130;
 *      models.py:915-926,1075-1077)
 *   y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual)
 *   act: 0 none, 1 exact-erf GELU.  pre_act (optional, [M,N]) receives x@w^T+bias before the
 *   activation (saved for backward).  residual (optional, [M,N]) is added after the activation;
 *   it may alias y. */
int tt_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                  float* pre_act, int M, int N, int K, int act, int precision, tt_stream_t stream);

/* ---- k16: nn.Linear backward (autograd of the sites above)
 *   dx[M,K] = dy[M,N] @ w[N,K]            (* gelu'(gelu_pre[M,K]) if gelu_pre != NULL)
 *   dw[N,K] = dy[M,N]^T @ x[M,K],  db[N] = column sums of dy (db may be NULL)
 *   workspace for tt_linear_bwd_weight: tt_colsum_workspace_bytes(M, N) when db != NULL. */
int tt_linear_bwd_data(const float* dy, const float* w, const float* gelu_pre, float* dx, int M, int N, int K,
                       tt_stream_t stream);
int tt_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N, int K,
                         void* workspace, size_t workspace_bytes, tt_stream_t stream);
size_t tt_linear_bwd_weight_workspace_bytes(int M, int N, int K); /* split-K partials and/or the column-sum scratch */
/* Both products of one nn.Linear in one call (they share dy and do not depend on each other): dx as tt_linear_bwd_data, dw / db as
 * tt_linear_bwd_weight, bit for bit.  Where both lean kernels apply and the weight gradient is split along its reduction, ONE launch carries
 * the dgrad tiles and the weight-gradient slices (neither fills the chip on the target frames alone), else the two are launched one after
 * the other.  x is the layer's input [M, Kx] with Kx = K; workspace: tt_linear_bwd_weight_workspace_bytes(M, N, K). */
int tt_linear_bwd(const float* dy, const float* w, const float* x, const float* gelu_pre, float* dx, float* dw, float* db, int M, int N, int K,
                  void* workspace, size_t workspace_bytes, tt_stream_t stream);
size_t tt_colsum_workspace_bytes(int M, int N);
/* out[N] = sum over rows of a[M,N] (deterministic two-stage reduction). */
int tt_colsum(const float* a, float* out, int M, int N, void* workspace, size_t workspace_bytes, tt_stream_t stream);

/* ---- generic fp32 MFMA GEMM used by the sites above and by k11/k14:
 *   C[M,N] = alpha * op(A)[M,K] @ op(B)[K,N]
 *   a_mmajor = 0: A stored [M][lda] (k contiguous); 1: stored [K][lda] (m contiguous)
 *   b_nmajor = 0: B stored [N][ldb] (k contiguous, the nn.Linear weight layout); 1: stored [K][ldb]
 *   batch > 1 runs `batch` independent problems at the given element strides. */
int tt_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int a_mmajor, int b_nmajor, float alpha, int batch, long long strideA, long long strideB,
                long long strideC, tt_stream_t stream);

/* Which tile shape the launcher picks for an M x N (x batch) product: 0 = 128x128, 1 = 64x128, 2 = 128x64,
 * 3 = 64x64 (block tile; 4 waves each).  Exposed so that profilers can attribute launches to instantiations. */
int tt_gemm_tile_choice(int M, int N, int batch);
/* Which kernel an fp32 tt_linear_fwd of this shape runs: bits 0-1 tile (0 128x128, 1 64x128, 2 128x64, 3 64x64), bit 8 set = the
 * lean whole-tile instance (gemm_nt_fast_kernel), clear = the general kernel.  For profilers' labels only. */
int tt_linear_fwd_route(int M, int N, int K);
/* Which kernel a tt_linear_fwd_planes call with these arguments runs: 8 = the persistent 8-phase kernel (gemm_planes8_kernel: whole
 * 256-wide column tiles, a grid that fills the chip, one of its compiled epilogues), 0 = gemm_planes_kernel.  For profilers' labels only. */
int tt_linear_fwd_planes_route(int planes, int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int y_nplanes,
                               int has_pre_out);

/* ---- k1b: interpolate_pos_encoding for inputs whose token grid differs from the stored one
 *      (dino_vision_transformer.py:214-234): bicubic resampling of the patch position table, as
 *      nn.functional.interpolate(..., scale_factor=(scale_h, scale_w), mode="bicubic") computes it (align_corners False,
 *      coordinate scale 1/scale_factor, A = -0.75, border-clamped taps); the class row is copied.
 *   pos [1 + g*g, D] -> out [1 + gh*gw, D].  The reference passes scale = (rows + 0.1) / g, (cols + 0.1) / g. */
int tt_pos_embed_interpolate(const float* pos, float* out, int g, int gh, int gw, int D, float scale_h, float scale_w,
                             tt_stream_t stream);

/* `precision`: the arithmetic of the fp32-OPERAND forward products - an ARGUMENT of every entry point that has one (ABI 8; until ABI 7 a
 * process-wide switch, tt_set_gemm_precision: hidden state this interface promises not to have).  Taken by tt_linear_fwd,
 * tt_label_propagate[_maps], tt_mlp_head_forward, tt_scores_sinkhorn and, as tt_vit_params.precision, tt_vit_forward (planes == 0):
 *   TT_PRECISION_F32    0  f32 MFMA (exact fmaf chain)
 *   TT_PRECISION_BF16X3 1  operands split into bf16 hi + lo while staged, three bf16 MFMAs per product term (~2^-16 relative per product)
 *   TT_PRECISION_BF16   2  operands rounded to bf16 (BASELINE config C4's "MFMA bf16 path"; does not meet the 1e-3 fp32 contract);
 *                          tt_label_propagate[_maps] then computes its cosine similarities on bf16 MFMA too (what torch.autocast makes of
 *                          them); 0 and 1 leave them exact
 * Inputs / outputs stay fp32 in memory in every case.  The fp32-accurate split modes are not a `precision` but an operand FORMAT with
 * entry points of its own (fp16 pairs: tt_*_pairs*; three bf16 planes: tt_*_planes*).  tt_linear_bwd* always run in f32 (the bf16 path's
 * backward products have their own entry points: tt_linear_bwd_data_planes, tt_linear_bwd_weight_planes, tt_attention_bwd_bf16). */
#define TT_PRECISION_F32 0
#define TT_PRECISION_BF16X3 1
#define TT_PRECISION_BF16 2

/* ---- k1,k2: PatchEmbed conv (kernel = stride = P) + cls token + pos-embed
 *      (dino_vision_transformer.py:166-171, 236-247)
 *   img [F_src,C,H,W]; frame_map (optional int32[F]): output frame f reads img[frame_map[f]];
 *   w [D, C*P*P]; bias [D]; cls [D]; pos [(n+1), D]; tokens [F, n+1, D], n = (H/P)*(W/P). */
int tt_patch_embed_fwd(const float* img, const int32_t* frame_map, const float* w, const float* bias,
                       const float* cls, const float* pos, float* tokens, int F, int C, int H, int W, int P, int D,
                       tt_stream_t stream);
/* The same on bf16 operands - BASELINE C4's bf16 path, what torch.autocast makes of the conv (dino_vision_transformer.py:166-171):
 * w_planes = tt_split_planes(w, 1 plane) [D, C*P*P] bf16, the patches rounded to bf16 on the way into an im2col buffer, fp32
 * accumulation, fp32 tokens.  One row pass + ONE tt_linear_fwd_planes over all F (n + 1) token rows (the class-token rows are zero
 * rows that meet cls + pos[0] in the residual).  Needs P % 4 == 0, W % 4 == 0, C P P % 64 == 0, D % 64 == 0; workspace of
 * tt_patch_embed_planes_workspace_bytes = F (n + 1) C P P bf16. */
size_t tt_patch_embed_planes_workspace_bytes(int F, int C, int H, int W, int P);
int tt_patch_embed_fwd_planes(const float* img, const int32_t* frame_map, const void* w_planes, const float* bias, const float* cls,
                              const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, void* workspace,
                              size_t workspace_bytes, tt_stream_t stream);

/* ---- k3: LayerNorm over the last dim (dino_vision_transformer.py:139,143,196; eps = 1e-6)
 *   mean/rstd (optional, [rows]) are saved for backward.  `rows` counts OUTPUT rows.
 *   skip_group = 0: x is [rows, D].  skip_group = N (tokens per frame): x is [F, N, D] and the output
 *   [F*(N-1), D] drops token 0 of every frame - the final norm followed by `[:, 1:]` (models.py:966-967). */
int tt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                     int rows, int D, float eps, int skip_group, tt_stream_t stream);
/*   dx has x's layout; dgamma/dbeta [D] optional (NULL for frozen norms).  add_to_dx != 0 accumulates
 *   into dx (residual branch).  With skip_group the rows of dx that belong to token 0 are not touched.
 *   workspace: tt_layernorm_bwd_workspace_bytes(rows, D). */
/* amax_out (ABI 7, here and on tt_attention_bwd / tt_l2norm_bwd / tt_linear_bwd_data_pairs): an "amax slot" - tt_amax_slot_bytes()
 * bytes of device memory (16 floats 256 bytes apart: the waves of a kernel spread their atomics over them, thousands on one address
 * serialise) that the caller ZEROED - or NULL.  The kernel raises the slot to max |.| of the gradient it writes (relaxed atomic max on
 * the non-negative floats' bits: deterministic).  That gradient is the dy of the next Linear's backward: tt_split_pairs_dual_parts takes
 * the slot as amax_in and needs no max pass of its own for the power-of-two scale of the split (measured: C2 -0.8 %, C1 -1 %). */
size_t tt_amax_slot_bytes(void);
int tt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                     float* dx, float* dgamma, float* dbeta, int rows, int D, int add_to_dx, int skip_group,
                     void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream);
size_t tt_layernorm_bwd_workspace_bytes(int rows, int D);

/* ---- k5 (+k10): multi-head self-attention core (dino_vision_transformer.py:122-129)
 *   qkv [F, N, 3*H*hd] as written by the qkv Linear (q | k | v, each head-major);
 *   out [F, N, H*hd] = softmax(q k^T * scale) v ; lse [F,H,N] optional (saved for backward);
 *   probs [F,H,N,N] optional (get_last_selfattention, :256-263).  hd must be 64, N <= 256. */
int tt_attention_fwd(const float* qkv, float* out, float* lse, float* probs, int F, int N, int H, int hd,
                     float scale, tt_stream_t stream);
/*   dqkv [F,N,3*H*hd] from dout [F,N,H*hd]; workspace: tt_attention_bwd_workspace_bytes. */
int tt_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F,
                     int N, int H, int hd, float scale, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream);
size_t tt_attention_bwd_workspace_bytes(int F, int N, int H, int hd);
/* The same backward on bf16 MATRIX operands (BASELINE C4's bf16 path; what torch.autocast makes of the backward of q k^T and attn v,
 * dino_vision_transformer.py:125-129): q, k, v, dout, P and dS are rounded to bf16 where they enter a product, products on
 * v_mfma_f32_16x16x16_bf16 with fp32 accumulation; the softmax statistics (lse), delta = <dout, out>, P and dS themselves are fp32.
 * Inputs and outputs stay fp32 in memory; same arguments and workspace as tt_attention_bwd. */
int tt_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                          float scale, void* workspace, size_t workspace_bytes, tt_stream_t stream);
/* The same backward with its matrix products on fp16-PAIR operands (round 6: the "f16x3" mode's attention backward; fp32-class like
 * tt_attention_bwd: q (scaled), k, v, dout, P and dS are split into (hi, lo) where they enter a product, three fp16 MFMAs per product term,
 * two fp32 accumulators; lse, delta, P and dS themselves are fp32).  dout is a gradient: it is split as dout times the power of two S that
 * brings max |dout| into [2^13, 2^14), dS = P (dP - delta) as dS S 2^-12 (exact; the outputs are divided by the scales again).  dout_amax:
 * the amax slot the kernel that wrote dout raised (amax_out of tt_linear_bwd_data_pairs ...), or NULL - the call then measures the maximum
 * itself (one small launch).  range_flag: the pair entry points' range flag (below) - raised when a hi half of q, k, v, the scaled dout
 * or the scaled dS is not finite (dS: |v| beyond ~128).
 * Inputs and outputs fp32 in memory; workspace: tt_attention_bwd_pairs_workspace_bytes; amax_out as tt_attention_bwd. */
int tt_attention_bwd_pairs(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                           float scale, const float* dout_amax, void* workspace, size_t workspace_bytes, int* range_flag, float* amax_out,
                           tt_stream_t stream);
size_t tt_attention_bwd_pairs_workspace_bytes(int F, int N, int H, int hd);

/* ---- k11: F.normalize(x, dim=-1) (time_tuning.py:136; mask_propagation.py:418-419)
 *   xn[rows,D] = x / max(||x||, 1e-12); inv_norm[rows] optional.  x rows may be strided (ldx). */
int tt_l2norm_fwd(const float* x, int ldx, float* xn, float* inv_norm, int rows, int D, tt_stream_t stream);
/*   dx = (dxn - xn * <xn, dxn>) * inv_norm */
int tt_l2norm_bwd(const float* dxn, const float* xn, const float* inv_norm, float* dx, int rows, int D, float* amax_out,
                  tt_stream_t stream);
/* ---- k18: in-place row L2 normalisation of the prototypes (time_tuning.py:124-128). */
int tt_normalize_rows_inplace(float* w, int rows, int D, tt_stream_t stream);

/* ---- k12: Sinkhorn-Knopp (time_tuning.py:157-168 + my_utils.py:246-274)
 *   scores [B_total, K] (all columns of the global problem: local batch, queue rows, and for
 *   world_size > 1 every rank's rows after an all-gather); q_out [rows_out, K] receives the
 *   assignment of rows [row0, row0+rows_out).  Implements Q=exp(scores/eps)^T, Q/=sum,
 *   `iters` x (row-normalise to 1/K, column-normalise to 1/B_total), final column normalise,
 *   in scaling-vector form.  workspace: tt_sinkhorn_workspace_bytes(B_total, K). */
int tt_sinkhorn(const float* scores, float* q_out, int B_total, int K, int row0, int rows_out, float eps,
                int iters, void* workspace, size_t workspace_bytes, tt_stream_t stream);
size_t tt_sinkhorn_workspace_bytes(int B_total, int K);
/*   The reference's own DISTRIBUTED form (my_utils.py:250-272: the columns stay on their rank, the K row sums are all-reduced once per
 *   iteration), one rank's share in steps - the caller all-reduces u[K] (sum) between them and passes the SAME workspace to all three:
 *     tt_sinkhorn_local_begin   scores [B_loc, K] -> E = exp(scores / eps) in the workspace, u_out[k] = the local row sums
 *     tt_sinkhorn_local_step    u_in = the all-reduced row sums: row step, column step with c = 1 / B_total, u_out = the next local row sums
 *     tt_sinkhorn_local_end     the last row step (u_in NULL: none - zero iterations) + final column normalisation -> q_out [rows_out, K]
 *   iters iterations = begin, (all-reduce, step) x (iters - 1), all-reduce, end.  Equal to tt_sinkhorn on the gathered rows up to fp32
 *   rounding (the sums fold in a different order). */
size_t tt_sinkhorn_local_workspace_bytes(int B_loc, int K);
int tt_sinkhorn_local_begin(const float* scores, float* u_out, int B_loc, int K, float eps, void* workspace, size_t workspace_bytes,
                            tt_stream_t stream);
int tt_sinkhorn_local_step(const float* u_in, float* u_out, int B_loc, int B_total, int K, void* workspace, size_t workspace_bytes,
                           tt_stream_t stream);
int tt_sinkhorn_local_end(const float* u_in, float* q_out, int B_loc, int rows_out, int K, void* workspace, size_t workspace_bytes,
                          tt_stream_t stream);
/*   The reference-signature entry, my_utils.sinkhorn(Q, nmb_iters, world_size) (my_utils.py:246): takes the POSITIVE matrix
 *   exp(scores / eps) itself (no log / exp round trip) - as Q [K, B_total] (transposed = 0: the reference's layout) or as
 *   Q^T [B_total, K] (transposed = 1: what an all-gather of the ranks' columns yields; read in place).  Same iterations,
 *   same workspace. */
int tt_sinkhorn_from_q(const float* Q, int transposed, float* q_out, int B_total, int K, int row0, int rows_out, int iters,
                       void* workspace, size_t workspace_bytes, tt_stream_t stream);

/* ---- bf16-plane operands: BASELINE config C4's "MFMA bf16 path" (planes = 1) and the fp32-accurate split mode (planes = 3)
 *   of the forward nn.Linear sites (dino_vision_transformer.py:94-103,115-130).  A tensor "in P planes" is P bf16 arrays of
 *   the tensor's shape, `plane_stride` ELEMENTS apart, with value = plane0 + plane1 + plane2 (plane0 = bf16(x), plane1 =
 *   bf16(x - plane0), plane2 = bf16(x - plane0 - plane1): 8 / 16 / 24 significant bits).  Producers write planes, consumers
 *   read planes: nothing is converted on the GEMM's own path.
 *   tt_split_planes          fp32 [n] -> planes (weights; activations produced by fp32 kernels).  n % 8 == 0.
 *   tt_layernorm_fwd_planes  tt_layernorm_fwd with the result in planes (optional mean / rstd as there).
 *   tt_linear_fwd_planes     y = act(x @ w^T + bias) (+ residual): x [M,K] and w [N,K] in `planes` planes each; products
 *                            x_i w_j with i + j <= planes + 1 (1 / 3 / 6 bf16 MFMAs per term), fp32 accumulate.  Outputs, any of:
 *                            y fp32 [M,N], pre_out fp32 (pre-activation), y_planes (y_nplanes planes of y).  residual fp32 may
 *                            alias y.  N % 64 == 0, K % 64 == 0; any M.
 *   tt_attention_fwd_bf16    softmax(q k^T * scale) v on bf16 qkv [F,N,3*H*64] -> bf16 out [F,N,H*64]; fp32 scores and
 *                            accumulation, N <= 256 (dino_vision_transformer.py:122-129). */
int tt_split_planes(const float* src, void* dst_planes, long long plane_stride, int planes, long long n, tt_stream_t stream);
int tt_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, void* y_planes, long long plane_stride,
                            int planes, float* mean, float* rstd, int rows, int D, float eps, int skip_group,
                            tt_stream_t stream);
int tt_linear_fwd_planes(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes,
                         const float* bias, const float* residual, float* y, float* pre_out, void* y_planes,
                         long long y_plane_stride, int y_nplanes, int M, int N, int K, int act, void* workspace, size_t workspace_bytes,
                         tt_stream_t stream);   /* workspace: tt_linear_ksplit_workspace_bytes() or NULL (ABI 7) */
int tt_attention_fwd_bf16(const void* qkv, void* out, int F, int N, int H, int head_dim, float scale, tt_stream_t stream);

/* ---- fp16-PAIR operands: the fp32-accurate split mode "f16x3" (round 4) of the forward nn.Linear sites
 *   (dino_vision_transformer.py:94-103,115-130; models.py:915-926).  An fp32 value x is held as hi = fp16(x) and lo = fp16((x - hi) * 2^11):
 *   x = hi + lo * 2^-11 to <= 2^-23 relative.  A tensor "in pairs" has its elements in groups of 32 consecutive ones, each group stored
 *   as [hi x 32][lo x 32] fp16 - 4 bytes per element, rows as long as in fp32.  A product term costs three fp16 MFMAs (hi hi into one
 *   fp32 accumulator, hi lo + lo hi into a second one folded in with the exact 2^-11); the dropped lo lo term is <= 2^-22 relative.
 *   Operands must lie in fp16's range (|x| <= 65504: beyond it the result is inf / NaN, as an fp16 autocast's would be).
 *   The RANGE FLAG (ABI 7): every entry point that PRODUCES pairs from fp32 values - tt_split_pairs, tt_layernorm_fwd_pairs,
 *   tt_linear_fwd_pairs (y_pairs), tt_split_pairs_dual(_multi), tt_patch_embed_fwd_pairs, tt_vit_params.range_flag - takes `range_flag`, a
 *   DEVICE int (or NULL): the kernel stores 1 to it when a value it splits is beyond fp16's range or not finite (hi = inf / NaN) and never
 *   clears it.  The caller zeroes it, reads it at its own synchronisation points and decides (the Python host raises PairRangeError and names
 *   --precision f32; tests/test_hip_pairs.py).  tt_attention_fwd_pairs needs none: its outputs are convex combinations of v.
 *   tt_split_pairs          fp32 [n] -> pairs, n % 32 == 0 (weights; activations produced by fp32 kernels).  tt_join_pairs: back.
 *   tt_layernorm_fwd_pairs  tt_layernorm_fwd with the result in pairs [rows][2 D] (D % 32 == 0; optional mean / rstd as there).
 *   tt_linear_fwd_pairs     y = act(x @ w^T + bias) (+ residual): x [M,K], w [N,K] in pairs.  Outputs, any of: y fp32 [M,N], pre_out
 *                           fp32 (pre-activation), y_pairs [M][2 N].  residual fp32 may alias y.  N % 64 == 0, K % 32 == 0; any M.
 *   tt_linear_fwd_pairs_route  which kernel such a call runs: 8 = the persistent gemm_pairs8s_kernel (N % 128 == 0, K % 32 == 0,
 *                           M >= 256, at least 96 tiles of 256 x 128; every epilogue incl. pre_out + GELU pairs), 0 = the general
 *                           kernel.  Profilers' labels only. */
int tt_split_pairs(const float* src, void* dst_pairs, long long n, int* range_flag, tt_stream_t stream);
int tt_join_pairs(const void* src_pairs, float* dst, long long n, tt_stream_t stream);
int tt_layernorm_fwd_pairs(const float* x, const float* gamma, const float* beta, void* y_pairs, float* mean, float* rstd, int rows, int D,
                           float eps, int skip_group, int* range_flag, tt_stream_t stream);
int tt_linear_fwd_pairs(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out,
                        void* y_pairs, int M, int N, int K, int act, void* workspace, size_t workspace_bytes, int* range_flag,
                        tt_stream_t stream);   /* workspace: tt_linear_ksplit_workspace_bytes() or NULL (ABI 7) */
int tt_linear_fwd_pairs_route(int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int has_y_pairs, int has_pre_out);

/* The fused attention core on pair operands (dino_vision_transformer.py:120-132): qkv [F N][2 x 3 H 64] in pairs (the qkv Linear's y_pairs) ->
 * any of out_pairs [F N][2 H 64] (the proj Linear's operand), out_f32 [F, N, H 64] and lse [F, H, N] (what tt_attention_bwd recomputes from).
 * Both products take three fp16 MFMAs per term, as the pair GEMMs do; fp32 scores, softmax and accumulation.  head_dim 64; any N (K / V of a
 * head resident in LDS up to 256 tokens, KV-tiled with an online softmax beyond). */
int tt_attention_fwd_pairs(const void* qkv_pairs, void* out_pairs, float* out_f32, float* lse, int F, int N, int H, int head_dim, float scale,
                           tt_stream_t stream);

/* The backward products of an nn.Linear on pair operands (the "f16x3" mode's backward; autograd of dino_vision_transformer.py:94-103,
 * 115-130 and of the projection head, models.py:915-926):
 *   tt_split_pairs_dual           fp32 [R][C] -> transposed pairs [C][2 Rpad] (rows R..Rpad-1 zero, Rpad % 32 == 0) and, optionally in the
 *                                 same pass, row-major pairs [R][2 C] (C % 32 == 0) and the fp32 column sums [C]: a dy is read ONCE for
 *                                 the operand of its weight-gradient product, the operand of its data-gradient product and its bias
 *                                 gradient.  workspace (tt_split_pairs_dual_workspace_bytes; needed for column sums / a scale).
 *                                 dst_t_pairs may be null (row pairs + column sums only: all the transpose-free weight gradient
 *                                 below needs).  scale_out (device float, or null): a gradient's whole magnitude may sit below fp16's
 *                                 normal range (2^-14: a C2 step's dy tensors peak at 1e-3 .. 1e-6) - the source is then multiplied
 *                                 by the power of two S that brings max |src| into [2^13, 2^14) before the split (exact), S is
 *                                 written to *scale_out, and the products below divide by it (their dy_scale argument, a DEVICE
 *                                 pointer: no host round trip).  The column sums are those of the unscaled source.
 *   tt_transpose_pairs            pairs [R][2 C] -> transposed pairs [C][2 Rpad]: a saved forward operand for the weight gradient.
 *   tt_linear_bwd_data_pairs      dx[M,K] = dy[M,N] @ w[N,K] (* gelu'(gelu_pre[M,K])): dy in pairs [M][2 N], the weight transposed in pairs
 *                                 wT [K][2 N]; N % 32 == 0, K % 64 == 0.
 *   tt_linear_bwd_weight_pairs    dw[N,K] = dy^T @ x: dyT [N][2 Mpad], xT [K][2 Mpad] (K % 64 == 0); split-K partials in the workspace,
 *                                 folded in a fixed order.
 *   tt_linear_bwd_weight_pairs_tn the same product from ROW pairs - dy [M][2 N] (the data-gradient product's operand) and the layer's
 *                                 input x [M][2 K] as the forward kept it: no transposed copies (gemm_pairs_tn.hip: the fragments are
 *                                 gathered by transposing LDS reads).  N % 128 == 0, K % 128 == 0, any M (_ok says whether a shape is
 *                                 taken); partials of the split m range in the workspace, folded in a fixed order.
 *   dy_scale (all three products)  device scalar S or NULL: the dy pairs hold dy * S (tt_split_pairs_dual's scale_out) - the result is
 *                                 divided by S in the epilogue (exact: S is a power of two). */
size_t tt_split_pairs_dual_workspace_bytes(int R, int C, int Rpad);
int tt_split_pairs_dual(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum, float* scale_out, int R, int C, int Rpad,
                        void* workspace, size_t workspace_bytes, int* range_flag, tt_stream_t stream);
/* tt_split_pairs_dual that LEAVES the column partial sums - colsum_parts [ceil(Rpad / 64)][C] fp32, caller-owned - unfolded, for
 * tt_linear_bwd_weight_pairs_tn_bias to fold in the launch that folds the weight gradient's split partials (one launch less per dy;
 * the bias gradient has the same bits either way).  workspace: as tt_split_pairs_dual (needed for a scale only). */
int tt_split_pairs_dual_parts(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum_parts, float* scale_out, const float* amax_in,
                              int R, int C, int Rpad, void* workspace, size_t workspace_bytes, int* range_flag, tt_stream_t stream);
/* amax_in (with scale_out): the amax slot the kernel that wrote src raised (amax_out above) - the split then needs no max pass and no
 * workspace; NULL: it makes its own. */
int tt_transpose_pairs(const void* src_pairs, void* dst_t_pairs, int R, int C, int Rpad, tt_stream_t stream);
/* tt_split_pairs_dual (without column sums) for n matrices in ONE launch per 32 of them: host arrays of n pointers / sizes; dst_t_pairs[i] or
 * dst_row_pairs[i] may be null.  What a training step needs of every weight the optimizer rewrote (row pairs: forward and weight-gradient
 * operand; transposed pairs: the data-gradient operand). */
int tt_split_pairs_dual_multi(const float* const* src, void* const* dst_t_pairs, void* const* dst_row_pairs, const int* R, const int* C,
                              const int* Rpad, int n, int* range_flag, tt_stream_t stream);
int tt_linear_bwd_data_pairs(const void* dy_pairs, const void* wT_pairs, const float* gelu_pre, float* dx, const float* dy_scale, int M, int N, int K,
                             void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream);   /* workspace: the K-split block or NULL (ABI 7) */
size_t tt_linear_bwd_weight_pairs_workspace_bytes(int N, int K, int Mpad);
int tt_linear_bwd_weight_pairs(const void* dyT_pairs, const void* xT_pairs, float* dw, const float* dy_scale, int N, int K, int Mpad, void* workspace,
                               size_t workspace_bytes, tt_stream_t stream);
int tt_linear_bwd_weight_pairs_tn_ok(int N, int K, int M);
size_t tt_linear_bwd_weight_pairs_tn_workspace_bytes(int N, int K, int M);
int tt_linear_bwd_weight_pairs_tn(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M, void* workspace,
                                  size_t workspace_bytes, tt_stream_t stream);
/* ... and the bias gradient db [N] = the column sums of dy from their partials (colsum_parts [colsum_count][N] of
 * tt_split_pairs_dual_parts) in the same fold launch. */
int tt_linear_bwd_weight_pairs_tn_bias(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M, void* workspace,
                                       size_t workspace_bytes, const float* colsum_parts, int colsum_count, float* db, tt_stream_t stream);
/*   prepare_tokens (dino_vision_transformer.py:166-171,236-247) on pair operands: tt_patch_embed_fwd with the conv weight in pairs
 *   [D][2 C P P] (tt_split_pairs of patch_w viewed [D, C P P]); the patches are split into pairs on their way into an im2col buffer
 *   (workspace: the rows, then the GEMM's K-split block, whose counters the call zeroes itself), ONE pair GEMM over all F (n + 1) rows
 *   leaves the tokens.  P % 4 == 0, W % 4 == 0, C P P % 32 == 0, D % 64 == 0. */
size_t tt_patch_embed_pairs_workspace_bytes(int F, int C, int H, int W, int P);
int tt_patch_embed_fwd_pairs(const float* img, const int32_t* frame_map, const void* w_pairs, const float* bias, const float* cls, const float* pos,
                             float* tokens, int F, int C, int H, int W, int P, int D, void* workspace, size_t workspace_bytes, int* range_flag,
                             tt_stream_t stream);
/*   The backward products of the same nn.Linear sites on bf16-plane operands (autograd of dino_vision_transformer.py:94-103,
 *   115-130; the bf16 path only - the fp32 modes keep the f32-MFMA backward kernels):
 *   tt_transpose_planes               fp32 [R][C] -> bf16 [C][Rpad], transposed, columns R..Rpad-1 zero (reduction index contiguous)
 *   tt_linear_bwd_data_planes         dx[M,K] = dy[M,N] @ w[N,K] (* gelu'(gelu_pre[M,K]) when given): dy in planes [M][N], the weight
 *                                     transposed in planes wT [K][N]; N % 64 == 0, K % 64 == 0
 *   tt_linear_bwd_weight_planes       dw[N,K] = dy[M,N]^T @ x[M,K]: dyT [N][Mpad] and xT [K][Mpad] from tt_transpose_planes with the
 *                                     same Mpad (a multiple of 64); split-K over Mpad with a fixed-order fold; N, K % 64 == 0 */
int tt_transpose_planes(const float* src, void* dst, int R, int C, int Rpad, tt_stream_t stream);
/* The same pass that ALSO leaves the fp32 column sums of src in colsum [C] (the bias gradient dy.sum(0) of an nn.Linear whose dy is
 * being transposed for tt_linear_bwd_weight_planes: one read of dy instead of two); workspace: ceil(Rpad / 64) x C floats. */
size_t tt_transpose_planes_colsum_workspace_bytes(int R, int C, int Rpad);
int tt_transpose_planes_colsum(const float* src, void* dst, int R, int C, int Rpad, float* colsum, void* workspace, size_t workspace_bytes,
                               tt_stream_t stream);
int tt_linear_bwd_data_planes(const void* dy_planes, long long dy_plane_stride, const void* wT_planes, long long wT_plane_stride,
                              int planes, const float* gelu_pre, float* dx, int M, int N, int K, void* workspace, size_t workspace_bytes,
                              tt_stream_t stream);   /* workspace: the K-split block or NULL (ABI 7) */
int tt_linear_bwd_weight_planes(const void* dyT_planes, long long dyT_plane_stride, const void* xT_planes, long long xT_plane_stride,
                                int planes, float* dw, int N, int K, int Mpad, void* workspace, size_t workspace_bytes,
                                tt_stream_t stream);
size_t tt_linear_bwd_weight_planes_workspace_bytes(int N, int K, int Mpad);

/* ---- k14: temporal label propagation (time_tuning.py:143-154 -> mask_propagation.py:396-496)
 *   xn   [fs, bs, n, D]  L2-normalised backbone tokens, time-major (frame t of clip b at [t][b])
 *   seg0 [bs, n, K] fp32 Sinkhorn assignment of frame 0 (the seed labels)
 *   labels [bs, n] int64 = argmax_K of the propagated map of the LAST frame
 *   pmap_last [bs, n, K] fp64 optional (the map itself).
 *   workspace: tt_label_propagate_workspace_bytes(...). */
int tt_label_propagate(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g,
                       int D, int K, int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace,
                       size_t workspace_bytes, tt_stream_t stream);
size_t tt_label_propagate_workspace_bytes(int bs, int fs, int g, int D, int K, int n_last_frames);
/* The same in two calls, for a caller that overlaps the halves with other work (round 6: the cosine similarities do not depend on seg0 - the
 * training step computes them on a side stream beside the Sinkhorn solve that produces seg0):
 *   tt_label_propagate_sims       the similarities of ALL target frames into `workspace` (tt_label_propagate_workspace_bytes).  Returns TT_OK,
 *                                 or 1 when they do not fit one chunk of this workspace (nothing is written: call tt_label_propagate);
 *   tt_label_propagate_from_sims  tt_label_propagate on a workspace tt_label_propagate_sims prepared (same shapes, same workspace).
 * Together they launch exactly what tt_label_propagate launches. */
int tt_label_propagate_sims(const float* xn, int bs, int fs, int g, int D, int K, int n_last_frames, int precision, void* workspace,
                            size_t workspace_bytes, tt_stream_t stream);
int tt_label_propagate_from_sims(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g, int D, int K,
                                 int n_last_frames, int radius, int topk, float temperature, void* workspace, size_t workspace_bytes,
                                 tt_stream_t stream);

/* ---- N4: label-propagation EVALUATION (mask_propagation.py:816-833, DAVIS protocol
 *      n_last_frames 4, size_mask_neighborhood 12, topk 5), reusing k14.
 *   tt_label_propagate_maps  same propagation, but ALL fs-1 maps are returned, as propagate_labels does
 *                            (mask_propagation.py:448-496): pmap_all [fs-1, bs, n, K] fp64.  Workspace as above.
 *   tt_upsample_argmax       nn.functional.interpolate(maps, (R,R), mode="bilinear", align_corners=False) followed by
 *                            torch.max(dim=1) (mask_propagation.py:828-829), fused: maps [M, n, K] fp64 ->
 *                            labels_out [M, R, R] int64; the upsampled fp64 tensor is never written.
 *   tt_confusion_counts      counts[gt*C + pred] += 1 over n pixels (labels outside [0,C) ignored), uint64 [C,C],
 *                            C <= 4096: the confusion matrix from which the Jaccard index (J) of the propagated masks and
 *                            the evaluator's matched mIoU (metrics.py:357-432) follow. */
int tt_label_propagate_maps(const float* xn, const float* seg0, double* pmap_all, int bs, int fs, int g, int D, int K,
                            int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace,
                            size_t workspace_bytes, tt_stream_t stream);
int tt_upsample_argmax(const double* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream);
int tt_confusion_counts(const int64_t* pred, const int64_t* gt, long long n, int C, unsigned long long* counts,
                        tt_stream_t stream);

/* ---- k15: CrossEntropyLoss(scores/temp, labels), mean over patches then batch
 *      (time_tuning.py:296-302), with its gradient w.r.t. scores.
 *   scores [rows,K]; labels int64[rows]; loss_out[1]; dscores [rows,K] (= d loss / d scores).
 *   row_weight [rows] or NULL: the --use_mask form, CrossEntropyLoss(reduction='none') * mask followed by the
 *   mean over ALL rows (time_tuning.py:226-227,298-300).
 *   workspace: tt_ce_workspace_bytes(rows). */
int tt_ce_loss_fwd_bwd(const float* scores, const int64_t* labels, const float* row_weight, float* loss_out, float* dscores,
                       int rows, int K, float temperature, void* workspace, size_t workspace_bytes, tt_stream_t stream);
size_t tt_ce_workspace_bytes(int rows);

/* ---- k13: queue FIFO update (time_tuning.py:258-261): shift down by m rows, write feats[idx[i]]
 *   at the head.  queue [Q,D]; feats [R,D]; idx int64[m].  scratch [Q,D] (caller-owned). */
int tt_queue_push(float* queue, float* scratch, const float* feats, const int64_t* idx, int Q, int D, int m,
                  tt_stream_t stream);

/* ---- k17: AdamW over a table of tensors (time_tuning.py:413-429; torch.optim.AdamW defaults)
 *   Up to TT_MAX_TENSORS per call.  step is the 1-based step count of every tensor in the call. */
#define TT_MAX_TENSORS 40
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
  float lr;
  float weight_decay;
} tt_adamw_tensor;
int tt_adamw_step(const tt_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps,
                  tt_stream_t stream);

/* g[i] *= *scale_device for the `g` buffers (n floats each) of up to TT_MAX_TENSORS table entries (p, m, v, lr, weight_decay are
 * ignored), one launch: the chain rule of `loss.backward()` applied to the gradients the fused step produced
 * (time_tuning.py:420-423 calls backward on the scalar loss; its incoming gradient is a device scalar). */
int tt_scale_tensors(const tt_adamw_tensor* tensors, int count, const float* scale_device, tt_stream_t stream);

/* ---- k19: EMA teacher update (time_tuning.py:109-118): t = t*(1-m) + s*m over n floats. */
int tt_ema_update(float* teacher, const float* student, long long n, double momentum, tt_stream_t stream);

/* ---- misc elementwise used between the sites above */
int tt_add_inplace(float* dst, const float* src, long long n, tt_stream_t stream);
/* Number of positions where two fp32 buffers differ BITWISE.  Decides, once, whether the EMA teacher's frozen tensors still equal the student's
 * (time_tuning.py:113-114 blends identical tensors for every frozen parameter), i.e. whether the teacher pass may reuse the
 * student's frozen-block activations.  count_out: device int64. */
int tt_count_mismatch(const float* a, const float* b, long long n, long long* count_out, tt_stream_t stream);

/* ---- N1: attention foreground mask, models.process_attentions (models.py:93-131), used by
 *      apply_attention_mask (models.py:133-144) on the --use_mask branch of TimeT.get_loss.
 *   Per frame: cls-query attention of the LAST block averaged over heads -> Gaussian blur (ksize x ksize, sigma,
 *   reflect padding; the reference uses 7 / 0.6) -> keep `threshold` (0.65) of the mass by ascending sort +
 *   cumulative sum -> drop 8-connected components of <= 2 pixels.  One launch, no host round trip (the
 *   reference labels components with skimage on the CPU).
 *   tt_foreground_mask           qkv [F,N,3*H*hd]: the last block's qkv activations (same buffer tt_attention_fwd
 *                                reads); the cls-row probabilities softmax(q_cls k_j * scale) are recomputed in the
 *                                kernel, so attn[F,H,N,N] is never materialised.
 *   tt_foreground_mask_from_probs cls_probs [F,H,N]: row 0 of the attention probabilities, already computed.
 *   mask_out [F,g*g] floats in {0,1}; blurred_out [F,g*g] optional (the blurred mean attention);
 *   margin_out [F,g*g] optional (|cumulative mass - (1-threshold)| per pixel: how far it is from the cut). */
int tt_foreground_mask(const float* qkv, float* mask_out, float* blurred_out, float* margin_out, int F, int N, int H, int hd, int g,
                       float scale, float threshold, float sigma, int ksize, tt_stream_t stream);
int tt_foreground_mask_from_probs(const float* cls_probs, float* mask_out, float* blurred_out, float* margin_out, int F, int N,
                                  int H, int g, float threshold, float sigma, int ksize, tt_stream_t stream);

/* ---- N2: evaluator clustering - the device side of clustering.cluster_features / proto_clustering
 *      (clustering.py:20-117), my_utils.normalize_and_transform (my_utils.py:19-37) and of the Lloyd iterations the
 *      reference delegates to faiss.Kmeans(d, k, niter=50, nredo=5, seed=1).
 *   tt_col_moments              per-column mean and population variance (StandardScaler, my_utils.py:24-30), fp64.
 *   tt_upsample_bilinear_tokens token maps [M, g*g, C] -> [M, R*R, C] as nn.functional.interpolate(x.double(), (R,R),
 *                               mode="bilinear").float() (clustering.py:34-36).
 *   tt_upsample_argmax_f32      fp32 twin of tt_upsample_argmax for proto_clustering's prototype scores (clustering.py:101-104).
 *   tt_kmeans_assign            labels[p] = argmin_j |x_p - c_j|^2 (first minimum) over centroids [k, d]; dist2 optional.
 *   tt_kmeans_accumulate        sums[k, d] (fp64) and counts[k] of the points per label, deterministic (no atomics). */
int tt_affine_cols_inplace(float* x, const float* scale, const float* shift, long long rows, int cols,
                           tt_stream_t stream); /* x[r][c] = x[r][c] * scale[c] + shift[c] (StandardScaler.transform) */
size_t tt_col_moments_workspace_bytes(long long rows, int cols);
int tt_col_moments(const float* x, double* mean, double* var, long long rows, int cols, void* workspace, size_t workspace_bytes,
                   tt_stream_t stream);
int tt_upsample_bilinear_tokens(const float* x, float* out, int M, int g, int C, int R, tt_stream_t stream);
int tt_upsample_argmax_f32(const float* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream);
int tt_kmeans_assign(const float* x, const float* centroids, int32_t* labels, float* dist2, long long P, int d, int k,
                     tt_stream_t stream);
size_t tt_kmeans_accumulate_workspace_bytes(long long P, int d, int k);
int tt_kmeans_accumulate(const float* x, const int32_t* labels, double* sums, long long* counts, long long P, int d, int k,
                         void* workspace, size_t workspace_bytes, tt_stream_t stream);

/* ---- N3: the clip input pipeline - the pixel work of video_transformations.py as wired at
 *      time_tuning.py:588-593, bit-exact with Pillow (which the reference calls per frame on the host).
 *   Frames are interleaved uint8 RGB [F, H, W, 3] in device memory.
 *   tt_img_resample_h   horizontal pass of Image.resize(BILINEAR) (Resample.c) over the crop rows y0..y0+h, columns starting
 *                       at x0: out [F, h, OW, 3].  coeffs int32 [OW, ksize] / bounds int32 [OW, 2] are Resample.c's 22-bit taps
 *                       (device memory; timetuning_amd.video_transformations.resample_coeffs builds them on the host).
 *   tt_img_resample_v   vertical pass over in [F, Hin, W, 3] starting at row y0: either out_u8 [F, OH, W, 3] or out_f32
 *                       [F, 3, OH, W] = ClipToTensor(mean, std) of the (optionally horizontally flipped) result
 *                       (video_transformations.py:168-179,262-276); mean3 / std3 are HOST pointers to 3 floats.
 *   tt_img_color        in place: mode 0 RandomGrayscale's convert("L") replicated to 3 channels; 1 / 2 / 3 torchvision
 *                       adjust_brightness / adjust_contrast / adjust_saturation (ImageEnhance = Blend.c); 4 adjust_hue
 *                       (hue_shift = uint8(hue_factor * 255)).  gray_sums: F uint64 of workspace for mode 2.
 *   tt_img_box_blur     one pass of BoxBlur.c along x (direction 0) or y (1); ImageFilter.GaussianBlur(radius) is three x
 *                       passes then three y passes with (radius, ww, fw) from the box radius (video_transformations.py:604-647). */
int tt_img_resample_h(const unsigned char* in, unsigned char* out, const int* coeffs, const int* bounds, int F, int H, int W, int y0, int x0,
                      int h, int OW, int ksize, tt_stream_t stream);
int tt_img_resample_v(const unsigned char* in, unsigned char* out_u8, float* out_f32, const int* coeffs, const int* bounds, int F, int Hin,
                      int W, int y0, int OH, int ksize, int flip, const float* mean3, const float* std3, tt_stream_t stream);
int tt_img_color(unsigned char* img, int F, int H, int W, int mode, float factor, int hue_shift, unsigned long long* gray_sums,
                 tt_stream_t stream);
int tt_img_box_blur(const unsigned char* in, unsigned char* out, int F, int H, int W, int direction, int radius, unsigned ww, unsigned fw,
                    tt_stream_t stream);

/* ---- Coarse entry points: whole reference functions as ONE call each.  They sequence the op-level entry
 *      points above on `stream` (same kernels, same results bit for bit as calling those one by one; they honour
 *      their `precision` argument / tt_vit_params.precision the way tt_linear_fwd does) and add nothing but the scratch layout.  Parameter tables are HOST
 *      arrays of DEVICE pointers, read during the call only.
 *
 *   tt_vit_forward       VisionTransformer.prepare_tokens + blocks + norm (dino_vision_transformer.py:236-252,265-273; Block
 *                        :135-153, Attention :108-132, Mlp :89-105), i.e. FeatureExtractor.get_features' backbone pass
 *                        (models.py:965-969).  img [F_src, C, H, W] and frame_map (int32 [F], optional) as tt_patch_embed_fwd;
 *                        img == NULL: `tokens` already holds a residual stream [F, N, D] (an EMA teacher continuing from the
 *                        student's frozen blocks) and p->patch_* / cls / pos are not read.  Blocks p->blocks[0 .. n_blocks) run
 *                        in place on tokens [F, N, D], N = 1 + (H / patch)(W / patch).  normed (optional): the final LayerNorm of
 *                        the tokens, [F, N, D] or, with drop_cls != 0, [F, N - 1, D] (get_features drops the cls token).
 *                        last_qkv (optional, [F, N, 3 D]): the last block's qkv activations (what tt_foreground_mask reads);
 *                        last_probs (optional, [F, heads, N, N]): its attention probabilities (get_last_selfattention, :256-263).
 *   tt_mlp_head_forward  the projection head (models.py:915-926,1075-1077): Linear (GELU Linear)*, x [M, layers[0].in_features]
 *                        -> out [M, layers[n - 1].out_features].
 *   tt_scores_sinkhorn   TimeT.get_scores (time_tuning.py:195-217) for one rank: scores = normalize(z) @ prototypes^T for the
 *                        batch rows z [B, dim] and, when queue != NULL, the queue rows [queue_rows, dim] (:207-211), then the
 *                        Sinkhorn assignment over all B + queue_rows columns; q_out [rows_out, K] = the first rows_out rows.
 *                        scores [B + queue_rows, K] is an output too (batch_scores = its first B rows).  For world_size > 1 the
 *                        caller all-gathers `scores` and calls tt_sinkhorn on the gathered rows (the "gathered pointer" form).
 *   tt_adamw_ema_step    optimizer.step() + normalize_prototypes() + update_momentum_teacher() (time_tuning.py:659-663 ->
 *                        :413-429, :124-128, :109-122): AdamW over `tensors` (any count), prototypes [K, dim] renormalised in
 *                        place (skipped when NULL), then - when teacher_flat != NULL - teacher <- teacher (1 - m) + student m
 *                        over the flat parameter buffers and the teacher prototypes, which are renormalised too. */
typedef struct {
  const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  const void *qkv_wp, *proj_wp, *fc1_wp, *fc2_wp;   /* planes > 0: the four weights as bf16 planes [planes][out][in] (tt_split_planes),
                                                       or, planes == 2, as fp16 pairs [out][2 in] (tt_split_pairs) */
} tt_vit_block_params;
typedef struct {
  const float *patch_w, *patch_b, *cls, *pos;   /* [D, C P P], [D], [D], [N, D] (position table at the input's grid) */
  const tt_vit_block_params* blocks;            /* host array, n_blocks entries */
  int n_blocks;
  const float *norm_w, *norm_b;                 /* final LayerNorm (may be NULL when normed == NULL) */
  int dim, heads, hidden, patch;                /* D, attention heads (head_dim = D / heads), MLP width, patch size */
  int planes;                                   /* 0: fp32 operands (tt_linear_fwd); 1 / 3: the bf16-plane path of the blocks
                                                   (tt_linear_fwd_planes; 1 = BASELINE C4's bf16 path, 3 = fp32-accurate), D % 64 == 0;
                                                   2: fp16 PAIRS (tt_linear_fwd_pairs, the fp32-accurate "f16x3" mode; *_wp are
                                                   tt_split_pairs of the weights), D % 64 == 0 and hidden % 64 == 0 */
  const void* patch_wp;                         /* optional.  planes == 1: patch_w as one bf16 plane [D, C P P] - prepare_tokens then runs
                                                   tt_patch_embed_fwd_planes where its shape rules hold (ABI 4); planes == 2: patch_w in
                                                   pairs [D][2 C P P] - tt_patch_embed_fwd_pairs where ITS rules hold and C P P <= 3 D
                                                   (the rows then fit the scratch) (ABI 6) */
  int* range_flag;                              /* planes == 2: the pair producers' range flag (device int or NULL; ABI 7) */
  int precision;                                /* planes == 0: the `precision` of the blocks' tt_linear_fwd products (TT_PRECISION_*; ABI 8) */
} tt_vit_params;
typedef struct {
  const float* w;   /* [out_features, in_features] */
  const float* b;   /* [out_features] or NULL */
  int out_features, in_features;
} tt_linear_params;
size_t tt_vit_forward_workspace_bytes(int F, int N, int D, int hidden, int planes);
int tt_vit_forward(const tt_vit_params* p, const float* img, const int32_t* frame_map, int F, int C, int H, int W, float* tokens,
                   float* normed, int drop_cls, float* last_qkv, float* last_probs, void* workspace, size_t workspace_bytes,
                   tt_stream_t stream);
size_t tt_mlp_head_forward_workspace_bytes(int M, const tt_linear_params* layers, int n_layers);
int tt_mlp_head_forward(const float* x, int M, const tt_linear_params* layers, int n_layers, float* out, int precision, void* workspace,
                        size_t workspace_bytes, tt_stream_t stream);
size_t tt_scores_sinkhorn_workspace_bytes(int B, int queue_rows, int K, int dim);
int tt_scores_sinkhorn(const float* z, int B, const float* queue, int queue_rows, const float* prototypes, int K, int dim,
                       float* scores, float* q_out, int rows_out, float eps, int iters, int precision, void* workspace,
                       size_t workspace_bytes, tt_stream_t stream);
int tt_adamw_ema_step(const tt_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps, float* prototypes,
                      int K, int dim, float* teacher_flat, const float* student_flat, long long n_flat, float* teacher_prototypes,
                      double momentum, tt_stream_t stream);

/* features * mask[..., None] (models.py:142) and its backward: x[r][:] *= row_scale[r], cols % 4 == 0. */
int tt_scale_rows_inplace(float* x, const float* row_scale, int rows, int cols, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TIMETUNING_HIP_H */
