/*
 * vitk: C ABI of the MI355X (gfx950) kernels behind the ViT forward path.
 *
 * The reference (ZhiyuanChen/vit) has no FFI or operator plug-in interface: its
 * forward path is a Python nn.Module that dispatches stock torch ops.  This
 * library sits UNDER a module with that same Python surface
 * (vit_amd/vision_transformer.py), one entry point per fused kernel; each
 * comment names the reference call sites the entry point replaces (paths are
 * relative to the reference tree).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller; the library never
 *    allocates, frees or retains memory.  Tensors are contiguous row-major unless
 *    a stride argument says otherwise, and 16-byte aligned.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    Calls only enqueue work; nothing synchronises.
 *  - Return value: VITK_OK, or an error code; vitk_last_error() then returns a
 *    message for the calling thread.  No C++ exception crosses this boundary.
 *  - bf16 tensors are raw 16-bit brain-float words (torch.bfloat16 storage).
 */
#ifndef VITK_H_
#define VITK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_OK 0
#define VITK_ERR_INVALID 1      /* null pointer, non-positive size, misaligned pointer */
#define VITK_ERR_UNSUPPORTED 2  /* a shape or dtype no kernel of this library covers */
#define VITK_ERR_LAUNCH 3       /* the HIP runtime rejected the launch */

#define VITK_F32 0
#define VITK_BF16 1

/* ABI version of this header (bumped on any signature change). */
int vitk_version(void);

/* Message for the last non-zero return on the calling thread ("" if none). */
const char* vitk_last_error(void);

/* Test and measurement hook (one setting per PROCESS, default 0 = production kernels; it also reaches kernels
 * launched from other threads, e.g. torch's autograd workers; not meant to be changed while launches are being issued): choose an
 * alternative implementation of the tiled GEMM and of attention.
 * gemm_variant, low 4 bits: 0 (and 4) = default (two-group phased kernel where K/64 is even and A is a
 *   plain matrix, else the one-group pipelined kernel), 1 = basic kernel with LDS-DMA staging, 2 = force
 *   the small untiled kernel, 3 = basic kernel with register staging, 5 = one-group pipelined kernel.
 *   gemm_variant >> 4 (= "order"): bits 0-3 tile order (1 = no n-grouping, 3 = group width from bits
 *   4-7), bits 8-12 force the tile height (in 16-row units), bit 13 = equal tiles where the two-height plan
 *   would apply.
 *   Timing probes (WRONG OR NO RESULTS) exist only in diagnostic builds of the library (-DVITK_PROBES,
 *   `python -m vit_amd.build --probes` -> libvitk_probes.so), never in libvitk.so: order bits 17-19 on the
 *   bias -> bf16 epilogue of the default kernel (1 = no DMA, 2 = no epilogue, 3 = accumulator clearing
 *   instead of the zero C operand, 7 = cycle stamps written over out[0..]), order bit 22 = bf16 epilogues
 *   without stores.
 * attention_variant, low 2 bits: 0 = automatic (persistent workgroups that hold two heads' K/V in LDS and
 *   stage the next head under the current one's key loop when two heads fit: ntok <= 256 at head_dim 64,
 *   160 at 80 / 96; one head per workgroup when one fits: ntok <= 608 / 288; else streamed), 1 = one head
 *   per workgroup with V stored transposed (head_dim 64), 2 = K/V streamed through LDS in 64-key chunks,
 *   3 = one head per workgroup; bits 5-8 force the waves per workgroup (4..8; 10 and, at head_dim 64 with one
 *   head per workgroup, 12 in the resident modes; 0 = chosen from the token count); bit 9 = the attention backward
 *   as two kernels (dQ, then dK / dV) also where the one-kernel form applies (ntok <= 288); diagnostic builds only:
 *   bit 2 = K/V of head 0 for every head, bit 3 = no key loop, bit 4 = no output stores.
 * Results are identical across variants up to fp32 summation order. */
void vitk_set_debug_variant(int gemm_variant, int attention_variant);

/* Measurement hook (one setting per process): plan the persistent kernels (tiled GEMMs, attention, weight-gradient
 * GEMM) for `cus` compute units instead of all of them; 0 restores the default.  Results do not depend on it.  Used to
 * measure what a kernel gains per CU when part of the chip is idle (tools/exp_half_chip.sh) and to run micro-batches
 * side by side on separate streams (tools/two_stream.py). */
void vitk_set_cu_limit(int cus);

/* y[r, :] = LayerNorm(x[r, :]) * gamma + beta over D columns, fp32 statistics.
 * x: f32, row r at x + r*x_row_stride; y: bf16 or f32 (out_dtype), row stride y_row_stride
 * (strides in elements).  D in {256, 512, 768, 1024, 1280}.
 * Replaces nn.LayerNorm at models/transformer.py:68,73,98 (built :56,60,93). */
int vitk_layernorm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y,
                   int64_t y_row_stride, int rows, int D, float eps, int out_dtype, void* stream);

/* out[i] = bf16(in[i]), any n > 0.  Input images arrive fp32 when the caller has not cast them. */
int vitk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);

/* Patch embedding + cls token + position embedding:
 *   x[b, 0, :]     = cls + pos[0, :]
 *   x[b, 1 + p, :] = conv(img)[b, :, p] + bias + pos[1 + p, :]
 * img: bf16 [batch, 3, img_h, img_w]; w: bf16 [D, 3*patch*patch] (Conv2d weight, OIHW flattened);
 * bias, cls: f32 [D]; pos: f32 [1 + (img_h/patch)*(img_w/patch), D]; x: f32 [batch, ntok, D].
 * patch must be a power of two >= 8, img_w % 8 == 0, D % 256 == 0.
 * Replaces nn.Conv2d + flatten + transpose + cat + add at models/vision_transformer.py:47-50. */
int vitk_patch_embed(const void* img, const void* w, const float* bias, const float* cls, const float* pos,
                     float* x, int batch, int img_h, int img_w, int patch, int D, void* stream);

/* The same result for any patch size (ViT-H/14's 14 included), through an explicit patch matrix:
 * cols: caller-provided scratch, bf16 [batch * (img_h/patch) * (img_w/patch), Kpad] with
 * Kpad = 3*patch*patch rounded up to a multiple of 64; w_padded: bf16 [D, Kpad], the flattened Conv2d
 * weight zero-padded to Kpad columns.  D % 256 == 0; any image size (also the route for a width that is not a
 * multiple of 8).
 * models/vision_transformer.py:47-50. */
int vitk_patch_embed_cols(const void* img, const void* w_padded, const float* bias, const float* cls,
                          const float* pos, float* x, void* cols, int batch, int img_h, int img_w, int patch, int D,
                          void* stream);

/* out[M,N] = A[M,K] @ W[N,K]^T + bias, A and W bf16, fp32 accumulate; out bf16 or f32 (out_dtype).
 * bias may be NULL.  K % 32 == 0.  Replaces nn.Linear (models/vision_transformer.py:30,55). */
int vitk_gemm_bias(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                   int out_dtype, void* stream);

/* out bf16 [M,N] = gelu_erf(A @ W^T + bias).  Replaces fc1 + nn.GELU at models/transformer.py:21-22. */
int vitk_gemm_bias_gelu(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                        void* stream);

/* out bf16 [M,N] = tanh(A @ W^T + bias).  Replaces pre_logits + tanh at models/vision_transformer.py:53-54. */
int vitk_gemm_bias_tanh(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                        void* stream);

/* out f32 [M,N] = resid + A @ W^T + bias; out may alias resid (in-place residual update).
 * Replaces out_proj + residual add (models/transformer.py:47,72) and fc2 + residual add (:24,76). */
int vitk_gemm_bias_residual(const void* A, const void* W, const float* bias, const float* resid, float* out,
                            int M, int N, int K, void* stream);

/* Fused q/k/v projection with the head split:
 *   qkv[which, b, h, t, e] = (A @ W^T + bias)[b*ntok + t, which*D + h*hd + e],  D = heads*hd
 * A: bf16 [batch*ntok, D]; W: bf16 [3D, D]; bias f32 [3D] or NULL; qkv: bf16 [3, batch, heads, ntok, hd].
 * Replaces in_proj + reshape + permute at models/transformer.py:42. */
int vitk_qkv_proj(const void* A, const void* W, const float* bias, void* qkv, int batch, int ntok, int heads,
                  int hd, void* stream);

/* ctx[b, t, h*hd + e] = sum_s softmax_s(q[b,h,t,:] . k[b,h,s,:] * scale) * v[b,h,s,e]
 * q, k, v: bf16 [batch, heads, ntok, hd]; ctx: bf16 [batch, ntok, heads*hd].  hd in {64, 80, 96}, ntok <= 16384
 * (K and V of a head stay resident in LDS when they fit, else they stream through it in 64-key chunks);
 * other head sizes return VITK_ERR_UNSUPPORTED.
 * Replaces q@k^T, scaling, softmax, @v, transpose, reshape at models/transformer.py:43-46. */
int vitk_attention(const void* q, const void* k, const void* v, void* ctx, int batch, int heads, int ntok,
                   int hd, float scale, void* stream);

/* ---- fp32 compute mode -------------------------------------------------------------------------
 * The reference runs in the dtype of its parameters (fp32 unless the caller casts the model); these
 * entry points restate the same ops with f32 operands, f32 storage and exact f32 multiply-add (the
 * f32-input MFMA), for callers that need max-abs logit error <= 1e-4 against the reference's CPU
 * forward instead of bf16 throughput.  LayerNorm is vitk_layernorm with out_dtype = VITK_F32. */

#define VITK_EPI_BIAS 0      /* out = A W^T + bias */
#define VITK_EPI_GELU 1      /* out = gelu(A W^T + bias), exact erf (nn.GELU(), models/transformer.py:16) */
#define VITK_EPI_RESIDUAL 2  /* out = resid + A W^T + bias; out may alias resid (models/transformer.py:69,75) */
#define VITK_EPI_TANH 3      /* out = tanh(A W^T + bias) (models/vision_transformer.py:53-54) */

/* A f32 [M,K], W f32 [N,K] (nn.Linear weight), bias f32 [N] or NULL, out f32 [M,N]; any M, N, K.
 * Replaces nn.Linear (+ GELU / residual add / tanh) at models/transformer.py:21-25,42,47,69,75 and
 * models/vision_transformer.py:53-55 when the model computes in fp32. */
int vitk_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N,
                  int K, int epilogue, void* stream);

/* qkv f32 [3, batch, heads, ntok, hd] <- (A W^T + bias) split head-major; A f32 [batch*ntok, heads*hd].
 * fp32 counterpart of vitk_qkv_proj (models/transformer.py:42). */
int vitk_qkv_proj_f32(const float* A, const float* W, const float* bias, float* qkv, int batch, int ntok, int heads,
                      int hd, void* stream);

/* fp32 counterpart of vitk_patch_embed: img f32 [batch,3,img_h,img_w], w f32 [D, 3*patch*patch]; any patch
 * size (14 for ViT-H/14 included).  models/vision_transformer.py:47-50. */
int vitk_patch_embed_f32(const float* img, const float* w, const float* bias, const float* cls, const float* pos,
                         float* x, int batch, int img_h, int img_w, int patch, int D, void* stream);

/* fp32 counterpart of vitk_attention: q, k, v f32 [batch, heads, ntok, hd], ctx f32 [batch, ntok, heads*hd];
 * hd in {16, 32, 64, 80, 96, 128}, ntok <= 1024.  models/transformer.py:43-46. */
int vitk_attention_f32(const float* q, const float* k, const float* v, float* ctx, int batch, int heads, int ntok,
                       int hd, float scale, void* stream);

/* ---- next to the forward path --------------------------------------------------------------- */

/* Position-embedding rescale for loading a checkpoint at another resolution:
 * pos_out[0] = pos_in[0] (cls); the grid_in x grid_in grid of rows pos_in[1:] is resampled to grid_out x
 * grid_out by order-1 interpolation, output index o at input coordinate o (grid_in-1)/(grid_out-1).
 * pos_in f32 [1 + grid_in^2, D], pos_out f32 [1 + grid_out^2, D], distinct buffers.
 * Replaces pos_embed_scale (scipy.ndimage.zoom, order=1, mode='constant') at utils.py:241-261. */
int vitk_pos_embed_scale(const float* pos_in, float* pos_out, int grid_in, int grid_out, int D, void* stream);

/* Input step in front of the forward: out[b,c,y,x] = (float(img[b,c,y,x]) - mean[c]) / std[c].
 * img: uint8 [batch,3,img_h,img_w] (what fast_collate builds, data.py:143-154); mean, std: HOST pointers to 3
 * floats in pixel units (the reference's 0.485*255 ... / 0.229*255 ..., data.py:54-55); out bf16 or f32
 * (out_dtype), img_h*img_w % 16 == 0.  Replaces .float().sub_(mean).div_(std) at data.py:69-72. */
int vitk_normalize_u8(const void* img, void* out, int batch, int img_h, int img_w, const float* mean,
                      const float* std, int out_dtype, void* stream);

/* Step after the forward: rank[r] = number of classes ranked above target[r] in logits[r, :] (ties towards
 * the lower class index), so the prediction is correct at k iff rank[r] < k; targets outside [0, classes)
 * get rank = classes.  logits f32 [rows, classes], target int64 [rows], rank int32 [rows].
 * Replaces topk + eq + sum of accuracy() at utils.py:182-195. */
int vitk_target_rank(const float* logits, const int64_t* target, int* rank, int rows, int classes, void* stream);

/* Backward of vitk_layernorm for the training step (the reference trains through nn.LayerNorm's autograd,
 * main.py:207-218): dx[r, :] (f32 [rows, D], contiguous), dgamma, dbeta (f32 [D]) from x (f32, row r at
 * x + r*x_row_stride, as in the forward), gamma, and the output gradient g (bf16 or f32 [rows, D], g_dtype).
 * Mean and rstd are recomputed from x.  add (may be NULL): f32 [rows, D] added to dx, the gradient reaching x
 * along the residual branch (may alias dx).  dx_bf16 (may be NULL): dx again, rounded to bf16 [rows, D], for the
 * backward GEMMs that take it as an operand.  partials: caller's scratch, f32 [partial_rows, 2, D]; the kernel uses
 * min(partial_rows, ceil(rows/4), 1024) workgroups, one partial each, added in a fixed order. */
int vitk_layernorm_backward(const float* x, int64_t x_row_stride, const float* gamma, const void* g, int g_dtype,
                            const float* add, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* partials, int partial_rows, int rows,
                            int D, float eps, void* stream);

/* Weight gradient of a Linear layer in the training step: C[m][n] = sum_r G[r][m] * X[r][n] (dW = dY^T X, what
 * autograd computes for nn.Linear at models/transformer.py:15,17,36,38).  G bf16 [R, M], X bf16 [R, N], both
 * contiguous; C f32 [M, N].  M, N multiples of 256, any R >= 1 (rows past the last whole 64-row tile are masked inside the kernel).  The k range is split over workgroups; the caller
 * provides the scratch for the partial tiles: vitk_gemm_tn_workspace() floats (0 when no split is needed,
 * -1 for an unsupported shape).  Partials are added in a fixed order: results are reproducible. */
int64_t vitk_gemm_tn_workspace(int M, int N, int R);
int vitk_gemm_tn(const void* G, const void* X, float* C, float* workspace, int64_t workspace_floats, int M, int N,
                 int R, void* stream);

/* vitk_gemm_tn that also returns the layer's bias gradient from the same pass over G: colsum f32 [M] =
 * sum_r G[r][m] (what autograd computes for the bias of nn.Linear at models/transformer.py:15,17,36,38), with f32
 * accumulation on the matrix pipe and partial vectors added in a fixed order.  Workspace:
 * vitk_gemm_tn_colsum_workspace() floats (the partial tiles, then the partial vectors). */
int64_t vitk_gemm_tn_colsum_workspace(int M, int N, int R);
int vitk_gemm_tn_colsum(const void* G, const void* X, float* C, float* colsum, float* workspace, int64_t workspace_floats,
                        int M, int N, int R, void* stream);

/* vitk_attention for the training step: additionally writes lse f32 [batch, heads, ntok], the log2 of each query
 * row's softmax denominator in units of scores * scale * log2(e), which vitk_attention_backward needs. */
int vitk_attention_lse(const void* q, const void* k, const void* v, void* ctx, float* lse, int batch, int heads,
                       int ntok, int hd, float scale, void* stream);

/* Backward of vitk_attention (the reference differentiates models/transformer.py:43-46 by autograd): from q, k, v
 * (bf16 [batch, heads, ntok, hd]), the forward's ctx and lse, and dctx (bf16 [batch, ntok, heads*hd]) to
 * dqkv bf16 [batch, ntok, 3, heads, hd] = the gradient of the fused projection's output as a [rows, 3 D] matrix.
 * dbuf: caller's scratch, f32 [batch, heads, ntok] (rowsum(dctx * ctx) per query: written and read above 288 tokens,
 * where the gradient takes two kernels; untouched below, where one kernel keeps it in LDS).  hd = 64 with ntok <= 608,
 * or hd = 80 / 96 with ntok <= 288 (always the two kernels). */
int vitk_attention_backward(const void* q, const void* k, const void* v, const void* ctx, const void* dctx,
                            const float* lse, float* dbuf, void* dqkv, int batch, int heads, int ntok, int hd,
                            float scale, void* stream);

/* vitk_gemm_bias_gelu for the training step: also writes the pre-activation pre bf16 [M,N] = A W^T + bias, which
 * GELU's backward needs.  N a multiple of 256, K of 64, M >= 128 (other shapes: VITK_ERR_UNSUPPORTED). */
int vitk_gemm_bias_gelu_dual(const void* A, const void* W, const float* bias, void* pre, void* out, int M, int N, int K,
                             void* stream);

/* Input gradient of a Linear layer whose output went through an activation, in the training step (autograd of
 * models/transformer.py:21-24 and models/vision_transformer.py:53-54):
 *   out bf16 [M,N] = bf16(G Wt^T) * f'(aux)
 * G bf16 [M,K] = the gradient arriving at the NEXT layer's output, Wt bf16 [N,K] = that layer's weight transposed
 * (so G Wt^T = G W, the gradient of the next layer's input = this activation's output), aux bf16 [M,N]:
 * act = VITK_ACT_GELU: aux is the pre-activation (vitk_gemm_bias_gelu_dual's `pre`), f' = exact-erf GELU';
 * act = VITK_ACT_TANH: aux is the tanh OUTPUT, f' = 1 - aux^2.  K % 32 == 0.
 * Input gradients without an activation are vitk_gemm_bias(G, Wt, NULL, ...). */
#define VITK_ACT_GELU 0
#define VITK_ACT_TANH 1
int vitk_gemm_grad_act(const void* G, const void* Wt, const void* aux, void* out, int M, int N, int K, int act,
                       void* stream);

/* Bias gradient of a Linear layer in the training step: out[m] = sum_r G[r][m] with f32 accumulation; G bf16
 * [R, M] contiguous, M a multiple of 256.  partials: caller's scratch, f32 [partial_rows, M]; up to
 * min(partial_rows, 256) row chunks are summed separately and added in a fixed order. */
int vitk_colsum_bf16(const void* G, float* out, float* partials, int partial_rows, int R, int M, void* stream);

/* Diagnostic: the tile plan the tiled GEMM would use for out[M,N] = A[M,K] W[N,K]^T on a device with `cus` compute
 * units (0 = the current device's, 256 without a device), without launching anything.  plan[10] = {m-tiles per
 * tile, tiles along M, tiles along N, n-group width of the tile order (0 = none), two-height plan in use (0/1),
 * m-tiles per XCD, ha, na, hb, nb} (two-height plan: an XCD's rows are cut into na panels of ha m-tiles followed
 * by nb panels of hb; all 0 otherwise).  An m-tile is 16 rows.  No counterpart in the reference: the planner is
 * host logic of this library, exposed so that it can be tested without a GPU. */
int vitk_gemm_plan(int M, int N, int K, int cus, int* plan);

/* ---- the last block evaluated for the cls token only --------------------------------------------------
 * Encoder.forward keeps token 0 of the last block's output (models/transformer.py:98-99), and past the last block's
 * attention every op is per token: its out_proj, norm2, MLP and all but the cls query of its attention feed nothing.
 * These two entry points let a caller compute just that row per image with the arithmetic of the full launch. */

/* vitk_attention for the query rows below nrows only (whole 32-row tiles are computed; the other rows of ctx are
 * left untouched).  Tiling and summation order are those of vitk_attention: a computed row equals the full launch's. */
int vitk_attention_rows(const void* q, const void* k, const void* v, void* ctx, int batch, int heads, int ntok, int hd,
                        float scale, int nrows, void* stream);

/* vitk_gemm_bias_residual on strided rows: row m of A at A + m*lda, of resid and out at + m*ldo (elements; lda >= K a
 * multiple of 8, ldo >= N a multiple of 4).  With lda = ldo = ntok * D and M = batch it updates the cls rows of the
 * residual stream in place from the cls rows of ctx (out_proj, models/transformer.py:47,72) or of a [batch, M] tensor
 * (fc2, :24,76). */
int vitk_gemm_bias_residual_strided(const void* A, int64_t lda, const void* W, const float* bias, const float* resid,
                                    float* out, int64_t ldo, int M, int N, int K, void* stream);

/* ---- LayerNorm folded into the GEMMs either side of it (inference path) ----------------------------
 * Encoder1DBlock computes x = x + out_proj(..); y = norm2(x); h = gelu(fc1(y)) and x = x + fc2(h); y = norm1(x);
 * qkv = in_proj(y) (models/transformer.py:66-76, :40-48, :20-26).  With these four entry points the LayerNorm
 * between a residual GEMM and the projection that follows it needs no pass of its own over x:
 *   LN(x) W^T + b = rstd (x W'^T - mean s) + c,  W' = W * gamma (per input column), s[n] = sum_k W'[n][k],
 *   c[n] = b[n] + sum_k beta[k] W[n][k].
 * The residual GEMM's epilogue writes the statistics and a bf16 copy of the new rows, the projection runs on that
 * copy with W' and applies the two row factors in its epilogue.  Shapes: M >= 128, N a multiple of 256, K of 128,
 * every tensor below 4 GiB; anything else returns VITK_ERR_UNSUPPORTED and the caller keeps vitk_layernorm. */

/* vitk_gemm_bias_residual that also writes out_bf16 [M,N] = bf16(out) and stats f32 [M, N/64, 2]: per row and
 * 64-column strip the sum and the sum of squares of the new row's f32 values.  out may alias resid. */
int vitk_gemm_bias_residual_ln(const void* A, const void* W, const float* bias, const float* resid, float* out,
                               void* out_bf16, float* stats, int M, int N, int K, void* stream);

/* The same producer with the residual stream x kept as two 16-bit planes instead of f32 plus the bf16 copy (the
 * stream between the blocks of Encoder.forward, models/transformer.py:95-97; 4 bytes per element written per
 * residual GEMM instead of 6): plane_h[m][n] = upper half of (bits(x) + 0x8000), i.e. bf16(x) rounded half up, the
 * operand of the projection that follows as it lies; plane_l[m][n] = lower half of bits(x);
 * bits(x) = (plane_h << 16) + sext16(plane_l) restores the f32 value exactly.  Three forms, chosen by the pointers:
 *   resid_f32 != NULL:                 x read as f32, x + A W^T + bias written to the planes, statistics to stats
 *   resid_f32 == NULL, out_f32 == NULL: planes read and rewritten in place, statistics to stats
 *   out_f32 != NULL:                   planes read, x + A W^T + bias written as f32 (stats must be NULL)
 * The sums are formed in the order of vitk_gemm_bias_residual, so the stream's values are bit-identical to the f32
 * form's.  Same shape limits as vitk_gemm_bias_residual_ln. */
int vitk_gemm_bias_residual_planes(const void* A, const void* W, const float* bias, const float* resid_f32, void* plane_h,
                                   void* plane_l, float* out_f32, float* stats, int M, int N, int K, void* stream);

/* ab f32 [rows, 2] = (rstd, -mean * rstd) of every row from the strip statistics over D columns (strips added in
 * index order; biased variance, eps inside the root as nn.LayerNorm has it, models/transformer.py:56,60). */
int vitk_ln_finalize(const float* stats, float* ab, int rows, int D, float eps, void* stream);

/* out bf16 [M,N] = gelu(ab[m][0] * (A @ Wg^T)[m][n] + ab[m][1] * colsum[n] + c[n]): norm2 + fc1 + GELU
 * (models/transformer.py:73-74, :21-22) on A = bf16(x), Wg = bf16(W * gamma), colsum[n] = sum_k Wg[n][k]. */
int vitk_gemm_ln_bias_gelu(const void* A, const void* Wg, const float* ab, const float* colsum, const float* c, void* out,
                           int M, int N, int K, void* stream);

/* The same value for the fused q/k/v projection, scattered head-major like vitk_qkv_proj: norm1 + in_proj
 * (models/transformer.py:68-69, :42). */
int vitk_qkv_proj_ln(const void* A, const void* Wg, const float* ab, const float* colsum, const float* c, void* qkv,
                     int batch, int ntok, int heads, int hd, void* stream);

/* ---- for callers that cannot allocate ------------------------------------------------------------- */

/* Optional set-up for the current device: resolves the library's built-in scratch (used by
 * vitk_colsum_bf16_static) so that the first real call does no look-up.  Allocates nothing. */
int vitk_init(void);

/* vitk_colsum_bf16 without a workspace argument and without any allocation: the partial sums go to a 4 MiB
 * buffer built into the library's device code (one per GPU, held until the process ends).  It has no
 * counterpart in the reference; it exists for callers that cannot call hipMalloc or bring a workspace: code
 * under HIP graph capture, tight serving loops, harnesses that forbid allocation.  M a multiple of 256 and
 * <= 4096 (larger M: VITK_ERR_UNSUPPORTED).  Same kernels and summation order as vitk_colsum_bf16 with
 * partial_rows = 256, so the results are bit-identical.  All calls on one GPU share the buffer: calls that
 * overlap on one GPU (different streams) race; calls on one stream are ordered and safe. */
int vitk_colsum_bf16_static(const void* G, float* out, int R, int M, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H_ */
