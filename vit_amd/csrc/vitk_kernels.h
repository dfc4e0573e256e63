// Internal launch interface between the C ABI (vitk_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitk {

enum Epilogue {
    EPI_BIAS_BF16 = 0,       // out bf16 [M,N] = acc + bias
    EPI_BIAS_GELU_BF16 = 1,  // out bf16 [M,N] = gelu(acc + bias)
    EPI_BIAS_RESID_F32 = 2,  // out f32  [M,N] = resid + acc + bias
    EPI_QKV_BF16 = 3,        // out bf16 [3,B,H,Ntok,hd] <- (acc + bias)[b*Ntok+t, which*D + h*hd + e]
    EPI_PATCH_F32 = 4,       // out f32  [B,Ntok,N] row b*Ntok+1+p = acc + bias + pos[1+p]
    EPI_BIAS_F32 = 5,        // out f32  [M,N] = acc + bias
    EPI_BIAS_TANH_BF16 = 6,  // out bf16 [M,N] = tanh(acc + bias)
    EPI_BIAS_GELU_DUAL_BF16 = 7,  // training: out = gelu(acc + bias) and out2 bf16 [M,N] = acc + bias (the pre-activation)
    EPI_MUL_DGELU_BF16 = 8,  // training: out bf16 [M,N] = bf16(acc) * gelu'(aux), aux bf16 [M,N] = the pre-activation
    EPI_MUL_DTANH_BF16 = 9,  // training: out bf16 [M,N] = bf16(acc) * (1 - aux^2), aux bf16 [M,N] = the tanh output
    // LayerNorm folded into the GEMMs either side of it (inference).  Producer: the residual epilogue also writes
    // out2 = bf16(out) and, per row and 64-column strip, (sum, sum of squares) of the new row to ln_stats.
    // Consumers: A = bf16(x), W pre-multiplied by gamma; LN(x) W^T = rstd (x W'^T - mean s) + c with
    // s[n] = sum_k W'[n][k], c[n] = bias[n] + sum_k beta[k] W[n][k] (passed as `bias`), per-row (rstd, -mean rstd) in ln_ab.
    EPI_RESID_LN_F32 = 10,   // out f32 [M,N] = resid + acc + bias; out2 bf16 [M,N]; ln_stats f32 [M, N/64, 2]
    EPI_LN_GELU_BF16 = 11,   // out bf16 [M,N] = gelu(ln_ab[m][0] acc + ln_ab[m][1] ln_s[n] + bias[n])
    EPI_LN_QKV_BF16 = 12,    // the same value scattered head-major as EPI_QKV_BF16
    // The producer with the residual stream kept as two 16-bit planes instead of f32 plus a bf16 copy: plane H = the upper
    // half of (bits(x) + 0x8000), i.e. bf16(x) rounded half up, a GEMM operand as it lies; plane L = the lower half of
    // bits(x); bits(x) = (H << 16) + sext16(L) restores the f32 value exactly.  4 bytes per element written instead of 6.
    EPI_RESID_LN_FP = 13,    // resid f32 in; out = plane L, out2 = plane H of resid + acc + bias; ln_stats as above
    EPI_RESID_LN_PP = 14,    // planes in (aux = H, resid = L; may alias out2 / out), planes out, ln_stats
    EPI_RESID_PF = 15,       // planes in (aux = H, resid = L), out f32 [M,N] = x + acc + bias; no statistics
};

// epilogues of the fp32-mode GEMM (f32path.hip)
enum F32Epilogue { F32_BIAS = 0, F32_GELU = 1, F32_RESID = 2, F32_TANH = 3, F32_QKV = 4, F32_PATCH = 5 };

struct GemmArgs {
    const void* A;       // bf16 activations [M,K] (row stride lda), or the bf16 image for the gather loader
    const void* W;       // bf16 weights [N,K], row-major
    const float* bias;   // [N] or nullptr
    void* out;
    void* out2;          // EPI_BIAS_GELU_DUAL_BF16: the pre-activation
    const float* resid;  // EPI_BIAS_RESID_F32 (may alias out); EPI_RESID_LN_PP / EPI_RESID_PF: plane L, 16-bit [M,N]
    const float* pos;    // EPI_PATCH_F32: pos_embed [Ntok,N]
    const void* aux;     // EPI_MUL_*: bf16 [M,N] (row stride ldo), what the activation's derivative is taken of;
                         // EPI_RESID_LN_PP / EPI_RESID_PF: plane H
    float* ln_stats;     // EPI_RESID_LN_F32: f32 [M, N/64, 2]
    const float* ln_ab;  // EPI_LN_*: f32 [M, 2] = (rstd, -mean * rstd) of every row of x
    const float* ln_s;   // EPI_LN_*: f32 [N] column sums of the gamma-scaled bf16 weight
    int M, N, K;
    int lda, ldo;        // row strides in elements
    int ntok, heads, hd, batch;                 // EPI_QKV_BF16 / EPI_PATCH_F32
    int img_h, img_w, p_log2, grid_w, npatch;   // gather loader
    int order;           // workgroup -> tile order (low 4 bits: mode, bits 4..: n-group width)
};

// Every launcher returns hipSuccess or the launch error; shape checks are the caller's job.
hipError_t launch_gemm_tiled(const GemmArgs& a, int epilogue, bool gather, int variant, hipStream_t s);
hipError_t launch_gemm_small(const GemmArgs& a, int epilogue, hipStream_t s);
hipError_t launch_layernorm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta,
                            void* y, int64_t y_row_stride, int rows, int D, float eps, bool out_bf16,
                            hipStream_t s);
hipError_t launch_ln_finalize(const float* stats, float* ab, int rows, int D, float eps, hipStream_t s);
hipError_t launch_cls_pos(const float* cls, const float* pos, float* x, int batch, int ntok, int D,
                          hipStream_t s);
hipError_t launch_im2col(const void* img, void* cols, int batch, int img_h, int img_w, int patch, int kpad,
                         hipStream_t s);
hipError_t launch_layernorm_backward(const float* x, int64_t x_row_stride, const float* gamma, const void* g,
                                     bool g_bf16, const float* add, float* dx, void* dx16, float* dgamma, float* dbeta, float* partials,
                                     int partial_rows, int rows, int D, float eps, hipStream_t s);
int gemm_tn_splits(int M, int N, int R);
hipError_t launch_gemm_tn(const void* G, const void* X, float* C, float* colsum, float* part, int M, int N, int R,
                          int ldg, int ldx, hipStream_t s);
hipError_t launch_colsum_bf16(const void* G, float* out, float* partials, int partial_rows, int R, int M, hipStream_t s);
// the library's own partial-sum buffer for vitk_colsum_bf16_static: f32 [kColsumStaticRows, kColsumStaticMaxM]
// in device memory of the current device (resolved once per device); out may be nullptr (resolve only)
constexpr int kColsumStaticRows = 256, kColsumStaticMaxM = 4096;
hipError_t colsum_static_scratch(float** out);
hipError_t launch_normalize_u8(const void* in, void* out, int64_t n, int64_t plane, const float* mean,
                               const float* std, bool out_bf16, hipStream_t s);
hipError_t launch_target_rank(const float* logits, const int64_t* target, int* rank, int rows, int classes,
                              hipStream_t s);
hipError_t launch_pos_embed_scale(const float* in, float* out, int g, int G, int D, hipStream_t s);
hipError_t launch_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t s);
int attention_pick_waves(int tiles, int max_waves);
int attention_max_tokens(int hd);
hipError_t launch_attention(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                            int ntok, int hd, float scale, int variant, hipStream_t s, float* lse = nullptr, int qrows = 0);

int attention_bwd_max_tokens(int hd);
hipError_t launch_attention_backward(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                     const float* lse, float* dbuf, void* dqkv, int batch, int heads, int ntok, int hd,
                                     float scale, int variant, hipStream_t s);
hipError_t launch_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* out, int M,
                           int N, int K, int epilogue, hipStream_t s);
hipError_t launch_qkv_f32(const float* A, const float* W, const float* bias, float* qkv, int batch, int ntok,
                          int heads, int hd, hipStream_t s);
hipError_t launch_patch_f32(const float* img, const float* w, const float* bias, const float* pos, float* x,
                            int batch, int img_h, int img_w, int patch, int D, hipStream_t s);
hipError_t launch_attention_f32(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                int ntok, int hd, float scale, hipStream_t s);

}  // namespace vitk
