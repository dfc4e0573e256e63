// C ABI of the ViT forward kernels: argument checks, kernel selection, error
// reporting.  Declarations and the reference call sites each entry point
// replaces are in include/vitk.h.
#include <stdarg.h>
#include <atomic>
#include <stdio.h>

#include "../../include/vitk.h"
#include "gemm_plan.h"
#include "vitk_kernels.h"

using namespace vitk;

namespace vitk {
std::atomic<int> g_cu_limit{0};
}

namespace {

thread_local char g_err[256] = "";
// Test hook (vitk_set_debug_variant), one setting per process: torch runs autograd backward on its own worker
// threads, and a per-thread setting made on the main thread would never reach the kernels launched from there.
std::atomic<int> g_gemm_variant_a{0}, g_attn_variant_a{0};
#define g_gemm_variant (g_gemm_variant_a.load(std::memory_order_relaxed))
#define g_attn_variant (g_attn_variant_a.load(std::memory_order_relaxed))

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int launched(hipError_t e, const char* what) {
    if (e == hipSuccess) return VITK_OK;
    return fail(VITK_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Shared path of the four dense entry points.
int dense(const char* name, const void* A, const void* W, const float* bias, const float* resid, void* out,
          int M, int N, int K, int epilogue, void* stream) {
    if (!A || !W || !out) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(W) || !aligned16(out) || (bias && !aligned16(bias)) ||
        (resid && !aligned16(resid)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (K % 32 != 0) return fail(VITK_ERR_UNSUPPORTED, "%s: K=%d is not a multiple of 32", name, K);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.resid = resid;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    const bool tiled_ok = (N % 256 == 0) && (K % 64 == 0) && M >= 128 && (g_gemm_variant & 15) != 2;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (tiled_ok) return launched(launch_gemm_tiled(a, epilogue, false, g_gemm_variant & 15, s), name);
    return launched(launch_gemm_small(a, epilogue, s), name);
}

}  // namespace

extern "C" {

int vitk_version(void) { return 10; }

const char* vitk_last_error(void) { return g_err; }

void vitk_set_debug_variant(int gemm_variant, int attention_variant) {
    g_gemm_variant_a.store(gemm_variant, std::memory_order_relaxed);
    g_attn_variant_a.store(attention_variant, std::memory_order_relaxed);
}

void vitk_set_cu_limit(int cus) { vitk::g_cu_limit.store(cus > 0 ? cus : 0, std::memory_order_relaxed); }

int vitk_layernorm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y,
                   int64_t y_row_stride, int rows, int D, float eps, int out_dtype, void* stream) {
    if (!x || !gamma || !beta || !y) return fail(VITK_ERR_INVALID, "vitk_layernorm: null pointer");
    if (rows <= 0) return fail(VITK_ERR_INVALID, "vitk_layernorm: rows=%d", rows);
    if (out_dtype != VITK_F32 && out_dtype != VITK_BF16)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_layernorm: out_dtype=%d", out_dtype);
    if (D % 256 != 0 || D < 256 || D > 1280)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_layernorm: D=%d (supported: 256..1280 in steps of 256)", D);
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(y) || x_row_stride % 4 != 0 ||
        y_row_stride % 4 != 0 || x_row_stride < D || y_row_stride < D)
        return fail(VITK_ERR_INVALID, "vitk_layernorm: misaligned pointer or row stride");
    return launched(launch_layernorm(x, x_row_stride, gamma, beta, y, y_row_stride, rows, D, eps,
                                     out_dtype == VITK_BF16, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_layernorm");
}

int vitk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
    if (!in || !out) return fail(VITK_ERR_INVALID, "vitk_cast_f32_bf16: null pointer");
    if (n <= 0) return fail(VITK_ERR_INVALID, "vitk_cast_f32_bf16: n=%lld must be positive", (long long)n);
    if (!aligned16(in) || !aligned16(out)) return fail(VITK_ERR_INVALID, "vitk_cast_f32_bf16: misaligned pointer");
    return launched(launch_cast_f32_bf16(in, out, n, reinterpret_cast<hipStream_t>(stream)), "vitk_cast_f32_bf16");
}

int vitk_patch_embed(const void* img, const void* w, const float* bias, const float* cls, const float* pos,
                     float* x, int batch, int img_h, int img_w, int patch, int D, void* stream) {
    if (!img || !w || !bias || !cls || !pos || !x) return fail(VITK_ERR_INVALID, "vitk_patch_embed: null pointer");
    if (batch <= 0 || img_h <= 0 || img_w <= 0 || patch <= 0 || D <= 0)
        return fail(VITK_ERR_INVALID, "vitk_patch_embed: non-positive size");
    if (!aligned16(img) || !aligned16(w) || !aligned16(bias) || !aligned16(cls) || !aligned16(pos) || !aligned16(x))
        return fail(VITK_ERR_INVALID, "vitk_patch_embed: pointers must be 16-byte aligned");
    if (!is_pow2(patch) || patch < 8)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_patch_embed: patch=%d (needs a power of two >= 8)", patch);
    if (img_w % 8 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_patch_embed: img_w=%d not a multiple of 8", img_w);
    if (D % 256 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_patch_embed: D=%d not a multiple of 256", D);
    const int gh = img_h / patch, gw = img_w / patch;   // Conv2d without padding floors
    if (gh < 1 || gw < 1) return fail(VITK_ERR_INVALID, "vitk_patch_embed: image smaller than one patch");
    GemmArgs a = {};
    a.A = img; a.W = w; a.bias = bias; a.out = x; a.pos = pos;
    a.npatch = gh * gw; a.ntok = a.npatch + 1; a.grid_w = gw;
    a.M = batch * a.npatch; a.N = D; a.K = 3 * patch * patch;
    a.lda = 0; a.ldo = D; a.batch = batch;
    a.img_h = img_h; a.img_w = img_w; a.p_log2 = ilog2(patch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = launched(launch_cls_pos(cls, pos, x, batch, a.ntok, D, s), "vitk_patch_embed(cls)");
    if (rc) return rc;
    a.order = g_gemm_variant >> 4;
    const int v = g_gemm_variant & 15;
    return launched(launch_gemm_tiled(a, EPI_PATCH_F32, true, v == 2 ? 0 : v, s), "vitk_patch_embed");
}

int vitk_patch_embed_cols(const void* img, const void* w_padded, const float* bias, const float* cls,
                          const float* pos, float* x, void* cols, int batch, int img_h, int img_w, int patch, int D,
                          void* stream) {
    if (!img || !w_padded || !bias || !cls || !pos || !x || !cols)
        return fail(VITK_ERR_INVALID, "vitk_patch_embed_cols: null pointer");
    if (batch <= 0 || img_h <= 0 || img_w <= 0 || patch <= 0 || D <= 0)
        return fail(VITK_ERR_INVALID, "vitk_patch_embed_cols: non-positive size");
    if (!aligned16(w_padded) || !aligned16(bias) || !aligned16(cls) || !aligned16(pos) || !aligned16(x) || !aligned16(cols))
        return fail(VITK_ERR_INVALID, "vitk_patch_embed_cols: pointers must be 16-byte aligned");
    if (D % 256 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_patch_embed_cols: D=%d not a multiple of 256", D);
    const int gh = img_h / patch, gw = img_w / patch;   // Conv2d without padding floors
    if (gh < 1 || gw < 1) return fail(VITK_ERR_INVALID, "vitk_patch_embed_cols: image smaller than one patch");
    const int kpad = (3 * patch * patch + 63) / 64 * 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = launched(launch_im2col(img, cols, batch, img_h, img_w, patch, kpad, s), "vitk_patch_embed_cols(im2col)");
    if (rc) return rc;
    GemmArgs a = {};
    a.A = cols; a.W = w_padded; a.bias = bias; a.out = x; a.pos = pos;
    a.npatch = gh * gw; a.ntok = a.npatch + 1; a.grid_w = gw;
    a.M = batch * a.npatch; a.N = D; a.K = kpad;
    a.lda = kpad; a.ldo = D; a.batch = batch;
    rc = launched(launch_cls_pos(cls, pos, x, batch, a.ntok, D, s), "vitk_patch_embed_cols(cls)");
    if (rc) return rc;
    a.order = g_gemm_variant >> 4;
    const int v = g_gemm_variant & 15;
    return launched(launch_gemm_tiled(a, EPI_PATCH_F32, false, v == 2 ? 0 : v, s), "vitk_patch_embed_cols");
}

int vitk_gemm_bias(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                   int out_dtype, void* stream) {
    if (out_dtype != VITK_F32 && out_dtype != VITK_BF16)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_gemm_bias: out_dtype=%d", out_dtype);
    return dense("vitk_gemm_bias", A, W, bias, nullptr, out, M, N, K,
                 out_dtype == VITK_F32 ? EPI_BIAS_F32 : EPI_BIAS_BF16, stream);
}

int vitk_gemm_bias_gelu(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                        void* stream) {
    return dense("vitk_gemm_bias_gelu", A, W, bias, nullptr, out, M, N, K, EPI_BIAS_GELU_BF16, stream);
}

int vitk_gemm_bias_tanh(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                        void* stream) {
    return dense("vitk_gemm_bias_tanh", A, W, bias, nullptr, out, M, N, K, EPI_BIAS_TANH_BF16, stream);
}

int vitk_gemm_bias_residual(const void* A, const void* W, const float* bias, const float* resid, float* out,
                            int M, int N, int K, void* stream) {
    if (!resid) return fail(VITK_ERR_INVALID, "vitk_gemm_bias_residual: null residual");
    return dense("vitk_gemm_bias_residual", A, W, bias, resid, out, M, N, K, EPI_BIAS_RESID_F32, stream);
}

int vitk_qkv_proj(const void* A, const void* W, const float* bias, void* qkv, int batch, int ntok, int heads,
                  int hd, void* stream) {
    if (!A || !W || !qkv) return fail(VITK_ERR_INVALID, "vitk_qkv_proj: null pointer");
    if (batch <= 0 || ntok <= 0 || heads <= 0 || hd <= 0)
        return fail(VITK_ERR_INVALID, "vitk_qkv_proj: non-positive size");
    if (!aligned16(A) || !aligned16(W) || !aligned16(qkv) || (bias && !aligned16(bias)))
        return fail(VITK_ERR_INVALID, "vitk_qkv_proj: pointers must be 16-byte aligned");
    const int D = heads * hd;
    if (hd % 8 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_qkv_proj: hd=%d not a multiple of 8", hd);
    if ((3 * D) % 256 != 0 || D % 64 != 0)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_qkv_proj: D=%d (needs 3*D %% 256 == 0 and D %% 64 == 0)", D);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = qkv;
    a.M = batch * ntok; a.N = 3 * D; a.K = D; a.lda = D; a.ldo = 0;
    a.ntok = ntok; a.heads = heads; a.hd = hd; a.batch = batch;
    a.order = g_gemm_variant >> 4;
    const int v = g_gemm_variant & 15;
    return launched(launch_gemm_tiled(a, EPI_QKV_BF16, false, v == 2 ? 0 : v, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_qkv_proj");
}

// ---- LayerNorm folded into the GEMMs either side of it (inference path) ------------------------

static bool ln_fold_shape_ok(int M, int N, int K) {
    return M >= 128 && N % 256 == 0 && K % 128 == 0 && K >= 128;
}

int vitk_gemm_bias_residual_ln(const void* A, const void* W, const float* bias, const float* resid, float* out,
                               void* out_bf16, float* stats, int M, int N, int K, void* stream) {
    const char* name = "vitk_gemm_bias_residual_ln";
    if (!A || !W || !resid || !out || !out_bf16 || !stats) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(W) || !aligned16(out) || !aligned16(resid) || !aligned16(out_bf16) || !aligned16(stats) ||
        (bias && !aligned16(bias)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (!ln_fold_shape_ok(M, N, K))
        return fail(VITK_ERR_UNSUPPORTED, "%s: M=%d N=%d K=%d (M >= 128, N a multiple of 256, K of 128)", name, M, N, K);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.resid = resid; a.out2 = out_bf16; a.ln_stats = stats;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    hipError_t e = launch_gemm_tiled(a, EPI_RESID_LN_F32, false, 0, reinterpret_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(VITK_ERR_UNSUPPORTED, "%s: a tensor of 4 GiB or more", name);
    return launched(e, name);
}

int vitk_gemm_bias_residual_planes(const void* A, const void* W, const float* bias, const float* resid_f32, void* plane_h,
                                   void* plane_l, float* out_f32, float* stats, int M, int N, int K, void* stream) {
    const char* name = "vitk_gemm_bias_residual_planes";
    if (!A || !W || !plane_h || !plane_l) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (resid_f32 && out_f32) return fail(VITK_ERR_INVALID, "%s: f32 stream in and out (use vitk_gemm_bias_residual_ln)", name);
    if ((out_f32 != nullptr) == (stats != nullptr))
        return fail(VITK_ERR_INVALID, "%s: statistics go with the planes written, none with the f32 stream written", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(W) || !aligned16(plane_h) || !aligned16(plane_l) || (bias && !aligned16(bias)) ||
        (resid_f32 && !aligned16(resid_f32)) || (out_f32 && !aligned16(out_f32)) || (stats && !aligned16(stats)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (!ln_fold_shape_ok(M, N, K))
        return fail(VITK_ERR_UNSUPPORTED, "%s: M=%d N=%d K=%d (M >= 128, N a multiple of 256, K of 128)", name, M, N, K);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.ln_stats = stats;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    int epi;
    if (resid_f32) {                 // f32 stream in, planes out
        epi = EPI_RESID_LN_FP;
        a.resid = resid_f32; a.out = plane_l; a.out2 = plane_h;
    } else if (!out_f32) {           // planes in, planes out (in place)
        epi = EPI_RESID_LN_PP;
        a.resid = reinterpret_cast<const float*>(plane_l); a.aux = plane_h; a.out = plane_l; a.out2 = plane_h;
    } else {                         // planes in, f32 stream out
        epi = EPI_RESID_PF;
        a.resid = reinterpret_cast<const float*>(plane_l); a.aux = plane_h; a.out = out_f32;
    }
    hipError_t e = launch_gemm_tiled(a, epi, false, 0, reinterpret_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(VITK_ERR_UNSUPPORTED, "%s: a tensor of 4 GiB or more", name);
    return launched(e, name);
}

int vitk_ln_finalize(const float* stats, float* ab, int rows, int D, float eps, void* stream) {
    if (!stats || !ab) return fail(VITK_ERR_INVALID, "vitk_ln_finalize: null pointer");
    if (rows <= 0 || D <= 0 || D % 128 != 0) return fail(VITK_ERR_INVALID, "vitk_ln_finalize: rows=%d D=%d (D a multiple of 128)", rows, D);
    if (!aligned16(stats) || !aligned16(ab)) return fail(VITK_ERR_INVALID, "vitk_ln_finalize: pointers must be 16-byte aligned");
    return launched(launch_ln_finalize(stats, ab, rows, D, eps, reinterpret_cast<hipStream_t>(stream)), "vitk_ln_finalize");
}

int vitk_gemm_ln_bias_gelu(const void* A, const void* Wg, const float* ab, const float* colsum, const float* c, void* out,
                           int M, int N, int K, void* stream) {
    const char* name = "vitk_gemm_ln_bias_gelu";
    if (!A || !Wg || !ab || !colsum || !c || !out) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(Wg) || !aligned16(ab) || !aligned16(colsum) || !aligned16(c) || !aligned16(out))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (!ln_fold_shape_ok(M, N, K))
        return fail(VITK_ERR_UNSUPPORTED, "%s: M=%d N=%d K=%d (M >= 128, N a multiple of 256, K of 128)", name, M, N, K);
    GemmArgs a = {};
    a.A = A; a.W = Wg; a.bias = c; a.out = out; a.ln_ab = ab; a.ln_s = colsum;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    hipError_t e = launch_gemm_tiled(a, EPI_LN_GELU_BF16, false, 0, reinterpret_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(VITK_ERR_UNSUPPORTED, "%s: a tensor of 4 GiB or more", name);
    return launched(e, name);
}

int vitk_qkv_proj_ln(const void* A, const void* Wg, const float* ab, const float* colsum, const float* c, void* qkv,
                     int batch, int ntok, int heads, int hd, void* stream) {
    const char* name = "vitk_qkv_proj_ln";
    if (!A || !Wg || !ab || !colsum || !c || !qkv) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (batch <= 0 || ntok <= 0 || heads <= 0 || hd <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size", name);
    if (!aligned16(A) || !aligned16(Wg) || !aligned16(ab) || !aligned16(colsum) || !aligned16(c) || !aligned16(qkv))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    const int D = heads * hd;
    if (hd % 8 != 0) return fail(VITK_ERR_UNSUPPORTED, "%s: hd=%d not a multiple of 8", name, hd);
    if (!ln_fold_shape_ok(batch * ntok, 3 * D, D))
        return fail(VITK_ERR_UNSUPPORTED, "%s: rows=%d D=%d (rows >= 128, 3 D a multiple of 256, D of 128)", name, batch * ntok, D);
    GemmArgs a = {};
    a.A = A; a.W = Wg; a.bias = c; a.out = qkv; a.ln_ab = ab; a.ln_s = colsum;
    a.M = batch * ntok; a.N = 3 * D; a.K = D; a.lda = D; a.ldo = 0;
    a.ntok = ntok; a.heads = heads; a.hd = hd; a.batch = batch;
    a.order = g_gemm_variant >> 4;
    hipError_t e = launch_gemm_tiled(a, EPI_LN_QKV_BF16, false, 0, reinterpret_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(VITK_ERR_UNSUPPORTED, "%s: a tensor of 4 GiB or more", name);
    return launched(e, name);
}

static int attention_forward(const char* name, const void* q, const void* k, const void* v, void* ctx, float* lse,
                             int batch, int heads, int ntok, int hd, float scale, void* stream, int qrows = 0) {
    if (!q || !k || !v || !ctx) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (batch <= 0 || heads <= 0 || ntok <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size", name);
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(ctx))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (hd != 64 && hd != 80 && hd != 96)
        return fail(VITK_ERR_UNSUPPORTED, "%s: hd=%d (64, 80 and 96 are implemented)", name, hd);
    if (ntok > attention_max_tokens(hd))
        return fail(VITK_ERR_UNSUPPORTED, "%s: ntok=%d exceeds the limit %d at hd=%d", name, ntok,
                    attention_max_tokens(hd), hd);
    return launched(launch_attention(q, k, v, ctx, batch, heads, ntok, hd, scale, g_attn_variant,
                                     reinterpret_cast<hipStream_t>(stream), lse, qrows),
                    name);
}

int vitk_attention_rows(const void* q, const void* k, const void* v, void* ctx, int batch, int heads, int ntok, int hd,
                        float scale, int nrows, void* stream) {
    if (nrows < 1 || nrows > ntok) return fail(VITK_ERR_INVALID, "vitk_attention_rows: nrows=%d outside 1..%d", nrows, ntok);
    return attention_forward("vitk_attention_rows", q, k, v, ctx, nullptr, batch, heads, ntok, hd, scale, stream, nrows);
}

int vitk_gemm_bias_residual_strided(const void* A, int64_t lda, const void* W, const float* bias, const float* resid,
                                    float* out, int64_t ldo, int M, int N, int K, void* stream) {
    const char* name = "vitk_gemm_bias_residual_strided";
    if (!A || !W || !resid || !out) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(W) || !aligned16(out) || !aligned16(resid) || (bias && !aligned16(bias)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (lda < K || ldo < N || lda % 8 != 0 || ldo % 4 != 0)
        return fail(VITK_ERR_INVALID, "%s: lda=%lld ldo=%lld (at least K / N, multiples of 8 / 4 elements)", name, (long long)lda, (long long)ldo);
    if (K % 32 != 0) return fail(VITK_ERR_UNSUPPORTED, "%s: K=%d is not a multiple of 32", name, K);
    if ((int64_t)(M + 256) * ldo * 4 >= ((int64_t)1 << 32) || (int64_t)(M + 256) * lda * 2 >= ((int64_t)1 << 32))
        return fail(VITK_ERR_UNSUPPORTED, "%s: rows x stride of 4 GiB or more", name);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.resid = resid;
    a.M = M; a.N = N; a.K = K; a.lda = (int)lda; a.ldo = (int)ldo;
    a.order = g_gemm_variant >> 4;
    const bool tiled_ok = (N % 256 == 0) && (K % 64 == 0) && M >= 128 && (g_gemm_variant & 15) != 2;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (tiled_ok) return launched(launch_gemm_tiled(a, EPI_BIAS_RESID_F32, false, g_gemm_variant & 15, s), name);
    return launched(launch_gemm_small(a, EPI_BIAS_RESID_F32, s), name);
}

int vitk_attention(const void* q, const void* k, const void* v, void* ctx, int batch, int heads, int ntok,
                   int hd, float scale, void* stream) {
    return attention_forward("vitk_attention", q, k, v, ctx, nullptr, batch, heads, ntok, hd, scale, stream);
}

int vitk_attention_lse(const void* q, const void* k, const void* v, void* ctx, float* lse, int batch, int heads,
                       int ntok, int hd, float scale, void* stream) {
    if (!lse) return fail(VITK_ERR_INVALID, "vitk_attention_lse: null pointer");
    return attention_forward("vitk_attention_lse", q, k, v, ctx, lse, batch, heads, ntok, hd, scale, stream);
}

int vitk_attention_backward(const void* q, const void* k, const void* v, const void* ctx, const void* dctx,
                            const float* lse, float* dbuf, void* dqkv, int batch, int heads, int ntok, int hd,
                            float scale, void* stream) {
    if (!q || !k || !v || !ctx || !dctx || !lse || !dbuf || !dqkv)
        return fail(VITK_ERR_INVALID, "vitk_attention_backward: null pointer");
    if (batch <= 0 || heads <= 0 || ntok <= 0) return fail(VITK_ERR_INVALID, "vitk_attention_backward: non-positive size");
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(ctx) || !aligned16(dctx) || !aligned16(dqkv))
        return fail(VITK_ERR_INVALID, "vitk_attention_backward: pointers must be 16-byte aligned");
    if (hd != 64 && hd != 80 && hd != 96)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_attention_backward: hd=%d (64, 80 and 96 are implemented)", hd);
    if (ntok > attention_bwd_max_tokens(hd))
        return fail(VITK_ERR_UNSUPPORTED, "vitk_attention_backward: ntok=%d exceeds the limit %d at hd=%d", ntok,
                    attention_bwd_max_tokens(hd), hd);
    return launched(launch_attention_backward(q, k, v, ctx, dctx, lse, dbuf, dqkv, batch, heads, ntok, hd, scale,
                                              g_attn_variant, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_attention_backward");
}

// ---- fp32 compute mode ------------------------------------------------------------------------

int vitk_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N,
                  int K, int epilogue, void* stream) {
    if (!A || !W || !out) return fail(VITK_ERR_INVALID, "vitk_gemm_f32: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "vitk_gemm_f32: non-positive size %dx%dx%d", M, N, K);
    if (epilogue < VITK_EPI_BIAS || epilogue > VITK_EPI_TANH)
        return fail(VITK_ERR_INVALID, "vitk_gemm_f32: epilogue=%d", epilogue);
    if (epilogue == VITK_EPI_RESIDUAL && !resid) return fail(VITK_ERR_INVALID, "vitk_gemm_f32: null residual");
    return launched(launch_gemm_f32(A, W, bias, resid, out, M, N, K, epilogue, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_gemm_f32");
}

int vitk_qkv_proj_f32(const float* A, const float* W, const float* bias, float* qkv, int batch, int ntok, int heads,
                      int hd, void* stream) {
    if (!A || !W || !qkv) return fail(VITK_ERR_INVALID, "vitk_qkv_proj_f32: null pointer");
    if (batch <= 0 || ntok <= 0 || heads <= 0 || hd <= 0)
        return fail(VITK_ERR_INVALID, "vitk_qkv_proj_f32: non-positive size");
    return launched(launch_qkv_f32(A, W, bias, qkv, batch, ntok, heads, hd, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_qkv_proj_f32");
}

int vitk_patch_embed_f32(const float* img, const float* w, const float* bias, const float* cls, const float* pos,
                         float* x, int batch, int img_h, int img_w, int patch, int D, void* stream) {
    if (!img || !w || !bias || !cls || !pos || !x) return fail(VITK_ERR_INVALID, "vitk_patch_embed_f32: null pointer");
    if (batch <= 0 || img_h <= 0 || img_w <= 0 || patch <= 0 || D <= 0)
        return fail(VITK_ERR_INVALID, "vitk_patch_embed_f32: non-positive size");
    if (!aligned16(cls) || !aligned16(pos) || !aligned16(x) || D % 4 != 0)
        return fail(VITK_ERR_INVALID, "vitk_patch_embed_f32: cls/pos/x must be 16-byte aligned and D a multiple of 4");
    const int gh = img_h / patch, gw = img_w / patch;   // Conv2d without padding floors
    if (gh < 1 || gw < 1) return fail(VITK_ERR_INVALID, "vitk_patch_embed_f32: image smaller than one patch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = launched(launch_cls_pos(cls, pos, x, batch, gh * gw + 1, D, s), "vitk_patch_embed_f32(cls)");
    if (rc) return rc;
    return launched(launch_patch_f32(img, w, bias, pos, x, batch, img_h, img_w, patch, D, s), "vitk_patch_embed_f32");
}

int vitk_attention_f32(const float* q, const float* k, const float* v, float* ctx, int batch, int heads, int ntok,
                       int hd, float scale, void* stream) {
    if (!q || !k || !v || !ctx) return fail(VITK_ERR_INVALID, "vitk_attention_f32: null pointer");
    if (batch <= 0 || heads <= 0 || ntok <= 0) return fail(VITK_ERR_INVALID, "vitk_attention_f32: non-positive size");
    if (hd != 16 && hd != 32 && hd != 64 && hd != 80 && hd != 96 && hd != 128)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_attention_f32: hd=%d (16, 32, 64, 80, 96 and 128 are implemented)", hd);
    if (ntok > 1024) return fail(VITK_ERR_UNSUPPORTED, "vitk_attention_f32: ntok=%d exceeds the LDS-resident limit 1024", ntok);
    return launched(launch_attention_f32(q, k, v, ctx, batch, heads, ntok, hd, scale,
                                         reinterpret_cast<hipStream_t>(stream)), "vitk_attention_f32");
}

int vitk_pos_embed_scale(const float* pos_in, float* pos_out, int grid_in, int grid_out, int D, void* stream) {
    if (!pos_in || !pos_out) return fail(VITK_ERR_INVALID, "vitk_pos_embed_scale: null pointer");
    if (grid_in < 1 || grid_out < 1 || D < 1) return fail(VITK_ERR_INVALID, "vitk_pos_embed_scale: non-positive size");
    if (pos_in == pos_out) return fail(VITK_ERR_INVALID, "vitk_pos_embed_scale: in place is not supported");
    return launched(launch_pos_embed_scale(pos_in, pos_out, grid_in, grid_out, D, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_pos_embed_scale");
}

int vitk_normalize_u8(const void* img, void* out, int batch, int img_h, int img_w, const float* mean,
                      const float* std, int out_dtype, void* stream) {
    if (!img || !out || !mean || !std) return fail(VITK_ERR_INVALID, "vitk_normalize_u8: null pointer");
    if (batch <= 0 || img_h <= 0 || img_w <= 0) return fail(VITK_ERR_INVALID, "vitk_normalize_u8: non-positive size");
    if (out_dtype != VITK_F32 && out_dtype != VITK_BF16)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_normalize_u8: out_dtype=%d", out_dtype);
    if (!aligned16(img) || !aligned16(out)) return fail(VITK_ERR_INVALID, "vitk_normalize_u8: pointers must be 16-byte aligned");
    const int64_t plane = (int64_t)img_h * img_w;
    for (int c = 0; c < 3; ++c)
        if (!(std[c] > 0.f)) return fail(VITK_ERR_INVALID, "vitk_normalize_u8: std[%d] must be positive", c);
    return launched(launch_normalize_u8(img, out, (int64_t)batch * 3 * plane, plane, mean, std, out_dtype == VITK_BF16,
                                        reinterpret_cast<hipStream_t>(stream)), "vitk_normalize_u8");
}

int vitk_target_rank(const float* logits, const int64_t* target, int* rank, int rows, int classes, void* stream) {
    if (!logits || !target || !rank) return fail(VITK_ERR_INVALID, "vitk_target_rank: null pointer");
    if (rows <= 0 || classes <= 0) return fail(VITK_ERR_INVALID, "vitk_target_rank: non-positive size");
    return launched(launch_target_rank(logits, target, rank, rows, classes, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_target_rank");
}

int vitk_layernorm_backward(const float* x, int64_t x_row_stride, const float* gamma, const void* g, int g_dtype,
                            const float* add, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* partials, int partial_rows, int rows,
                            int D, float eps, void* stream) {
    if (!x || !gamma || !g || !dx || !dgamma || !dbeta || !partials)
        return fail(VITK_ERR_INVALID, "vitk_layernorm_backward: null pointer");
    if (rows <= 0 || partial_rows <= 0)
        return fail(VITK_ERR_INVALID, "vitk_layernorm_backward: rows=%d partial_rows=%d", rows, partial_rows);
    if (g_dtype != VITK_F32 && g_dtype != VITK_BF16)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_layernorm_backward: g_dtype=%d", g_dtype);
    if (D % 256 != 0 || D < 256 || D > 1280)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_layernorm_backward: D=%d (supported: 256..1280 in steps of 256)", D);
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(g) || !aligned16(dx) || !aligned16(partials) ||
        (add && !aligned16(add)) || (dx_bf16 && !aligned16(dx_bf16)) || x_row_stride % 4 != 0 || x_row_stride < D)
        return fail(VITK_ERR_INVALID, "vitk_layernorm_backward: misaligned pointer or row stride");
    return launched(launch_layernorm_backward(x, x_row_stride, gamma, g, g_dtype == VITK_BF16, add, dx, dx_bf16, dgamma, dbeta, partials,
                                              partial_rows, rows, D, eps, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_layernorm_backward");
}

int64_t vitk_gemm_tn_workspace(int M, int N, int R) {
    if (M <= 0 || N <= 0 || R <= 0 || M % 256 || N % 256) return -1;
    const int splits = gemm_tn_splits(M, N, R);
    return splits <= 1 ? 0 : (int64_t)splits * M * N;
}

static int gemm_tn_checked(const char* name, const void* G, const void* X, float* C, float* colsum, float* workspace,
                           int64_t workspace_floats, int64_t need, int M, int N, int R, void* stream) {
    if (!G || !X || !C) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || R <= 0 || M % 256 || N % 256)
        return fail(VITK_ERR_UNSUPPORTED, "%s: M=%d N=%d R=%d (M, N multiples of 256)", name, M, N, R);
    if ((int64_t)R * (M > N ? M : N) * 2 >= ((int64_t)1 << 32))
        return fail(VITK_ERR_UNSUPPORTED, "%s: operand of 4 GiB or more", name);
    if (need > 0 && (!workspace || workspace_floats < need))
        return fail(VITK_ERR_INVALID, "%s: workspace of %lld floats needed", name, (long long)need);
    if (!aligned16(G) || !aligned16(X) || !aligned16(C) || (workspace && !aligned16(workspace)) || (colsum && !aligned16(colsum)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    return launched(launch_gemm_tn(G, X, C, colsum, workspace, M, N, R, M, N, reinterpret_cast<hipStream_t>(stream)), name);
}

int vitk_gemm_tn(const void* G, const void* X, float* C, float* workspace, int64_t workspace_floats, int M, int N,
                 int R, void* stream) {
    return gemm_tn_checked("vitk_gemm_tn", G, X, C, nullptr, workspace, workspace_floats, vitk_gemm_tn_workspace(M, N, R), M, N, R,
                           stream);
}

int64_t vitk_gemm_tn_colsum_workspace(int M, int N, int R) {
    if (M <= 0 || N <= 0 || R <= 0 || M % 256 || N % 256) return -1;
    const int splits = gemm_tn_splits(M, N, R);
    return splits <= 1 ? 0 : (int64_t)splits * M * N + (int64_t)splits * M;
}

int vitk_gemm_tn_colsum(const void* G, const void* X, float* C, float* colsum, float* workspace, int64_t workspace_floats,
                        int M, int N, int R, void* stream) {
    if (!colsum) return fail(VITK_ERR_INVALID, "vitk_gemm_tn_colsum: null pointer");
    return gemm_tn_checked("vitk_gemm_tn_colsum", G, X, C, colsum, workspace, workspace_floats,
                           vitk_gemm_tn_colsum_workspace(M, N, R), M, N, R, stream);
}

int vitk_gemm_bias_gelu_dual(const void* A, const void* W, const float* bias, void* pre, void* out, int M, int N, int K,
                             void* stream) {
    const char* name = "vitk_gemm_bias_gelu_dual";
    if (!A || !W || !pre || !out) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (!aligned16(A) || !aligned16(W) || !aligned16(pre) || !aligned16(out) || (bias && !aligned16(bias)))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (N % 256 != 0 || K % 64 != 0 || M < 128)
        return fail(VITK_ERR_UNSUPPORTED, "%s: M=%d N=%d K=%d (N a multiple of 256, K of 64, M >= 128)", name, M, N, K);
    GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.out2 = pre;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    hipError_t e = launch_gemm_tiled(a, EPI_BIAS_GELU_DUAL_BF16, false, g_gemm_variant & 15, reinterpret_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(VITK_ERR_UNSUPPORTED, "%s: not available with this kernel variant / size", name);
    return launched(e, name);
}

int vitk_gemm_grad_act(const void* G, const void* Wt, const void* aux, void* out, int M, int N, int K, int act,
                       void* stream) {
    const char* name = "vitk_gemm_grad_act";
    if (!G || !Wt || !aux || !out) return fail(VITK_ERR_INVALID, "%s: null pointer", name);
    if (M <= 0 || N <= 0 || K <= 0) return fail(VITK_ERR_INVALID, "%s: non-positive size %dx%dx%d", name, M, N, K);
    if (act != VITK_ACT_GELU && act != VITK_ACT_TANH) return fail(VITK_ERR_INVALID, "%s: act=%d", name, act);
    if (!aligned16(G) || !aligned16(Wt) || !aligned16(aux) || !aligned16(out))
        return fail(VITK_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
    if (K % 32 != 0) return fail(VITK_ERR_UNSUPPORTED, "%s: K=%d is not a multiple of 32", name, K);
    GemmArgs a = {};
    a.A = G; a.W = Wt; a.aux = aux; a.out = out;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldo = N;
    a.order = g_gemm_variant >> 4;
    const int epi = act == VITK_ACT_GELU ? EPI_MUL_DGELU_BF16 : EPI_MUL_DTANH_BF16;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool tiled_ok = (N % 256 == 0) && (K % 64 == 0) && M >= 128 && (g_gemm_variant & 15) != 2;
    if (tiled_ok) {
        hipError_t e = launch_gemm_tiled(a, epi, false, g_gemm_variant & 15, s);
        if (e != hipErrorInvalidValue) return launched(e, name);      // else: a tensor of 4 GiB or more
    }
    return launched(launch_gemm_small(a, epi, s), name);
}

int vitk_colsum_bf16(const void* G, float* out, float* partials, int partial_rows, int R, int M, void* stream) {
    if (!G || !out || !partials) return fail(VITK_ERR_INVALID, "vitk_colsum_bf16: null pointer");
    if (R <= 0 || M <= 0 || partial_rows <= 0) return fail(VITK_ERR_INVALID, "vitk_colsum_bf16: non-positive size");
    if (M % 256 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_colsum_bf16: M=%d is not a multiple of 256", M);
    if (!aligned16(G) || !aligned16(out) || !aligned16(partials))
        return fail(VITK_ERR_INVALID, "vitk_colsum_bf16: pointers must be 16-byte aligned");
    return launched(launch_colsum_bf16(G, out, partials, partial_rows, R, M, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_colsum_bf16");
}

int vitk_gemm_plan(int M, int N, int K, int cus, int* plan) {
    if (!plan) return fail(VITK_ERR_INVALID, "vitk_gemm_plan: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || N % 256 != 0 || K % 64 != 0 || cus < 0)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_gemm_plan: M=%d N=%d K=%d (N a multiple of 256, K of 64)", M, N, K);
    GemmArgs a = {};
    a.M = M; a.N = N; a.K = K;
    const PipePlan pl = make_plan(a, cus > 0 ? cus : num_cus(), gemm_pp_supports(a, false));
    const int out[10] = {pl.mtot, pl.tiles_m, pl.tiles_n, pl.gn, pl.mixed, pl.xcd_units, pl.ha, pl.na, pl.hb, pl.nb};
    for (int i = 0; i < 10; ++i) plan[i] = out[i];
    return VITK_OK;
}

int vitk_init(void) {
    return launched(colsum_static_scratch(nullptr), "vitk_init");
}

int vitk_colsum_bf16_static(const void* G, float* out, int R, int M, void* stream) {
    if (!G || !out) return fail(VITK_ERR_INVALID, "vitk_colsum_bf16_static: null pointer");
    if (R <= 0 || M <= 0) return fail(VITK_ERR_INVALID, "vitk_colsum_bf16_static: non-positive size");
    if (M % 256 != 0) return fail(VITK_ERR_UNSUPPORTED, "vitk_colsum_bf16_static: M=%d is not a multiple of 256", M);
    if (M > kColsumStaticMaxM)
        return fail(VITK_ERR_UNSUPPORTED, "vitk_colsum_bf16_static: M=%d exceeds the built-in scratch (M <= %d); use vitk_colsum_bf16 with a caller workspace",
                    M, kColsumStaticMaxM);
    if (!aligned16(G) || !aligned16(out)) return fail(VITK_ERR_INVALID, "vitk_colsum_bf16_static: pointers must be 16-byte aligned");
    float* scratch = nullptr;
    hipError_t e = colsum_static_scratch(&scratch);
    if (e != hipSuccess) return launched(e, "vitk_colsum_bf16_static(scratch)");
    return launched(launch_colsum_bf16(G, out, scratch, kColsumStaticRows, R, M, reinterpret_cast<hipStream_t>(stream)),
                    "vitk_colsum_bf16_static");
}

}  // extern "C"
