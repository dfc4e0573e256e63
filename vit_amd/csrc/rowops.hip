// Bandwidth-bound row kernels of the ViT forward path (gfx950 / CDNA4):
// LayerNorm over the fp32 residual stream, the cls-token row, and the
// fp32 -> bf16 image cast.
#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

// ---------------------------------------------------------------------------
// LayerNorm (nn.LayerNorm(D, eps): models/transformer.py:56,60,93): per row,
// mean and biased variance in fp32, y = (x - mean) * rsqrt(var + eps) * g + b.
// One wave owns one row: each lane keeps NV float4 of it in registers, so the
// row is read from HBM exactly once (4*D bytes) and written once (2*D bytes as
// bf16, the next GEMM's operand type).  Two-pass statistics on the register
// copy (mean first, then centred squares) match torch's numerics closely.
// ---------------------------------------------------------------------------
template <int NV, bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_rows(const float* __restrict__ x, int64_t x_stride,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, void* __restrict__ y,
                                                      int64_t y_stride, int rows, float eps) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    f32x4 g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + i * 256 + lane * 4);
        b[i] = *reinterpret_cast<const f32x4*>(beta + i * 256 + lane * 4);
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += waves_total) {
        const float* xr = x + (int64_t)row * x_stride;
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = wave_sum(s) * (1.0f / D);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                ss += d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) * (1.0f / D) + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[i][j] + b[i][j];
            if constexpr (OUT_BF16) {
                u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + (int64_t)row * y_stride + i * 256 +
                                          lane * 4) = pk;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (int64_t)row * y_stride + i * 256 +
                                          lane * 4) = o;
            }
        }
    }
}

template <int NV>
static hipError_t ln_launch(const float* x, int64_t xs, const float* g, const float* b, void* y, int64_t ys,
                            int rows, float eps, bool out_bf16, hipStream_t s) {
    int blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;   // grid-stride beyond 16 row-waves per CU
    if (out_bf16)
        hipLaunchKernelGGL((layernorm_rows<NV, true>), dim3(blocks), dim3(256), 0, s, x, xs, g, b, y, ys, rows, eps);
    else
        hipLaunchKernelGGL((layernorm_rows<NV, false>), dim3(blocks), dim3(256), 0, s, x, xs, g, b, y, ys, rows, eps);
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta,
                            void* y, int64_t y_row_stride, int rows, int D, float eps, bool out_bf16,
                            hipStream_t s) {
    switch (D) {
        case 256: return ln_launch<1>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, eps, out_bf16, s);
        case 512: return ln_launch<2>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, eps, out_bf16, s);
        case 768: return ln_launch<3>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, eps, out_bf16, s);
        case 1024: return ln_launch<4>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, eps, out_bf16, s);
        case 1280: return ln_launch<5>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, eps, out_bf16, s);
        default: return hipErrorInvalidValue;
    }
}

// Row factors of a LayerNorm folded into the GEMMs either side of it: the producing GEMM's epilogue left (sum, sum of
// squares) per row and 64-column strip; this adds the strips in index order and writes (rstd, -mean * rstd) per row
// (biased variance, eps inside the root: nn.LayerNorm, models/transformer.py:56,60).  One lane per row.
__global__ __launch_bounds__(256) void ln_finalize_rows(const float* __restrict__ stats, float* __restrict__ ab, int rows,
                                                        int strips, float inv_d, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const f32x4* sp = reinterpret_cast<const f32x4*>(stats + (size_t)row * strips * 2);     // strips is even: 16-byte pieces
    float s = 0.f, q = 0.f;
    for (int i = 0; i < strips / 2; ++i) {
        const f32x4 v = sp[i];
        s += v[0]; q += v[1];
        s += v[2]; q += v[3];
    }
    const float mean = s * inv_d;
    const float var = fmaxf(q * inv_d - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<f32x2*>(ab + 2 * (size_t)row) = f32x2{rstd, -mean * rstd};
}

hipError_t launch_ln_finalize(const float* stats, float* ab, int rows, int D, float eps, hipStream_t s) {
    hipLaunchKernelGGL(ln_finalize_rows, dim3((rows + 255) / 256), dim3(256), 0, s, stats, ab, rows, D / 64, 1.0f / D, eps);
    return hipGetLastError();
}

// x[b, 0, :] = cls_token + pos_embed[0]   (vision_transformer.py:48-50)
__global__ void cls_pos_rows(const float* __restrict__ cls, const float* __restrict__ pos,
                             float* __restrict__ x, int batch, int ntok, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // float4 index inside a row
    const int b = blockIdx.y;
    if (i * 4 < D) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cls + i * 4);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos + i * 4);
        *reinterpret_cast<f32x4*>(x + (size_t)b * ntok * D + i * 4) = c + p;
    }
}

hipError_t launch_cls_pos(const float* cls, const float* pos, float* x, int batch, int ntok, int D,
                          hipStream_t s) {
    dim3 grid((D / 4 + 255) / 256, batch);
    hipLaunchKernelGGL(cls_pos_rows, grid, dim3(256), 0, s, cls, pos, x, batch, ntok, D);
    return hipGetLastError();
}

// fp32 -> bf16, 8 elements (32 B in, 16 B out) per lane per step; n % 8 == 0.
__global__ __launch_bounds__(256) void cast_f32_bf16(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                     int64_t n8, int tail) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 8 + 4);
        u32x4 o = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail)             // the last n % 8 elements
        out[n8 * 8 + threadIdx.x] = (bf16_t)in[n8 * 8 + threadIdx.x];
}

hipError_t launch_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t s) {
    const int64_t n8 = n / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f32_bf16, dim3((unsigned)blocks), dim3(256), 0, s, in,
                       reinterpret_cast<bf16_t*>(out), n8, (int)(n - n8 * 8));
    return hipGetLastError();
}

// Position-embedding rescale (utils.py:241-261): cls row copied, the g x g grid of rows resampled to
// G x G by order-1 interpolation with output index o at input coordinate o (g-1)/(G-1) per axis (what
// scipy.ndimage.zoom(order=1) computes).  One block per output row, threads over channels.
__global__ __launch_bounds__(256) void pos_embed_scale_rows(const float* __restrict__ in, float* __restrict__ out,
                                                            int g, int G, int D) {
    const int row = blockIdx.x;                      // output token
    float w00 = 1.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
    int r00 = 0, r01 = 0, r10 = 0, r11 = 0;
    if (row > 0) {
        const int oi = (row - 1) / G, oj = (row - 1) - oi * G;
        const double step = G > 1 ? (double)(g - 1) / (double)(G - 1) : 0.0;
        const double ci = oi * step, cj = oj * step;
        int i0 = (int)ci, j0 = (int)cj;
        i0 = i0 < g - 1 ? i0 : g - 1;
        j0 = j0 < g - 1 ? j0 : g - 1;
        const int i1 = i0 + 1 < g ? i0 + 1 : g - 1, j1 = j0 + 1 < g ? j0 + 1 : g - 1;
        const double fi = ci - i0, fj = cj - j0;
        w00 = (float)((1.0 - fi) * (1.0 - fj)); w01 = (float)((1.0 - fi) * fj);
        w10 = (float)(fi * (1.0 - fj));         w11 = (float)(fi * fj);
        r00 = 1 + i0 * g + j0; r01 = 1 + i0 * g + j1; r10 = 1 + i1 * g + j0; r11 = 1 + i1 * g + j1;
    }
    for (int d = threadIdx.x; d < D; d += 256)
        out[(size_t)row * D + d] = w00 * in[(size_t)r00 * D + d] + w01 * in[(size_t)r01 * D + d] +
                                   w10 * in[(size_t)r10 * D + d] + w11 * in[(size_t)r11 * D + d];
}

// Patch rows for patch sizes the gather loader of the GEMM cannot address (not a power of two, e.g.
// ViT-H/14): cols[m][k] = img[b][c][ph*P + i][pw*P + j] for m = (b, ph, pw), k = c*P*P + i*P + j
// (the Conv2d weight's flattening, models/vision_transformer.py:18), zero for k in [3*P*P, kpad).
// One thread per (row m, channel c, patch row i): P contiguous source elements.
__global__ __launch_bounds__(256) void im2col_patches(const bf16_t* __restrict__ img, bf16_t* __restrict__ cols,
                                                      int batch, int img_h, int img_w, int patch, int grid_w,
                                                      int npatch, int kpad) {
    const int64_t total = (int64_t)batch * npatch * 3 * patch;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int k0 = 3 * patch * patch;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t % patch);
        const int c = (int)((t / patch) % 3);
        const int64_t m = t / (3 * patch);
        const int b = (int)(m / npatch), pidx = (int)(m - (int64_t)b * npatch);
        const int ph = pidx / grid_w, pw = pidx - ph * grid_w;
        const bf16_t* src = img + (((int64_t)b * 3 + c) * img_h + (ph * patch + i)) * img_w + pw * patch;
        bf16_t* dst = cols + m * kpad + (c * patch + i) * patch;
        for (int j = 0; j < patch; ++j) dst[j] = src[j];
        if (c == 2 && i == patch - 1)
            for (int k = k0; k < kpad; ++k) cols[m * kpad + k] = (bf16_t)0.0f;
    }
}

hipError_t launch_im2col(const void* img, void* cols, int batch, int img_h, int img_w, int patch, int kpad,
                         hipStream_t s) {
    const int grid_w = img_w / patch, npatch = (img_h / patch) * grid_w;
    const int64_t total = (int64_t)batch * npatch * 3 * patch;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_patches, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(img),
                       reinterpret_cast<bf16_t*>(cols), batch, img_h, img_w, patch, grid_w, npatch, kpad);
    return hipGetLastError();
}

// Input step in front of the forward (data.py:59-72, DataFetcher.preload): uint8 NCHW pixels ->
// (float(x) - mean[c]) / std[c], written as bf16 (the patch GEMM's operand type) or f32.  16 pixels per
// lane per step: one 16-byte load, two 16-byte bf16 stores (or four f32 ones).
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void normalize_u8(const uint8_t* __restrict__ in, void* __restrict__ out,
                                                    int64_t n16, int64_t plane16, float m0, float m1, float m2,
                                                    float s0, float s1, float s2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const int c = (int)((i / plane16) % 3);               // a 16-pixel group never straddles a channel plane
        const float m = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
        const u32x4 px = *reinterpret_cast<const u32x4*>(in + i * 16);
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ((float)((px[j >> 2] >> (8 * (j & 3))) & 0xFFu) - m) / sd;
        if constexpr (OUT_BF16) {
            bf16_t* o = reinterpret_cast<bf16_t*>(out) + i * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 w = {pack_bf16(v[8 * h], v[8 * h + 1]), pack_bf16(v[8 * h + 2], v[8 * h + 3]),
                           pack_bf16(v[8 * h + 4], v[8 * h + 5]), pack_bf16(v[8 * h + 6], v[8 * h + 7])};
                *reinterpret_cast<u32x4*>(o + 8 * h) = w;
            }
        } else {
            float* o = reinterpret_cast<float*>(out) + i * 16;
#pragma unroll
            for (int h = 0; h < 4; ++h)
                *reinterpret_cast<f32x4*>(o + 4 * h) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        }
    }
}

// the same per pixel, for planes whose size is not a multiple of 16 (a 16-pixel group would straddle channels)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void normalize_u8_any(const uint8_t* __restrict__ in, void* __restrict__ out, int64_t n,
                                                        int64_t plane, float m0, float m1, float m2, float s0, float s1,
                                                        float s2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / plane) % 3);
        const float m = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
        const float v = ((float)in[i] - m) / sd;
        if constexpr (OUT_BF16) reinterpret_cast<bf16_t*>(out)[i] = (bf16_t)v;
        else reinterpret_cast<float*>(out)[i] = v;
    }
}

hipError_t launch_normalize_u8(const void* in, void* out, int64_t n, int64_t plane, const float* mean,
                               const float* std, bool out_bf16, hipStream_t s) {
    if (plane % 16 != 0) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (out_bf16)
            hipLaunchKernelGGL(normalize_u8_any<true>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(in),
                               out, n, plane, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
        else
            hipLaunchKernelGGL(normalize_u8_any<false>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(in),
                               out, n, plane, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
        return hipGetLastError();
    }
    const int64_t n16 = n / 16;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (out_bf16)
        hipLaunchKernelGGL(normalize_u8<true>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(in),
                           out, n16, plane / 16, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    else
        hipLaunchKernelGGL(normalize_u8<false>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(in),
                           out, n16, plane / 16, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    return hipGetLastError();
}

// Step after the forward (utils.py:182-195, accuracy): rank[b] = position of the target class in the
// descending order of logits[b] (0 = top-1), ties broken towards the lower class index, so
// "correct at k" is rank < k.  One wave per row.
__global__ __launch_bounds__(256) void target_rank_rows(const float* __restrict__ logits,
                                                        const int64_t* __restrict__ target, int* __restrict__ rank,
                                                        int rows, int classes) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = logits + (size_t)row * classes;
    const int64_t t = target[row];
    if (t < 0 || t >= classes) {                  // not a class: never counted as correct
        if (lane == 0) rank[row] = classes;
        return;
    }
    const float xt = x[t];
    int above = 0;
    for (int j = lane; j < classes; j += 64) {
        const float v = x[j];
        above += (v > xt || (v == xt && j < (int)t)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o, 64);
    if (lane == 0) rank[row] = above;
}

hipError_t launch_target_rank(const float* logits, const int64_t* target, int* rank, int rows, int classes,
                              hipStream_t s) {
    hipLaunchKernelGGL(target_rank_rows, dim3((rows + 3) / 4), dim3(256), 0, s, logits, target, rank, rows, classes);
    return hipGetLastError();
}

hipError_t launch_pos_embed_scale(const float* in, float* out, int g, int G, int D, hipStream_t s) {
    hipLaunchKernelGGL(pos_embed_scale_rows, dim3(1 + G * G), dim3(256), 0, s, in, out, g, G, D);
    return hipGetLastError();
}

}  // namespace vitk
