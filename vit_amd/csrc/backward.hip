// Backward kernels of the training step (gfx950 / CDNA4).  Bandwidth-bound row work only so far: the
// GEMM and attention gradients still come from library calls on the host side.
#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

// ---------------------------------------------------------------------------
// LayerNorm backward (the adjoint of layernorm_rows; nn.LayerNorm at models/transformer.py:56,60,93).
// Per row, with xh = (x - mean) * rstd and gy = g * gamma:
//     dx = rstd * (gy - mean_D(gy) - xh * mean_D(gy * xh))
//     dgamma[d] += g[d] * xh[d],   dbeta[d] += g[d]
// One wave owns a row at a time and keeps it in registers (x is read once: 4 D bytes, g once: 2 or 4 D,
// dx written once: 4 D); mean and rstd are recomputed from x exactly as the forward computes them, so
// nothing but x has to be kept from the forward.  The column sums are kept per lane across the rows a
// wave walks, reduced over the block's 4 waves in LDS and written as one [2, D] partial per block; a
// second kernel adds the partials in a fixed order (no atomics: results are reproducible).
// ---------------------------------------------------------------------------
template <int NV, bool G_BF16>
__global__ __launch_bounds__(256) void layernorm_bwd_rows(const float* __restrict__ x, int64_t x_stride,
                                                          const float* __restrict__ gamma,
                                                          const void* __restrict__ g, const float* __restrict__ add,
                                                          float* __restrict__ dx, float* __restrict__ partials, int rows,
                                                          float eps) {
    constexpr int D = NV * 256;
    __shared__ float red[4][2][D];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    f32x4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gam[i] = *reinterpret_cast<const f32x4*>(gamma + i * 256 + lane * 4);
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += waves_total) {
        const float* xr = x + (int64_t)row * x_stride;
        f32x4 v[NV], gr[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
            if constexpr (G_BF16) {
                const u32x2 p = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g) + (int64_t)row * D +
                                                                i * 256 + lane * 4);
                gr[i] = f32x4{__builtin_bit_cast(float, p[0] << 16), __builtin_bit_cast(float, p[0] & 0xFFFF0000u),
                              __builtin_bit_cast(float, p[1] << 16), __builtin_bit_cast(float, p[1] & 0xFFFF0000u)};
            } else {
                gr[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + (int64_t)row * D + i * 256 +
                                                        lane * 4);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = wave_sum(s) * (1.0f / D);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                ss += d * d;
            }
        const float rstd = rsqrtf(wave_sum(ss) * (1.0f / D) + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[i][j] - mean) * rstd;
                const float gy = gr[i][j] * gam[i][j];
                v[i][j] = xh;
                c1 += gy;
                c2 += gy * xh;
                dg[i][j] += gr[i][j] * xh;
                db[i][j] += gr[i][j];
            }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (gr[i][j] * gam[i][j] - c1 - v[i][j] * c2);
            // the gradient that reaches x past the LayerNorm (the residual branch), added here instead of by
            // a separate pass over both tensors
            if (add) o = o + *reinterpret_cast<const f32x4*>(add + (int64_t)row * D + i * 256 + lane * 4);
            *reinterpret_cast<f32x4*>(dx + (int64_t)row * D + i * 256 + lane * 4) = o;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<f32x4*>(&red[wave][0][i * 256 + lane * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&red[wave][1][i * 256 + lane * 4]) = db[i];
    }
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, d = i - which * D;
        out[i] = (red[0][which][d] + red[1][which][d]) + (red[2][which][d] + red[3][which][d]);
    }
}

// dgamma / dbeta = sum over the blocks' partials in a fixed order: a workgroup of 16 waves takes 64 of the
// 2 D columns; wave w adds partial rows w, w + 16, ... (8 loads in flight per lane), then the 16 sums are
// added in wave order through LDS.
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce(const float* __restrict__ partials, int nblocks, int D,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;                   // 2 D is a multiple of 64
    const size_t stride = (size_t)2 * D;
    float acc = 0.f;
    int b = w;
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partials[(size_t)(b + j * 16) * stride + col];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; b < nblocks; b += 16) acc += partials[(size_t)b * stride + col];
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        float t = red[0][lane];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][lane];
        if (col < D) dgamma[col] = t;
        else dbeta[col - D] = t;
    }
}

template <int NV>
static hipError_t ln_bwd_launch(const float* x, int64_t xs, const float* gamma, const void* g, bool g_bf16,
                                const float* add, float* dx,
                                float* dgamma, float* dbeta, float* partials, int partial_rows, int rows, float eps,
                                hipStream_t s) {
    int blocks = (rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks > partial_rows) blocks = partial_rows;
    if (g_bf16)
        hipLaunchKernelGGL((layernorm_bwd_rows<NV, true>), dim3(blocks), dim3(256), 0, s, x, xs, gamma, g, add, dx, partials, rows, eps);
    else
        hipLaunchKernelGGL((layernorm_bwd_rows<NV, false>), dim3(blocks), dim3(256), 0, s, x, xs, gamma, g, add, dx, partials, rows, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    constexpr int D = NV * 256;
    hipLaunchKernelGGL(layernorm_bwd_reduce, dim3(2 * D / 64), dim3(1024), 0, s, partials, blocks, D, dgamma, dbeta);
    return hipGetLastError();
}

hipError_t launch_layernorm_backward(const float* x, int64_t x_row_stride, const float* gamma, const void* g,
                                     bool g_bf16, const float* add, float* dx, float* dgamma, float* dbeta, float* partials,
                                     int partial_rows, int rows, int D, float eps, hipStream_t s) {
    switch (D) {
        case 256: return ln_bwd_launch<1>(x, x_row_stride, gamma, g, g_bf16, add, dx, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 512: return ln_bwd_launch<2>(x, x_row_stride, gamma, g, g_bf16, add, dx, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 768: return ln_bwd_launch<3>(x, x_row_stride, gamma, g, g_bf16, add, dx, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 1024: return ln_bwd_launch<4>(x, x_row_stride, gamma, g, g_bf16, add, dx, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 1280: return ln_bwd_launch<5>(x, x_row_stride, gamma, g, g_bf16, add, dx, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vitk
