// Backward kernels of the training step (gfx950 / CDNA4).  Bandwidth-bound row work only so far: the
// GEMM and attention gradients still come from library calls on the host side.
#include <mutex>

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
                                                          float* __restrict__ dx, bf16_t* __restrict__ dx16,
                                                          float* __restrict__ partials, int rows, float eps) {
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
            if (dx16) {     // the same gradient as the bf16 operand of the backward GEMMs that consume it next
                u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
                *reinterpret_cast<u32x2*>(dx16 + (int64_t)row * D + i * 256 + lane * 4) = pk;
            }
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
                                const float* add, float* dx, void* dx16,
                                float* dgamma, float* dbeta, float* partials, int partial_rows, int rows, float eps,
                                hipStream_t s) {
    int blocks = (rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks > partial_rows) blocks = partial_rows;
    if (g_bf16)
        hipLaunchKernelGGL((layernorm_bwd_rows<NV, true>), dim3(blocks), dim3(256), 0, s, x, xs, gamma, g, add, dx, reinterpret_cast<bf16_t*>(dx16), partials, rows, eps);
    else
        hipLaunchKernelGGL((layernorm_bwd_rows<NV, false>), dim3(blocks), dim3(256), 0, s, x, xs, gamma, g, add, dx, reinterpret_cast<bf16_t*>(dx16), partials, rows, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    constexpr int D = NV * 256;
    hipLaunchKernelGGL(layernorm_bwd_reduce, dim3(2 * D / 64), dim3(1024), 0, s, partials, blocks, D, dgamma, dbeta);
    return hipGetLastError();
}

hipError_t launch_layernorm_backward(const float* x, int64_t x_row_stride, const float* gamma, const void* g,
                                     bool g_bf16, const float* add, float* dx, void* dx16, float* dgamma, float* dbeta, float* partials,
                                     int partial_rows, int rows, int D, float eps, hipStream_t s) {
    switch (D) {
        case 256: return ln_bwd_launch<1>(x, x_row_stride, gamma, g, g_bf16, add, dx, dx16, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 512: return ln_bwd_launch<2>(x, x_row_stride, gamma, g, g_bf16, add, dx, dx16, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 768: return ln_bwd_launch<3>(x, x_row_stride, gamma, g, g_bf16, add, dx, dx16, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 1024: return ln_bwd_launch<4>(x, x_row_stride, gamma, g, g_bf16, add, dx, dx16, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        case 1280: return ln_bwd_launch<5>(x, x_row_stride, gamma, g, g_bf16, add, dx, dx16, dgamma, dbeta, partials, partial_rows, rows, eps, s);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Bias gradient of a Linear layer: out[m] = sum_r G[r][m], G bf16 [R, M] (M a multiple of 256), f32
// accumulation.  Block (x, y) takes 256 columns and the y-th chunk of rows: a half-wave reads one row's
// 512 bytes, 8 columns per lane, the block's 8 half-waves walk the chunk's rows; their sums are added in a
// fixed order in LDS and written as one partial row; reduce_partial_rows adds the chunks in order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_bf16_partial(const bf16_t* __restrict__ G, float* __restrict__ partials,
                                                           int R, int M, int rows_per_chunk) {
    __shared__ float red[8][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hw = wave * 2 + (lane >> 5);                    // half-wave 0..7
    const int col = blockIdx.x * 256 + (lane & 31) * 8;
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    r1 = r1 < R ? r1 : R;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + hw; r < r1; r += 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(G + (size_t)r * M + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[2 * j] += bf16_lo(v[j]);
            acc[2 * j + 1] += bf16_hi(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[hw][(lane & 31) * 8 + j] = acc[j];
    __syncthreads();
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += red[j][threadIdx.x];
    partials[(size_t)blockIdx.y * M + blockIdx.x * 256 + threadIdx.x] = t;
}

// out[c] = sum over n partial rows of `width` floats (width a multiple of 64), in row order
__global__ __launch_bounds__(1024) void reduce_partial_rows(const float* __restrict__ partials, int n, int width,
                                                            float* __restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float acc = 0.f;
    for (int b = w; b < n; b += 16) acc += partials[(size_t)b * width + col];
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        float t = red[0][lane];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][lane];
        out[col] = t;
    }
}

hipError_t launch_colsum_bf16(const void* G, float* out, float* partials, int partial_rows, int R, int M, hipStream_t s) {
    int chunks = (R + 63) / 64;
    if (chunks > 256) chunks = 256;
    if (chunks > partial_rows) chunks = partial_rows;
    const int rows_per_chunk = (R + chunks - 1) / chunks;
    chunks = (R + rows_per_chunk - 1) / rows_per_chunk;
    hipLaunchKernelGGL(colsum_bf16_partial, dim3(M / 256, chunks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(G),
                       partials, R, M, rows_per_chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reduce_partial_rows, dim3(M / 64), dim3(1024), 0, s, partials, chunks, M, out);
    return hipGetLastError();
}

// Partial sums of vitk_colsum_bf16_static: 4 MiB of device memory that belongs to the library's code object,
// one copy per GPU the library runs on, for the life of the process.  Host code never uses the symbol as
// an address: the device address is looked up once per device and that pointer goes to the kernels.
// Every call on a device shares the buffer; each call overwrites the partial rows it then reads, so nothing
// has to be cleared, but two calls that overlap on one device (different streams) race.
static __device__ __align__(256) float scratch[kColsumStaticRows * kColsumStaticMaxM];

hipError_t colsum_static_scratch(float** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static float* ptrs[kMaxDevices] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        ptrs[dev] = static_cast<float*>(p);
    }
    if (out) *out = ptrs[dev];
    return hipSuccess;
}

}  // namespace vitk
