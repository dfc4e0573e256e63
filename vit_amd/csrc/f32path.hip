// fp32 compute mode of the ViT forward path (gfx950 / CDNA4): every operand and every intermediate
// is f32, for callers that need the reference's fp32 results (max-abs logit error <= 1e-4) rather
// than bf16 throughput.  The matrix products run on the f32-input MFMA (v_mfma_f32_32x32x2_f32:
// exact f32 multiply-add chains at the vector rate), softmax / GELU / tanh use the libm-grade device
// functions.  Kept simple on purpose: one tiled GEMM with pluggable loader and epilogue, and an
// attention kernel that holds a 32-query strip of the score matrix in LDS.
#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

// ---------------------------------------------------------------------------------------------
// C[m][n] = sum_k A(m, k) * W[n][k]  (+ epilogue).  Block tile 128 x 128, K in steps of 16 through
// one LDS buffer held k-major ([k][row]: lanes of a fragment read consecutive addresses), next tile
// prefetched into registers while the current one is multiplied.  4 waves as 2 x 2, 64 x 64 per wave
// = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).
// ---------------------------------------------------------------------------------------------
constexpr int FBM = 128, FBN = 128, FBK = 16, FLD = FBM + 4;

struct F32GemmArgs {
    const float* A;      // [M, K] row-major, or the NCHW image for the patch loader
    const float* W;      // [N, K] row-major
    const float* bias;   // [N] or nullptr
    const float* resid;  // F32_RESID (may alias out)
    const float* pos;    // F32_PATCH: pos_embed [ntok, N]
    float* out;
    int M, N, K;
    int ntok, heads, hd, batch;          // F32_QKV / F32_PATCH
    int img_h, img_w, patch, grid_w, npatch;
};

// element (m, k) of the implicit patch matrix: row m = (image, patch), k = c*P*P + i*P + j
// (Conv2d weight [D,3,P,P] flattened; models/vision_transformer.py:18-19,47)
__device__ __forceinline__ float patch_elem(const F32GemmArgs& p, int m, int k) {
    const int b = m / p.npatch, pidx = m - b * p.npatch;
    const int ph = pidx / p.grid_w, pw = pidx - ph * p.grid_w;
    const int pp = p.patch * p.patch;
    const int c = k / pp, r = k - c * pp;
    const int i = r / p.patch, j = r - i * p.patch;
    return p.A[(((size_t)b * 3 + c) * p.img_h + (ph * p.patch + i)) * p.img_w + pw * p.patch + j];
}

template <int EPI, bool GATHER>
__global__ __launch_bounds__(256) void gemm_f32_128x128(const F32GemmArgs p) {
    __shared__ float As[FBK][FLD];
    __shared__ float Ws[FBK][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;

    // loader role: one tile row, 8 consecutive k
    const int lrow = tid & 127, lk = (tid >> 7) * 8;
    float ra[8], rw[8];
    auto fetch = [&](int k0) {
        const int m = m0 + lrow, n = n0 + lrow;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + lk + i;
            float a = 0.f, w = 0.f;
            if (k < p.K) {
                if (m < p.M) a = GATHER ? patch_elem(p, m, k) : p.A[(size_t)m * p.K + k];
                if (n < p.N) w = p.W[(size_t)n * p.K + k];
            }
            ra[i] = a;
            rw[i] = w;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            As[lk + i][lrow] = ra[i];
            Ws[lk + i][lrow] = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
        __syncthreads();                 // previous tile fully consumed
        stash();
        __syncthreads();
        if (k0 + FBK < p.K) fetch(k0 + FBK);
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2) {
            float a[2], w[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[kk + fk][wm * 64 + t * 32 + fr];
                w[t] = Ws[kk + fk][wn * 64 + t * 32 + fr];
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], w[tb], acc[ta][tb], 0, 0, 0);
        }
    }

    // ---- epilogue: accumulator i of a lane is C[row 8(i>>2) + 4(lane>>5) + (i&3)][col lane&31] of its tile
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int n = n0 + wn * 64 + tb * 32 + fr;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + ta * 32 + 8 * (i >> 2) + 4 * fk + (i & 3);
                if (m >= p.M) continue;
                float v = acc[ta][tb][i] + bias;
                if constexpr (EPI == F32_BIAS) {
                    p.out[(size_t)m * p.N + n] = v;
                } else if constexpr (EPI == F32_GELU) {
                    p.out[(size_t)m * p.N + n] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                } else if constexpr (EPI == F32_TANH) {
                    p.out[(size_t)m * p.N + n] = tanhf(v);
                } else if constexpr (EPI == F32_RESID) {
                    p.out[(size_t)m * p.N + n] = p.resid[(size_t)m * p.N + n] + v;
                } else if constexpr (EPI == F32_QKV) {
                    // column n = which*D + h*hd + e, row m = b*ntok + t  ->  qkv[which][b][h][t][e]
                    // (reshape(B,N,3,H,hd).permute(2,0,3,1,4), models/transformer.py:42)
                    const int D = p.heads * p.hd;
                    const int which = n / D, rem = n - which * D, h = rem / p.hd, e = rem - h * p.hd;
                    const int b = m / p.ntok, t = m - b * p.ntok;
                    p.out[((((size_t)which * p.batch + b) * p.heads + h) * p.ntok + t) * p.hd + e] = v;
                } else if constexpr (EPI == F32_PATCH) {
                    // patch row (b, pidx) lands at token 1 + pidx with that token's position embedding
                    // (models/vision_transformer.py:47-50)
                    const int b = m / p.npatch, pidx = m - b * p.npatch;
                    p.out[((size_t)b * p.ntok + 1 + pidx) * p.N + n] = v + p.pos[(size_t)(1 + pidx) * p.N + n];
                }
            }
        }
}

// Few rows (the head: one row per image): the 128 x 128 tiles would leave all but a dozen CUs idle.  Here a
// wave computes 16 rows x 32 columns straight from global memory (both operands are L2-resident at these
// sizes) on v_mfma_f32_16x16x4_f32; a lane loads 4 consecutive k of its row per 16-k step and feeds them to 4
// MFMAs (the k order inside a step is permuted identically for both operands).  Operands swapped (MFMA row =
// n), so a lane ends with 4 consecutive n of one row m.  K % 4 == 0.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_rows(const F32GemmArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 32;
    if (n0 >= p.N) return;
    int am = m0 + fr;
    am = am < p.M ? am : p.M - 1;
    const float* ap = p.A + (size_t)am * p.K + fq * 4;
    const float* wp[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int wr = n0 + nt * 16 + fr;
        wr = wr < p.N ? wr : p.N - 1;
        wp[nt] = p.W + (size_t)wr * p.K + fq * 4;
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](const f32x4& a, const f32x4& w0, const f32x4& w1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[j], acc[1], 0, 0, 0);
        }
    };
    int k = 0;
    for (; k + 64 <= p.K; k += 64) {              // four 16-k steps, their 12 loads in flight together
        f32x4 a[4], w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + k + u * 16);
            w0[u] = *reinterpret_cast<const f32x4*>(wp[0] + k + u * 16);
            w1[u] = *reinterpret_cast<const f32x4*>(wp[1] + k + u * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) step(a[u], w0[u], w1[u]);
    }
    for (; k < p.K; k += 16) {
        const bool ok = k + fq * 4 < p.K;
        step(ok ? *reinterpret_cast<const f32x4*>(ap + k) : zero, ok ? *reinterpret_cast<const f32x4*>(wp[0] + k) : zero,
             ok ? *reinterpret_cast<const f32x4*>(wp[1] + k) : zero);
    }
    const int m = m0 + fr;
    if (m >= p.M) return;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + nt * 16 + fq * 4 + j;
            if (n >= p.N) continue;
            float v = acc[nt][j] + (p.bias ? p.bias[n] : 0.f);
            if constexpr (EPI == F32_TANH) v = tanhf(v);
            p.out[(size_t)m * p.N + n] = v;
        }
}

template <int EPI, bool GATHER>
static hipError_t launch_f32(const F32GemmArgs& a, hipStream_t s) {
    if constexpr (!GATHER && (EPI == F32_BIAS || EPI == F32_TANH)) {
        // every 16-row block streams the whole of W once with no reuse in LDS: only while that stays small (the
        // head at batches up to 1024: 64 blocks x 1000 x 768 words); anything heavier takes the tiled kernel
        const long long w_words = (long long)((a.M + 15) / 16) * a.N * a.K;
        if (a.M <= 1024 && w_words <= (64LL << 20) && a.K % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(a.W) & 15) == 0) {
            dim3 grid((a.N + 127) / 128, (a.M + 15) / 16);
            hipLaunchKernelGGL((gemm_f32_rows<EPI>), grid, dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    dim3 grid((a.N + FBN - 1) / FBN, (a.M + FBM - 1) / FBM);
    hipLaunchKernelGGL((gemm_f32_128x128<EPI, GATHER>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* out, int M,
                           int N, int K, int epilogue, hipStream_t s) {
    F32GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.resid = resid; a.out = out;
    a.M = M; a.N = N; a.K = K;
    switch (epilogue) {
        case F32_BIAS: return launch_f32<F32_BIAS, false>(a, s);
        case F32_GELU: return launch_f32<F32_GELU, false>(a, s);
        case F32_TANH: return launch_f32<F32_TANH, false>(a, s);
        case F32_RESID: return launch_f32<F32_RESID, false>(a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_qkv_f32(const float* A, const float* W, const float* bias, float* qkv, int batch, int ntok,
                          int heads, int hd, hipStream_t s) {
    F32GemmArgs a = {};
    a.A = A; a.W = W; a.bias = bias; a.out = qkv;
    a.M = batch * ntok; a.N = 3 * heads * hd; a.K = heads * hd;
    a.batch = batch; a.ntok = ntok; a.heads = heads; a.hd = hd;
    return launch_f32<F32_QKV, false>(a, s);
}

hipError_t launch_patch_f32(const float* img, const float* w, const float* bias, const float* pos, float* x,
                            int batch, int img_h, int img_w, int patch, int D, hipStream_t s) {
    F32GemmArgs a = {};
    a.A = img; a.W = w; a.bias = bias; a.pos = pos; a.out = x;
    a.grid_w = img_w / patch;
    a.npatch = (img_h / patch) * a.grid_w;
    a.M = batch * a.npatch; a.N = D; a.K = 3 * patch * patch;
    a.batch = batch; a.ntok = a.npatch + 1;
    a.img_h = img_h; a.img_w = img_w; a.patch = patch;
    return launch_f32<F32_PATCH, true>(a, s);
}

// ---------------------------------------------------------------------------------------------
// Attention in f32 (models/transformer.py:43-46).  One workgroup = one (image, head) and 32 query
// rows; 256 threads.  The strip's scores S[32][ntok] live in LDS:
//   1. S = (q k^T) * scale    thread = (query, key class mod 8), q rows staged in LDS
//   2. row softmax            one wave per 8 rows, wave reductions
//   3. ctx = P v              thread = (query, eighth of head_dim)
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attention_f32(const float* __restrict__ Q, const float* __restrict__ K,
                                                     const float* __restrict__ V, float* __restrict__ O, int ntok,
                                                     int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int sstride = ntok + 1;                    // odd-ish row stride: the 32 query lanes spread over banks
    float* S = lds;                                  // [32][sstride]
    float* Qs = lds + 32 * sstride;                  // [32][HD + 1]
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, q0 = blockIdx.y * 32;
    const float* qg = Q + (size_t)bh * ntok * HD;
    const float* kg = K + (size_t)bh * ntok * HD;
    const float* vg = V + (size_t)bh * ntok * HD;

    for (int i = tid; i < 32 * HD; i += 256) {
        const int r = i / HD, d = i - r * HD;
        const int qr = q0 + r < ntok ? q0 + r : ntok - 1;        // padded rows repeat the last query; never stored
        Qs[r * (HD + 1) + d] = qg[(size_t)qr * HD + d];
    }
    __syncthreads();

    const int q = tid & 31, cls = tid >> 5;
    for (int j = cls; j < ntok; j += 8) {
        const float* kr = kg + (size_t)j * HD;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < HD; ++d) acc = fmaf(Qs[q * (HD + 1) + d], kr[d], acc);
        S[q * sstride + j] = acc * scale;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {      // 4 waves x 8 rows
        float* row = S + r * sstride;
        float mx = -INFINITY;
        for (int j = lane; j < ntok; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < ntok; j += 64) {
            const float e = expf(row[j] - mx);
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < ntok; j += 64) row[j] *= inv;
    }
    __syncthreads();

    constexpr int DP = HD / 8;                       // head_dim columns per thread
    float o[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) o[i] = 0.f;
    for (int j = 0; j < ntok; ++j) {
        const float pj = S[q * sstride + j];
        const float* vr = vg + (size_t)j * HD + cls * DP;
#pragma unroll
        for (int i = 0; i < DP; ++i) o[i] = fmaf(pj, vr[i], o[i]);
    }
    if (q0 + q < ntok) {
        const int b = bh / heads, h = bh - b * heads;
        float* og = O + (((size_t)b * ntok + q0 + q) * heads + h) * HD + cls * DP;
#pragma unroll
        for (int i = 0; i < DP; ++i) og[i] = o[i];
    }
}

template <int HD>
static hipError_t attn_f32_launch(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                  int ntok, float scale, hipStream_t s) {
    const size_t lds = ((size_t)32 * (ntok + 1) + 32 * (HD + 1)) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(batch * heads, (ntok + 31) / 32);
    hipLaunchKernelGGL((attention_f32<HD>), grid, dim3(256), lds, s, q, k, v, out, ntok, heads, scale);
    return hipGetLastError();
}

hipError_t launch_attention_f32(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                int ntok, int hd, float scale, hipStream_t s) {
    switch (hd) {
        case 16: return attn_f32_launch<16>(q, k, v, out, batch, heads, ntok, scale, s);
        case 32: return attn_f32_launch<32>(q, k, v, out, batch, heads, ntok, scale, s);
        case 64: return attn_f32_launch<64>(q, k, v, out, batch, heads, ntok, scale, s);
        case 80: return attn_f32_launch<80>(q, k, v, out, batch, heads, ntok, scale, s);
        case 96: return attn_f32_launch<96>(q, k, v, out, batch, heads, ntok, scale, s);
        case 128: return attn_f32_launch<128>(q, k, v, out, batch, heads, ntok, scale, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vitk
