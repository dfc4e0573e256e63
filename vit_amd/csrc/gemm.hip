// bf16 MFMA GEMM family for the ViT forward path (gfx950 / CDNA4).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: activations [M,K], W: weights [N,K]
//
// which is what nn.Linear computes (reference call sites:
// models/transformer.py:15,17,36,38 and models/vision_transformer.py:18,28,30).
// One workgroup of 8 waves owns a 256(m) x 256(n) output tile and walks K in
// 64-wide steps through two LDS stages filled by LDS-DMA (global_load_lds).
// The epilogue variants fuse what the reference runs as separate passes: bias,
// exact GELU, the residual add, the head-major q/k/v split, and the
// cls/pos-embed bookkeeping of the patch embedding.  The patch embedding uses
// the same main loop with a gather loader that reads 16-byte runs of image
// rows directly (no im2col buffer).
#include <type_traits>

#include "vitk_common.h"
#include "vitk_kernels.h"
#include "gemm_common.h"

namespace vitk {

// LDS image of an operand tile: row r (0..255) occupies bytes [128r, 128r+128);
// its 16-byte chunk c (8 bf16 of K) is stored at slot c ^ ((r >> 1) & 7).  With
// this XOR the 16 lanes of every ds_read_b128 lane group of an MFMA fragment
// read (rows r..r+15, one chunk column) land on 16 distinct 16-byte slots of
// the 256-byte bank row.  LDS-DMA writes linearly (wave base + lane*16), so the
// XOR is applied to the per-lane SOURCE chunk and again on the read.

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, const f32x4& acc, int m, int n,
                                               const f32x4& bias4) {
    float v0 = acc[0] + bias4[0], v1 = acc[1] + bias4[1], v2 = acc[2] + bias4[2], v3 = acc[3] + bias4[3];
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_TANH_BF16) {
        if constexpr (EPI == EPI_BIAS_GELU_BF16) {
            f32x2 g01 = {v0, v1}, g23 = {v2, v3};              // the polynomial every GEMM kernel of the library uses
            gelu_erf_x4(g01, g23);
            v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
        }
        if constexpr (EPI == EPI_BIAS_TANH_BF16) {
            v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
        }
        u32x2 o = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n) = o;
    } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
        const size_t off = (size_t)m * p.ldo + n;
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + off);
        f32x4 o = {v0 + r[0], v1 + r[1], v2 + r[2], v3 + r[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = o;
    } else if constexpr (EPI == EPI_BIAS_F32) {
        f32x4 o = {v0, v1, v2, v3};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n) = o;
    } else if constexpr (EPI == EPI_QKV_BF16) {
        // column n of the fused projection = which*D + h*hd + e  (reshape(B,N,3,H,hd), transformer.py:42)
        const int D = p.heads * p.hd;
        const int which = n / D, rem = n - which * D;
        const int h = rem / p.hd, e = rem - h * p.hd;
        const int b = m / p.ntok, t = m - b * p.ntok;
        const size_t off = ((((size_t)which * p.batch + b) * p.heads + h) * p.ntok + t) * p.hd + e;
        u32x2 o = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + off) = o;
    } else if constexpr (EPI == EPI_PATCH_F32) {
        // patch row m = b*npatch + pidx lands at token 1 + pidx (cls token is row 0): vision_transformer.py:47-50
        const int b = m / p.npatch, pidx = m - b * p.npatch;
        const f32x4 pe = *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + pidx) * p.N + n);
        f32x4 o = {v0 + pe[0], v1 + pe[1], v2 + pe[2], v3 + pe[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) +
                                  ((size_t)b * p.ntok + 1 + pidx) * p.ldo + n) = o;
    }
}

// STAGING: 0 = LDS-DMA (global_load_lds, 16 B per lane), 1 = through registers.
template <int EPI, bool GATHER, int STAGING>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_256x256(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kStageBytes];   // 128 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- workgroup -> output tile.  Blocks are dealt round-robin over the 8 XCDs, so give
    // each XCD a contiguous run of tile ids (bijective for any grid size); consecutive ids
    // walk the n tiles of one activation row panel, which then stays in that XCD's L2.
    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging addresses: per operand tile each wave moves 4 x 1 KiB (8 rows x 128 B).
    const int srow = lane >> 3;                       // row within the 8-row piece
    const int sswz = (lane >> 4) + 4 * (wave & 1);    // ((tile row >> 1) & 7)
    const int schunk = (lane & 7) ^ sswz;             // source chunk for this lane's LDS slot
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + srow;
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;                 // rows past M re-read the last row; never stored
        if constexpr (GATHER) {
            const int b = gm / p.npatch, pidx = gm - b * p.npatch;
            const int ph = pidx / p.grid_w, pw = pidx - ph * p.grid_w;
            a_src[i] = reinterpret_cast<const bf16_t*>(p.A) +
                       ((size_t)b * 3 * p.img_h + ((size_t)ph << p.p_log2)) * p.img_w + ((size_t)pw << p.p_log2);
        } else {
            a_src[i] = reinterpret_cast<const bf16_t*>(p.A) + (size_t)gm * p.lda + schunk * 8;
        }
        w_src[i] = reinterpret_cast<const bf16_t*>(p.W) + (size_t)(n0 + row) * p.K + schunk * 8;
    }
    const int pmask = (1 << p.p_log2) - 1;
    const size_t plane = (size_t)p.img_h * p.img_w;

    auto a_koff = [&](int kt) -> size_t {
        if constexpr (GATHER) {
            // k = c*P*P + i*P + j  (Conv2d weight [D,3,P,P] flattened; vision_transformer.py:18)
            const int k = kt * BK + schunk * 8;
            const int c = k >> (2 * p.p_log2), i = (k >> p.p_log2) & pmask, j = k & pmask;
            return c * plane + (size_t)i * p.img_w + j;
        } else {
            return (size_t)kt * BK;
        }
    };

    u32x4 stg_a[4], stg_w[4];   // register staging only
    auto stage_issue = [&](int kt, int buf) {
        const size_t ka = a_koff(kt), kw = (size_t)kt * BK;
        if constexpr (STAGING == 0) {
            char* base = smem + buf * kStageBytes + wave * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(a_src[i] + ka),
                                                 (lds_void*)(base + i * 8192), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(w_src[i] + kw),
                                                 (lds_void*)(base + kTileBytes + i * 8192), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                stg_a[i] = *reinterpret_cast<const u32x4*>(a_src[i] + ka);
                stg_w[i] = *reinterpret_cast<const u32x4*>(w_src[i] + kw);
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if constexpr (STAGING == 1) {
            char* base = smem + buf * kStageBytes + wave * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(base + i * 8192) = stg_a[i];
                *reinterpret_cast<u32x4*>(base + kTileBytes + i * 8192) = stg_w[i];
            }
        }
    };

    // ---- MFMA fragment addresses.  Waves: 2 along m (128 rows each) x 4 along n (64 each).
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * kRowBytes + ((fq ^ ((fr >> 1) & 7)) << 4);   // k-step 0; k-step 1 = foff ^ 64
    const int a_frag = wm * 128 * kRowBytes + foff;
    const int w_frag = kTileBytes + wn * 64 * kRowBytes + foff;

    f32x4 acc[4][8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = p.K / BK;
    stage_issue(0, 0);
    stage_commit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) stage_issue(kt + 1, cur ^ 1);
        const char* st = smem + cur * kStageBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[8], wf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wf[nt] = *reinterpret_cast<const bf16x8*>(st + ((w_frag + nt * 16 * kRowBytes) ^ (kk << 6)));
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                af[mt] = *reinterpret_cast<const bf16x8*>(st + ((a_frag + mt * 16 * kRowBytes) ^ (kk << 6)));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    // operands swapped: the MFMA's row index is n, its column index is m, so each
                    // lane ends up with 4 consecutive n of one output row m.
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < ktiles) stage_commit(cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = .. + fr][n = .. + 4*fq + 0..3] for each (nt, mt).
    const int mw = m0 + wm * 128 + fr;
    const int nw = n0 + wn * 64 + fq * 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = nw + nt * 16;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = mw + mt * 16;
            if (m < p.M) epilogue_store<EPI>(p, acc[nt][mt], m, n, b4);
        }
    }
}


template <int EPI, bool GATHER>
static hipError_t launch_variant(const GemmArgs& a, int variant, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    // variant: 0 = default (two-group phased kernel where it applies, else the one-group pipelined
    // kernel); 1 = basic kernel, LDS-DMA staging; 3 = basic kernel, register staging; 4 = same as 0;
    // 5 = one-group pipelined kernel (2, the untiled kernel, is dispatched by the caller)
    if (variant == 3)
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 1>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else if (variant == 1)
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 0>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.K * 2 >= (1ull << 32) ||
             (GATHER && (size_t)a.batch * 3 * a.img_h * a.img_w * 2 >= (1ull << 32)) ||
             (size_t)(a.M + 256) * a.N * 4 >= (1ull << 32) ||
             (EPI == EPI_PATCH_F32 && (size_t)a.batch * a.ntok * a.N * 4 >= (1ull << 32)))
        // the persistent kernels address operands and outputs with 32-bit byte offsets
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 0>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else if (variant != 5 && EPI != EPI_PATCH_F32 && gemm_pp_supports(a, GATHER))
        return launch_gemm_pp(a, EPI, s);
    else
        return launch_gemm_pipe(a, EPI, GATHER, s);
    return hipGetLastError();
}

hipError_t launch_gemm_tiled(const GemmArgs& a, int epilogue, bool gather, int variant, hipStream_t s) {
    if (gather) {
        if (epilogue != EPI_PATCH_F32) return hipErrorInvalidValue;
        return launch_variant<EPI_PATCH_F32, true>(a, variant, s);
    }
    switch (epilogue) {
        case EPI_BIAS_BF16: return launch_variant<EPI_BIAS_BF16, false>(a, variant, s);
        case EPI_BIAS_GELU_BF16: return launch_variant<EPI_BIAS_GELU_BF16, false>(a, variant, s);
        case EPI_BIAS_GELU_DUAL_BF16:      // through the persistent kernels only (their shared epilogue)
            if (variant == 1 || variant == 3 || (size_t)(a.M + 256) * a.N * 4 >= (1ull << 32)) return hipErrorInvalidValue;
            if (variant != 5 && gemm_pp_supports(a, false)) return launch_gemm_pp(a, EPI_BIAS_GELU_DUAL_BF16, s);
            return launch_gemm_pipe(a, EPI_BIAS_GELU_DUAL_BF16, false, s);
        case EPI_MUL_DGELU_BF16:
        case EPI_MUL_DTANH_BF16:           // through the persistent kernels only (their shared epilogue)
            if ((size_t)(a.M + 256) * a.N * 4 >= (1ull << 32) || (size_t)a.M * a.lda * 2 >= (1ull << 32) ||
                (size_t)a.N * a.K * 2 >= (1ull << 32))
                return hipErrorInvalidValue;
            if (variant != 5 && gemm_pp_supports(a, false)) return launch_gemm_pp(a, epilogue, s);
            return launch_gemm_pipe(a, epilogue, false, s);
        case EPI_RESID_LN_F32:
        case EPI_RESID_LN_FP:
        case EPI_RESID_LN_PP:
        case EPI_RESID_PF:
        case EPI_LN_GELU_BF16:
        case EPI_LN_QKV_BF16:              // the two-group persistent kernel only
            if ((size_t)(a.M + 256) * a.N * 4 >= (1ull << 32) || (size_t)a.M * a.lda * 2 >= (1ull << 32) ||
                (size_t)a.N * a.K * 2 >= (1ull << 32) || !gemm_pp_supports(a, false))
                return hipErrorInvalidValue;
            return launch_gemm_pp(a, epilogue, s);
        case EPI_BIAS_RESID_F32: return launch_variant<EPI_BIAS_RESID_F32, false>(a, variant, s);
        case EPI_QKV_BF16: return launch_variant<EPI_QKV_BF16, false>(a, variant, s);
        case EPI_BIAS_F32: return launch_variant<EPI_BIAS_F32, false>(a, variant, s);
        case EPI_BIAS_TANH_BF16: return launch_variant<EPI_BIAS_TANH_BF16, false>(a, variant, s);
        case EPI_PATCH_F32: return launch_variant<EPI_PATCH_F32, false>(a, variant, s);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Small / irregular GEMM: operands straight from global memory (they are
// L2-resident at these sizes), every bound checked.  Used for the cls-row head
// (pre_logits + tanh, head: vision_transformer.py:53-55), whose N = num_classes
// is not a multiple of the big tile, and for any shape the tiled kernel rejects.
// One wave computes 16(m) x 64(n); a block is 4 waves stacked along m.
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_small(const GemmArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = (blockIdx.y * 4 + wave) * 16, n0 = blockIdx.x * 64;
    if (m0 >= p.M) return;
    int am = m0 + fr;
    am = am < p.M ? am : p.M - 1;
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (size_t)am * p.lda + fq * 8;
    const bf16_t* wp[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int wr = n0 + nt * 16 + fr;
        wr = wr < p.N ? wr : p.N - 1;
        wp[nt] = reinterpret_cast<const bf16_t*>(p.W) + (size_t)wr * p.K + fq * 8;
    }
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.K; k += 32) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap + k);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp[nt] + k);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[nt], 0, 0, 0);
        }
    }
    const int m = m0 + fr;
    if (m >= p.M) return;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + nt * 16 + fq * 4 + j;
            if (n >= p.N) continue;
            float v = acc[nt][j] + (p.bias ? p.bias[n] : 0.f);
            if constexpr (EPI == EPI_BIAS_GELU_BF16) {
                // the same polynomial as the tiled kernels' epilogue, so that a row's activations do not
                // depend on which kernel the batch size selects
                f32x2 g0 = {v, 0.f}, g1 = {0.f, 0.f};
                gelu_erf_x4(g0, g1);
                v = g0[0];
            }
            if constexpr (EPI == EPI_BIAS_TANH_BF16) v = tanhf(v);
            const size_t off = (size_t)m * p.ldo + n;
            if constexpr (EPI == EPI_MUL_DGELU_BF16 || EPI == EPI_MUL_DTANH_BF16) {
                // the product rounded to bf16 (what a bf16 matmul hands autograd) times the derivative at aux
                const float ax = (float)reinterpret_cast<const bf16_t*>(p.aux)[off];
                const float d = EPI == EPI_MUL_DGELU_BF16 ? dgelu_erf_x2(f32x2{ax, 0.f})[0] : fmaf(-ax, ax, 1.0f);
                v = (float)(bf16_t)v * d;
            }
            if constexpr (EPI == EPI_BIAS_F32) {
                reinterpret_cast<float*>(p.out)[off] = v;
            } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
                reinterpret_cast<float*>(p.out)[off] = v + p.resid[off];
            } else {
                reinterpret_cast<bf16_t*>(p.out)[off] = (bf16_t)v;
            }
        }
    }
}

hipError_t launch_gemm_small(const GemmArgs& a, int epilogue, hipStream_t s) {
    dim3 grid((a.N + 63) / 64, (a.M + 63) / 64), block(256);
    switch (epilogue) {
        case EPI_BIAS_BF16: hipLaunchKernelGGL(gemm_bf16_small<EPI_BIAS_BF16>, grid, block, 0, s, a); break;
        case EPI_BIAS_GELU_BF16: hipLaunchKernelGGL(gemm_bf16_small<EPI_BIAS_GELU_BF16>, grid, block, 0, s, a); break;
        case EPI_BIAS_RESID_F32: hipLaunchKernelGGL(gemm_bf16_small<EPI_BIAS_RESID_F32>, grid, block, 0, s, a); break;
        case EPI_BIAS_F32: hipLaunchKernelGGL(gemm_bf16_small<EPI_BIAS_F32>, grid, block, 0, s, a); break;
        case EPI_BIAS_TANH_BF16: hipLaunchKernelGGL(gemm_bf16_small<EPI_BIAS_TANH_BF16>, grid, block, 0, s, a); break;
        case EPI_MUL_DGELU_BF16: hipLaunchKernelGGL(gemm_bf16_small<EPI_MUL_DGELU_BF16>, grid, block, 0, s, a); break;
        case EPI_MUL_DTANH_BF16: hipLaunchKernelGGL(gemm_bf16_small<EPI_MUL_DTANH_BF16>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vitk
