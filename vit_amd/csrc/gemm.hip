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

namespace vitk {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kGemmThreads = 512;
constexpr int kTileBytes = BM * BK * 2;       // one operand tile, 32 KiB
constexpr int kStageBytes = 2 * kTileBytes;   // A tile + W tile
constexpr int kRowBytes = BK * 2;             // 128 B per tile row

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

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
            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
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


// ---------------------------------------------------------------------------
// Pipelined kernel (the default).  Same tile, LDS image and wave layout as the
// kernel above, with three changes:
//  * MFMA fragments are double-buffered in registers: while the 32 MFMAs of one
//    32-deep k-step run, the ds_read_b128s of the next k-step are in flight, so
//    LDS latency never sits in front of the matrix pipe; one barrier per K tile.
//  * Operand addresses are 32-bit lane offsets from a wave-uniform base that
//    advances by one K tile per step (scalar add), not 64-bit lane pointers.
//  * The epilogue goes through a wave-private 4 KiB LDS window: accumulators are
//    written there in MFMA layout and read back as 16-byte row pieces, so every
//    global store (and residual / pos-embed load) instruction covers whole
//    128-byte lines instead of 32-byte fragments of 16 different rows.
// LDS: 2 stages x 64 KiB + 8 waves x 4 KiB = 160 KiB (one workgroup per CU).
// ---------------------------------------------------------------------------
constexpr int kEpiBytesPerWave = 4096;
constexpr int kPipeLdsBytes = 2 * kStageBytes + 8 * kEpiBytesPerWave;

// workgroup id -> (tile_m, tile_n).  Blocks are dealt round-robin over the 8 XCDs, so mode 0/3
// first give each XCD a contiguous run of ids (bijective for any grid size).
__device__ __forceinline__ void tile_of_block(int order, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int mode = order & 15;
    int wg = b;
    if (mode != 1) {
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    if (mode == 3) {
        // n tiles in groups of gn: ids walk n inside the group, then m, then the next group, so a
        // group's weight panels stay in the XCD's L2 while the activation panels stream past.
        int gn = order >> 4;
        gn = gn < 1 ? 1 : (gn > tiles_n ? tiles_n : gn);
        const int per_group = gn * tiles_m;
        int g = wg / per_group, rem = wg - g * per_group;
        const int full = tiles_n / gn;             // groups of full width; a narrower one may follow
        int width = gn;
        if (g >= full) { g = full; rem = wg - full * per_group; width = tiles_n - full * gn; }
        tile_m = rem / width;
        tile_n = g * gn + (rem - tile_m * width);
    } else {
        tile_m = wg / tiles_n;
        tile_n = wg - tile_m * tiles_n;
    }
}

// rows advance by `step` inside one image of `per` rows: (b, t) -> (b', t')
__device__ __forceinline__ void advance_row(int& b, int& t, int step, int per) {
    t += step;
    while (t >= per) { t -= per; ++b; }
}

template <int EPI, bool GATHER>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_pipe(const GemmArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[kPipeLdsBytes];   // 160 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    int tile_m, tile_n;
    tile_of_block(p.order, tiles_m, tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging: per operand tile each wave moves 4 x 1 KiB (8 rows x 128 B) per K tile.
    const int srow = lane >> 3;
    const int sswz = (lane >> 4) + 4 * (wave & 1);    // ((tile row >> 1) & 7)
    const int schunk = (lane & 7) ^ sswz;             // source chunk that belongs in this lane's LDS slot
    uint32_t a_off[4], w_off[4];                      // byte offsets from the uniform bases
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + srow;
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;                 // rows past M re-read the last row; never stored
        if constexpr (GATHER) {
            const int b = gm / p.npatch, pidx = gm - b * p.npatch;
            const int ph = pidx / p.grid_w, pw = pidx - ph * p.grid_w;
            a_off[i] = (uint32_t)((((size_t)b * 3 * p.img_h + ((size_t)ph << p.p_log2)) * p.img_w +
                                   ((size_t)pw << p.p_log2)) * 2);
        } else {
            a_off[i] = (uint32_t)(((size_t)gm * p.lda + schunk * 8) * 2);
        }
        w_off[i] = (uint32_t)(((size_t)(n0 + row) * p.K + schunk * 8) * 2);
    }
    const char* a_base = reinterpret_cast<const char*>(p.A);
    const char* w_base = reinterpret_cast<const char*>(p.W);
    const int pmask = (1 << p.p_log2) - 1;
    const uint32_t plane = (uint32_t)p.img_h * p.img_w;

    auto stage_issue = [&](int kt, int buf) {
        char* dst = smem + buf * kStageBytes + wave * 1024;
        uint32_t ka;           // per-lane byte offset of this K tile inside the activation operand
        const char* ab = a_base;
        if constexpr (GATHER) {
            // k = c*P*P + i*P + j  (Conv2d weight [D,3,P,P] flattened; vision_transformer.py:18)
            const int k = kt * BK + schunk * 8;
            const int c = k >> (2 * p.p_log2), i = (k >> p.p_log2) & pmask, j = k & pmask;
            ka = (c * plane + (uint32_t)i * p.img_w + j) * 2;
        } else {
            ka = 0;
            ab += (size_t)kt * (BK * 2);
        }
        const char* wb = w_base + (size_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void*)(ab + (a_off[i] + ka)), (lds_void*)(dst + i * 8192), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(wb + w_off[i]), (lds_void*)(dst + kTileBytes + i * 8192),
                                             16, 0, 0);
        }
    };

    // ---- MFMA fragments.  Waves: 2 along m (128 rows each) x 4 along n (64 each).
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * kRowBytes + ((fq ^ ((fr >> 1) & 7)) << 4);   // k-step 0; k-step 1 = foff ^ 64
    const int a_frag = wm * 128 * kRowBytes + foff;
    const int w_frag = kTileBytes + wn * 64 * kRowBytes + foff;

    bf16x8 af0[8], wf0[4], af1[8], wf1[4];
    auto load_frags = [&](bf16x8 (&af)[8], bf16x8 (&wf)[4], int buf, int kk) {
        const char* st = smem + buf * kStageBytes;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wf[nt] = *reinterpret_cast<const bf16x8*>(st + ((w_frag + nt * 16 * kRowBytes) ^ (kk << 6)));
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            af[mt] = *reinterpret_cast<const bf16x8*>(st + ((a_frag + mt * 16 * kRowBytes) ^ (kk << 6)));
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const bf16x8 (&af)[8], const bf16x8 (&wf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                // operands swapped: MFMA row index = n, column index = m, so a lane ends up with
                // 4 consecutive n of one output row m.
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int ktiles = p.K / BK;
    // optional L2 prefetch pf_dist K tiles beyond the LDS-DMA refill (order bits 12..14; not for the gather loader)
    const int pf_dist = GATHER ? 0 : ((p.order >> 12) & 7);
    const char* pf_ptr;
    {
        const int t = tid & 255;
        int gm = m0 + t;
        gm = gm < p.M ? gm : p.M - 1;
        pf_ptr = tid < 256 ? a_base + (size_t)gm * p.lda * 2 : w_base + (size_t)(n0 + t) * p.K * 2;
        pf_ptr += (lane & 31) * 4;
    }
    stage_issue(0, 0);
    if (ktiles > 1) {
        stage_issue(1, 1);
        __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): the 8 loads of stage 1 may still fly
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
    __builtin_amdgcn_s_barrier();
    load_frags(af0, wf0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0)

    // s_waitcnt through the builtin so that the compiler's own wait bookkeeping sees it: with a wait it
    // cannot see (inline asm) it puts a full lgkmcnt(0) in front of each MFMA cluster, right after the
    // next fragments' ds_reads were issued, and the overlap is gone.
    constexpr int kWaitLgkm0 = 0xC07F;        // lgkmcnt(0), vmcnt / expcnt untouched
    constexpr int kWaitVmLgkm0 = 0x0070;      // vmcnt(0) lgkmcnt(0)
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        // phase A: fragments of k-step 1 in flight under the MFMAs of k-step 0
        load_frags(af1, wf1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        // every wave is done reading stage `cur`, and the next K tile has landed in the other stage
        __builtin_amdgcn_s_waitcnt(kWaitVmLgkm0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // phase B: refill stage `cur` two K tiles ahead, fetch the next tile's first fragments
        if (kt + 2 < ktiles && !(p.order & 0x100)) stage_issue(kt + 2, cur);   // 0x100: timing probe, no refills
        if (pf_dist && kt + 2 + pf_dist < ktiles) {
            // L2 prefetch: touch one dword of each of the 512 lines (256 activation rows + 256 weight
            // rows, 128 B per row per K tile) that the LDS-DMA of K tile kt+2+pf_dist will read.
            // The 4 bytes go to a dead spot of the epilogue window; nobody reads them.
            __builtin_amdgcn_global_load_lds((gbl_void*)(pf_ptr + (size_t)(kt + 2 + pf_dist) * (BK * 2)),
                                             (lds_void*)(smem + 2 * kStageBytes + wave * kEpiBytesPerWave), 4, 0, 0);
        }
        if (kt + 1 < ktiles) load_frags(af0, wf0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kWaitLgkm0);   // the k-step-0 fragments had 32 MFMAs to arrive
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue through the wave-private LDS window.
    // acc[nt][mt][j] = C[m = mw + 16 mt + fr][n = nw + 16 nt + 4 fq + j]
    char* win = smem + 2 * kStageBytes + wave * kEpiBytesPerWave;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;
    f32x4 b4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        b4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4[nt] = *reinterpret_cast<const f32x4*>(p.bias + nw + nt * 16 + fq * 4);
    }

    // Tiles that lie wholly inside M (all but the last row of tiles) take the unmasked copy, which
    // has no per-row branches, so the compiler can batch the loads and stores of a pass.
    auto epilogue = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_TANH_BF16 ||
                                   EPI == EPI_QKV_BF16);
        if constexpr (OUT_BF16) {
            // window = 32 rows x 128 B.  8-byte piece (row r, chunk c, half h) lives at
            // 128 r + 16 (c ^ (r & 7)) + 8 (h ^ ((r >> 3) & 1)): conflict-free for the MFMA-layout
            // writes (16 rows, one column) and for the row-wise 16-byte reads.
            const int rc = lane >> 3, cc = lane & 7;     // read-back role: row within 8, 16-byte chunk
            size_t col_off;                              // element offset contributed by the column
            if constexpr (EPI == EPI_QKV_BF16) {
                const int D = p.heads * p.hd;
                const int n = nw + cc * 8;
                const int which = n / D, rem = n - which * D;
                const int h = rem / p.hd, e = rem - h * p.hd;
                col_off = (((size_t)which * p.batch * p.heads + h) * p.ntok) * p.hd + e;
            } else {
                col_off = (size_t)(nw + cc * 8);
            }
    #pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
    #pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const int r = m2 * 16 + fr;
    #pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f32x4 a = acc[nt][pass * 2 + m2];
                        float v0 = a[0] + b4[nt][0], v1 = a[1] + b4[nt][1], v2 = a[2] + b4[nt][2], v3 = a[3] + b4[nt][3];
                        if constexpr (EPI == EPI_BIAS_GELU_BF16) {
                            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
                        }
                        if constexpr (EPI == EPI_BIAS_TANH_BF16) {
                            v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
                        }
                        const int c = nt * 2 + (fq >> 1), h = fq & 1;
                        u32x2 o = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
                        *reinterpret_cast<u32x2*>(win + r * 128 + ((c ^ (r & 7)) << 4) + ((h ^ ((r >> 3) & 1)) << 3)) = o;
                    }
                }
                int rb = 0, rt = 0;
                if constexpr (EPI == EPI_QKV_BF16) {
                    const int mrow = mw + pass * 32 + rc;
                    rb = mrow / p.ntok;
                    rt = mrow - rb * p.ntok;
                }
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 8 + rc;
                    u32x4 v = *reinterpret_cast<const u32x4*>(win + r * 128 + ((cc ^ (r & 7)) << 4));
                    if (i & 1) v = u32x4{v[2], v[3], v[0], v[1]};     // rows 8..15 and 24..31 hold swapped halves
                    const int m = mw + pass * 32 + r;
                    size_t off;
                    if constexpr (EPI == EPI_QKV_BF16) {
                        off = ((size_t)rb * p.heads * p.ntok + rt) * p.hd + col_off;
                        advance_row(rb, rt, 8, p.ntok);
                    } else {
                        off = (size_t)m * p.ldo + col_off;
                    }
                    if (!MASKED || m < p.M) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + off) = v;
                }
            }
        } else {
            // window = 16 rows x 256 B; 16-byte chunk c of row r lives at 256 r + 16 (c ^ (r & 15)).
            // The residual / pos-embed addend of a half tile (4 m-tiles = 16 loads per lane) is requested
            // up front, before any store of that half: out may alias resid, so loads placed after stores
            // would be kept behind them, one memory round trip per 16 bytes.
            const int rc = lane >> 4, cc = lane & 15;    // read-back role: row within 4, 16-byte chunk
            const int ncol = nw + cc * 4;
            constexpr bool ADDEND = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_PATCH_F32);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 add[4][4];
                uint32_t orow[4][4];     // output row of each piece (rows < 2^32 / ldo is checked by the launcher)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mt = half * 4 + q;
                    int rb = 0, rt = 0;
                    if constexpr (EPI == EPI_PATCH_F32) {
                        int mrow = mw + mt * 16 + rc;
                        mrow = mrow < p.M ? mrow : p.M - 1;
                        rb = mrow / p.npatch;
                        rt = mrow - rb * p.npatch;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int m = mw + mt * 16 + i * 4 + rc;
                        if (MASKED) m = m < p.M ? m : p.M - 1;      // loads stay in bounds; stores are masked below
                        if constexpr (EPI == EPI_PATCH_F32) {
                            // patch row (b, pidx) lands at token 1 + pidx (the cls token is row 0) and takes
                            // that token's position embedding: vision_transformer.py:47-50
                            orow[q][i] = (uint32_t)(rb * p.ntok + 1 + rt);
                            add[q][i] = *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + rt) * p.N + ncol);
                            advance_row(rb, rt, 4, p.npatch);
                        } else {
                            if constexpr (EPI == EPI_BIAS_RESID_F32)
                                add[q][i] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldo + ncol);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mt = half * 4 + q;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f32x4 a = acc[nt][mt];
                        f32x4 o = {a[0] + b4[nt][0], a[1] + b4[nt][1], a[2] + b4[nt][2], a[3] + b4[nt][3]};
                        *reinterpret_cast<f32x4*>(win + fr * 256 + (((nt * 4 + fq) ^ fr) << 4)) = o;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = i * 4 + rc;
                        f32x4 v = *reinterpret_cast<const f32x4*>(win + r * 256 + ((cc ^ r) << 4));
                        if constexpr (ADDEND) v = v + add[q][i];
                        const int m = mw + mt * 16 + r;
                        if (!MASKED || m < p.M)
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) +
                                                      (size_t)(EPI == EPI_PATCH_F32 ? orow[q][i] : (uint32_t)m) * p.ldo + ncol) = v;
                    }
                }
            }
        }
    };
    if (m0 + BM <= p.M) epilogue(std::false_type{});
    else epilogue(std::true_type{});
}

template <int EPI, bool GATHER>
static hipError_t launch_variant(const GemmArgs& a, int variant, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    // variant: 0 = pipelined kernel (default); 1 = basic kernel, LDS-DMA staging; 3 = basic kernel,
    // register staging (2, the untiled kernel, is dispatched by the caller)
    if (variant == 3)
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 1>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else if (variant == 1)
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 0>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (GATHER && (size_t)a.batch * 3 * a.img_h * a.img_w * 2 >= (1ull << 32)))
        // the pipelined kernel addresses the activation operand with 32-bit byte offsets
        hipLaunchKernelGGL((gemm_bf16_256x256<EPI, GATHER, 0>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_bf16_pipe<EPI, GATHER>), dim3(tiles), dim3(kGemmThreads), 0, s, a);
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
        case EPI_BIAS_RESID_F32: return launch_variant<EPI_BIAS_RESID_F32, false>(a, variant, s);
        case EPI_QKV_BF16: return launch_variant<EPI_QKV_BF16, false>(a, variant, s);
        case EPI_BIAS_F32: return launch_variant<EPI_BIAS_F32, false>(a, variant, s);
        case EPI_BIAS_TANH_BF16: return launch_variant<EPI_BIAS_TANH_BF16, false>(a, variant, s);
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
            if constexpr (EPI == EPI_BIAS_GELU_BF16) v = gelu_erf(v);
            if constexpr (EPI == EPI_BIAS_TANH_BF16) v = tanhf(v);
            const size_t off = (size_t)m * p.ldo + n;
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
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vitk
