// Persistent, software-pipelined bf16 MFMA GEMM (the default tiled kernel; gfx950 / CDNA4).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]   (+ fused epilogue, see gemm_common.h / vitk_kernels.h)
//
// One workgroup (8 waves) per CU walks a list of output tiles.  A tile is BMe x 256 with
// BMe = 16 * mtot rows (mtot <= 16, chosen by the launcher so that the tile count is close to a
// multiple of the CU count: no half-empty last round).  K is walked in 64-wide tiles through two
// 64 KiB LDS stages filled by LDS-DMA; the sequence of K tiles runs on ACROSS output tiles, so
// the first two K tiles of the next output tile are already in flight while the current tile's
// epilogue converts and stores its accumulators (through a separate 32 KiB LDS window).
//
// Per K tile and wave: 2 x { 12 ds_read_b128 of the next k-step's fragments, 32 MFMA 16x16x32 of
// the current one }, one s_barrier.  Waves are 2 (m) x 4 (n); the two waves that share a SIMD
// (w and w + 4) take the two m halves, so an odd mtot still loads every SIMD equally.
#include "gemm_plan.h"

namespace vitk {

template <int EPI, bool GATHER>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_pipe(const GemmArgs p, const PipePlan pl) {
    __shared__ __attribute__((aligned(1024))) char smem[kPipeLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's tiles.  Blocks are dealt round-robin over the 8 XCDs; XCD x owns a
    // contiguous chunk of the tile order and its workgroups take the chunk's tiles round-robin,
    // so at any time an XCD works on ~32 consecutive tiles (shared panels hit in its L2).
    const int total_tiles = pl.tiles_m * pl.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slots = ((int)gridDim.x - xcd + 7) >> 3;          // workgroups on this XCD
    const int cq = total_tiles >> 3, cr = total_tiles & 7;
    const int chunk0 = xcd * cq + (xcd < cr ? xcd : cr);
    const int chunk_n = cq + (xcd < cr ? 1 : 0);
    const int my_tiles = slot < chunk_n ? (chunk_n - slot + slots - 1) / slots : 0;
    if (my_tiles == 0) return;                                   // whole workgroup leaves together

    const int ktiles = p.K / BK;
    const int total_steps = my_tiles * ktiles;
    const int bme = pl.mtot * 16;
    const int a_pieces = pl.mtot * 2;                            // 8-row LDS-DMA pieces of the activation tile

    // ---- LDS-DMA issue stream (runs two K tiles ahead of the compute stream)
    const int srow = lane >> 3;
    const int sswz = (lane >> 4) + 4 * (wave & 1);               // ((tile row >> 1) & 7) of this lane's rows
    const int schunk = (lane & 7) ^ sswz;                        // source chunk that belongs in this lane's LDS slot
    const char* a_base = reinterpret_cast<const char*>(p.A);
    const char* w_base = reinterpret_cast<const char*>(p.W);
    const int pmask = (1 << p.p_log2) - 1;
    const uint32_t plane = (uint32_t)p.img_h * p.img_w;
    int iss_step = 0, iss_kt = 0, iss_tile = 0, iss_m0, iss_n0;
    {
        int tm, tn;
        tile_coords(pl, chunk0 + slot, tm, tn);
        iss_m0 = tm * bme;
        iss_n0 = tn * BN;
    }
    auto issue_next = [&]() {
        char* dst = smem + (iss_step & 1) * kStageBytes + wave * 1024;
        uint32_t ka = 0;
        const char* ab = a_base;
        if constexpr (GATHER) {
            // k = c*P*P + i*P + j  (Conv2d weight [D,3,P,P] flattened; vision_transformer.py:18)
            const int k = iss_kt * BK + schunk * 8;
            const int c = k >> (2 * p.p_log2), i = (k >> p.p_log2) & pmask, j = k & pmask;
            ka = (c * plane + (uint32_t)i * p.img_w + j) * 2;
        } else {
            ab += (size_t)iss_kt * (BK * 2);
        }
        const char* wb = w_base + (size_t)iss_kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + wave) * 8 + srow;
            if (i * 8 + wave < a_pieces) {                       // wave-uniform: pieces past the tile height are skipped
                int gm = iss_m0 + row;
                gm = gm < p.M ? gm : p.M - 1;                    // rows past M re-read the last row; never stored
                uint32_t ao;
                if constexpr (GATHER) {
                    const int b = gm / p.npatch, pidx = gm - b * p.npatch;
                    const int ph = pidx / p.grid_w, pw = pidx - ph * p.grid_w;
                    ao = (uint32_t)((((size_t)b * 3 * p.img_h + ((size_t)ph << p.p_log2)) * p.img_w +
                                     ((size_t)pw << p.p_log2)) * 2) + ka;
                } else {
                    ao = (uint32_t)(((size_t)gm * p.lda + schunk * 8) * 2);
                }
                __builtin_amdgcn_global_load_lds((gbl_void*)(ab + ao), (lds_void*)(dst + i * 8192), 16, 0, 0);
            }
            const uint32_t wo = (uint32_t)(((size_t)(iss_n0 + row) * p.K + schunk * 8) * 2);
            __builtin_amdgcn_global_load_lds((gbl_void*)(wb + wo), (lds_void*)(dst + kTileBytes + i * 8192), 16, 0, 0);
        }
        ++iss_step;
        if (++iss_kt == ktiles) {
            iss_kt = 0;
            if (++iss_tile < my_tiles) {
                int tm, tn;
                tile_coords(pl, chunk0 + slot + iss_tile * slots, tm, tn);
                iss_m0 = tm * bme;
                iss_n0 = tn * BN;
            }
        }
    };

    // ---- MFMA fragments.  wm = m half (the two waves of a SIMD differ in wm), wn = 64-column strip.
    const int wm = wave >> 2, wn = wave & 3;
    const int mt0 = (pl.mtot + 1) >> 1;
    const int my_mt = wm ? pl.mtot - mt0 : mt0;                  // m-tiles of this wave (wave-uniform)
    const int my_row0 = wm ? mt0 * 16 : 0;
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * kRowBytes + ((fq ^ ((fr >> 1) & 7)) << 4);   // k-step 0; k-step 1 = foff ^ 64
    const int a_frag = my_row0 * kRowBytes + foff;
    const int w_frag = kTileBytes + wn * 64 * kRowBytes + foff;

    bf16x8 af0[8], wf0[4], af1[8], wf1[4];
    auto load_frags = [&](bf16x8 (&af)[8], bf16x8 (&wf)[4], int buf, int kk) {
        const char* st = smem + buf * kStageBytes;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wf[nt] = *reinterpret_cast<const bf16x8*>(st + ((w_frag + nt * 16 * kRowBytes) ^ (kk << 6)));
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            if (mt < my_mt)
                af[mt] = *reinterpret_cast<const bf16x8*>(st + ((a_frag + mt * 16 * kRowBytes) ^ (kk << 6)));
    };

    f32x4 acc[4][8];
    auto mma = [&](const bf16x8 (&af)[8], const bf16x8 (&wf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            if (mt < my_mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    // operands swapped: MFMA row index = n, column index = m, so a lane ends up with
                    // 4 consecutive n of one output row m.
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };

    // s_waitcnt goes through the builtin so that the compiler's own wait bookkeeping sees it: with a
    // wait it cannot see (inline asm) it puts a full lgkmcnt(0) in front of each MFMA cluster, right
    // after the next fragments' ds_reads were issued, and the overlap is gone.
    constexpr int kWaitLgkm0 = 0xC07F;        // lgkmcnt(0), vmcnt / expcnt untouched
    constexpr int kWaitVmLgkm0 = 0x0070;      // vmcnt(0) lgkmcnt(0)

    issue_next();
    if (total_steps > 1) issue_next();
    __builtin_amdgcn_s_waitcnt(kWaitVmLgkm0);
    __builtin_amdgcn_s_barrier();
    load_frags(af0, wf0, 0, 0);
    __builtin_amdgcn_s_waitcnt(kWaitLgkm0);

    char* win = smem + 2 * kStageBytes + wave * kEpiBytesPerWave;
    int step = 0;
    for (int t = 0; t < my_tiles; ++t) {
        int tm, tn;
        tile_coords(pl, chunk0 + slot + t * slots, tm, tn);
        const int m0 = tm * bme, n0 = tn * BN;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < ktiles; ++kt, ++step) {
            const int cur = step & 1;
            // phase A: fragments of k-step 1 in flight under the MFMAs of k-step 0
            load_frags(af1, wf1, cur, 1);
            mma(af0, wf0);
            // every wave is done reading stage `cur`, and K tile step+1 has landed in the other stage
            __builtin_amdgcn_s_waitcnt(kWaitVmLgkm0);
            __builtin_amdgcn_s_barrier();
            // phase B: refill stage `cur` with K tile step+2 (possibly the next output tile's)
            if (step + 2 < total_steps) issue_next();
            // fetch the first fragments of K tile step+1 when it belongs to this output tile
            if (kt + 1 < ktiles) load_frags(af0, wf0, cur ^ 1, 0);
            mma(af1, wf1);
            __builtin_amdgcn_s_waitcnt(kWaitLgkm0);   // those fragments had 32 MFMAs to arrive
        }

        // ---- epilogue (the next output tile's first two K tiles are in flight or landed meanwhile)
        const int mw = m0 + my_row0, nw = n0 + wn * 64;
        int row_limit = p.M - mw;
        row_limit = row_limit < my_mt * 16 ? row_limit : my_mt * 16;
        // An opaque copy of the lane id: everything the epilogue derives from it is then computed
        // here, per tile, instead of being hoisted above the K loop and held (or spilled) across it.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // one (masked) instantiation only: with a second, unmasked copy beside it hipcc hoists shared
        // address math in front of both and spills ~30 more registers around the epilogue
        epilogue_tile<EPI, true>(p, acc, win, mw, nw, my_mt, row_limit, lane_e);

        if (t + 1 < my_tiles) {
            // first fragments of the next output tile: its K tile 0 landed before the last barrier
            load_frags(af0, wf0, step & 1, 0);
            __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
        }
    }
}

template <int EPI, bool GATHER>
static hipError_t launch_pipe(const GemmArgs& a, hipStream_t s) {
    const int cus = num_cus();
    const PipePlan pl = make_plan(a, cus);
    const int tiles = pl.tiles_m * pl.tiles_n;
    const int grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL((gemm_bf16_pipe<EPI, GATHER>), dim3(grid), dim3(kGemmThreads), 0, s, a, pl);
    return hipGetLastError();
}

hipError_t launch_gemm_pipe(const GemmArgs& a, int epilogue, bool gather, hipStream_t s) {
    if (gather) return launch_pipe<EPI_PATCH_F32, true>(a, s);
    switch (epilogue) {
        case EPI_BIAS_BF16: return launch_pipe<EPI_BIAS_BF16, false>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_pipe<EPI_BIAS_GELU_BF16, false>(a, s);
        case EPI_BIAS_GELU_DUAL_BF16: return launch_pipe<EPI_BIAS_GELU_DUAL_BF16, false>(a, s);
        case EPI_BIAS_RESID_F32: return launch_pipe<EPI_BIAS_RESID_F32, false>(a, s);
        case EPI_QKV_BF16: return launch_pipe<EPI_QKV_BF16, false>(a, s);
        case EPI_BIAS_F32: return launch_pipe<EPI_BIAS_F32, false>(a, s);
        case EPI_BIAS_TANH_BF16: return launch_pipe<EPI_BIAS_TANH_BF16, false>(a, s);
        case EPI_MUL_DGELU_BF16: return launch_pipe<EPI_MUL_DGELU_BF16, false>(a, s);
        case EPI_MUL_DTANH_BF16: return launch_pipe<EPI_MUL_DTANH_BF16, false>(a, s);
        case EPI_PATCH_F32: return launch_pipe<EPI_PATCH_F32, false>(a, s);     // rows already gathered (im2col)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vitk
