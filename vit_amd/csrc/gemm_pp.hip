// Persistent bf16 MFMA GEMM with two wave groups running half a phase apart (gfx950 / CDNA4).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]   (+ fused epilogue, see gemm_common.h / vitk_kernels.h)
//
// One workgroup (8 waves) per CU walks a list of output tiles of 16*mtot x 256 (gemm_plan.h).  The
// waves form two groups of four: group g owns the upper / lower share of the tile's rows, wave wn of
// a group a 64-column strip, so the two waves that share a SIMD (w and w + 4) belong to different
// groups.  A K tile (64 deep) is worked in four phases, one quadrant (4 m-tiles x 2 n-tiles x
// K = 64 = 16 MFMA) of the wave's 128 x 64 sub-tile each:
//
//     load section : ds_read the fragments this quadrant still lacks, issue one class of LDS-DMA
//                    pieces of a later K tile, counted s_waitcnt vmcnt          -> s_barrier
//     MFMA section : s_waitcnt lgkmcnt(0), 16 MFMA                              -> s_barrier
//
// Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA section while
// its partner reads fragments and issues DMA: the matrix pipe never waits for a ds_read or for the
// issue cost of the DMA.
//
// LDS-DMA stream.  A K tile's 64 KiB are split in four classes of 16 KiB (2 DMA instructions per
// wave each), in the order the phases first need them:
//     class 0  A rows of the quadrants' upper half (m-tiles 0..3 of both groups)   read in phase 0
//     class 1  W rows of the first two n-tiles of every strip                      read in phase 0
//     class 2  W rows of the last two n-tiles of every strip                       read in phase 1
//     class 3  A rows of the lower half (m-tiles 4..7 of both groups)              read in phase 2
// Phase f (counted from the first K tile of the workgroup's first output tile) issues class f + 6
// of that sequence, which runs on across K tiles and across output tiles, and then waits until at
// most 8 DMA instructions are outstanding: classes up to f + 2 have landed, and the barrier that
// ends the load section publishes them to the phase that reads them, f + 1 or later.  A class
// overwrites the class 8 places earlier, whose last ds_read was retired (lgkmcnt(0) of both groups)
// at least two barriers before: W fragments of the first two n-tiles stay in registers for phase 3.
//
// The load sections hold no vector ALU work at all: while the partner wave issues MFMAs (at raised
// priority) every VALU instruction of the loading wave waits for a free issue slot, one per MFMA or
// so, and the section stretches past the MFMA section it is meant to hide under.  So the LDS stage
// is a compile-time constant (the K loop walks two K tiles per trip; K/64 must be even), which
// turns every fragment address into base register + immediate offset, and the per-lane part of the
// DMA source addresses is computed once per output tile; per K tile only scalar registers change.
//
// LDS map: [A stage 0][A stage 1][W stage 0][W stage 1][epilogue windows], 32 KiB each.  An A stage
// is [group 0: 128 rows][group 1: 128 rows] (group-local rows, so that fragment addresses do not
// depend on the tile height), a W stage 256 rows; 128 B per row, 16-byte chunk c of row r at slot
// c ^ ((r >> 1) & 7) (gemm_common.h).
#include <type_traits>

#include "gemm_plan.h"

namespace vitk {

// PROBE (timing experiments): 1 = no DMA after the prologue (wrong results), 2 = no epilogue (no results),
// 7 = the cycles of the six
// segments of every phase, summed over workgroup 0's K tiles by one wave of each group and written over
// the first elements of `out` (tools/gemm_stamps.py)
template <int EPI, int PROBE = 0>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_pp(const GemmArgs p, const PipePlan pl) {
    __shared__ __attribute__((aligned(1024))) char smem[kPipeLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's tiles: XCD x owns a contiguous chunk of the tile order, its workgroups
    // take the chunk's tiles round-robin (shared panels hit in the XCD's L2)
    const int total_tiles = pl.tiles_m * pl.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slots = ((int)gridDim.x - xcd + 7) >> 3;
    const int cq = total_tiles >> 3, cr = total_tiles & 7;
    const int chunk0 = xcd * cq + (xcd < cr ? xcd : cr);
    const int chunk_n = pl.mixed ? (pl.na + pl.nb) * pl.tiles_n : cq + (xcd < cr ? 1 : 0);
    const int my_tiles = slot < chunk_n ? (chunk_n - slot + slots - 1) / slots : 0;
    if (my_tiles == 0) return;                                   // whole workgroup leaves together

    const int ktiles = p.K / BK;
    const int total_classes = my_tiles * ktiles * 4;
    const int grp = wave >> 2, wn = wave & 3;
    // tile i of this workgroup -> first row, height in m-tiles (8..16: a group has 4..8), n-tile
    auto tile_geo = [&](int i, int& m_first, int& mtot, int& tile_n) {
        if (pl.mixed) {
            mixed_tile(pl, xcd, slot + i * slots, m_first, mtot, tile_n);
        } else {
            int tm;
            tile_coords(pl, chunk0 + slot + i * slots, tm, tile_n);
            mtot = pl.mtot;
            m_first = tm * mtot * 16;
        }
    };

    constexpr int kAStage = 2 * 128 * kRowBytes;                 // 32 KiB: both groups' rows of one stage
    constexpr int kWRegion = 2 * kAStage;                        // W stages follow the two A stages

    // ---- LDS-DMA issue stream.  Source address = scalar base (tensor + K offset of the K tile) +
    // per-lane byte offset of the lane's row and 16-byte chunk, the latter fixed per output tile.
    const int srow = lane >> 3;
    const int sswz = (lane >> 4) + 4 * (wave & 1);               // ((image row >> 1) & 7) of this lane's row
    const int schunk = (lane & 7) ^ sswz;                        // source chunk that belongs in this lane's slot
    const char* a_base = reinterpret_cast<const char*>(p.A);
    const char* w_base = reinterpret_cast<const char*>(p.W);
    const uint32_t w_half = (uint32_t)((size_t)128 * p.K * 2);   // 128 weight rows further down
    int iss_c = 0, iss_buf = 0, iss_kt = 0, iss_tile = 0;
    uint32_t a_off[2][2], w_off[2];                              // [upper / lower half][group], [first / last n-tiles]
    auto tile_offsets = [&](int i) {
        int m_first, mtot, tn;
        tile_geo(i, m_first, mtot, tn);
        const int n_first = tn * BN;
        const int rows_g0 = ((mtot + 1) >> 1) * 16, rows_g1 = (mtot - ((mtot + 1) >> 1)) * 16;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int piece = mh * 64 + wave * 8;                // first group-local row of this wave's piece
            // pieces past the group's share of the tile (wave-uniform) re-read the tile's first row, so
            // that every wave issues the same number of DMA instructions (the counted waits rely on it)
            int r0 = m_first + (piece < rows_g0 ? piece + srow : 0);
            int r1 = m_first + (piece < rows_g1 ? rows_g0 + piece + srow : 0);
            r0 = r0 < p.M ? r0 : p.M - 1;                        // rows past M re-read the last row; never stored
            r1 = r1 < p.M ? r1 : p.M - 1;
            a_off[mh][0] = (uint32_t)r0 * (uint32_t)(p.lda * 2) + schunk * 16;
            a_off[mh][1] = (uint32_t)r1 * (uint32_t)(p.lda * 2) + schunk * 16;
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const int piece = (wave >> 2) * 64 + nh * 32 + (wave & 3) * 8;     // W tile row of the piece
            w_off[nh] = (uint32_t)(n_first + piece + srow) * (uint32_t)(p.K * 2) + schunk * 16;
        }
    };
    tile_offsets(0);
    // one class (J = 0..3 as listed above) of the stream: 2 DMA instructions per wave, always
    auto issue_class = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const size_t kbytes = (size_t)iss_kt * (BK * 2);
        if constexpr (J == 0 || J == 3) {
            constexpr int mh = (J == 3) ? 1 : 0;
            char* dst = smem + iss_buf * kAStage + (mh * 64 + wave * 8) * kRowBytes;
            const char* src = a_base + kbytes;
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + a_off[mh][0]), (lds_void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + a_off[mh][1]), (lds_void*)(dst + 128 * kRowBytes), 16, 0, 0);
        } else {
            constexpr int nh = (J == 2) ? 1 : 0;
            char* dst = smem + kWRegion + iss_buf * kAStage + ((wave >> 2) * 64 + nh * 32 + (wave & 3) * 8) * kRowBytes;
            const char* src = w_base + kbytes;
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + w_off[nh]), (lds_void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + w_half + w_off[nh]), (lds_void*)(dst + 128 * kRowBytes), 16, 0, 0);
        }
        ++iss_c;
        if constexpr (J == 3) {
            iss_buf ^= 1;
            if (++iss_kt == ktiles) {
                iss_kt = 0;
                if (++iss_tile < my_tiles) tile_offsets(iss_tile);
            }
        }
    };
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;

    // s_waitcnt goes through the builtin so that the compiler's own wait bookkeeping sees it.
    constexpr int kWaitLgkm0 = 0xC07F;        // lgkmcnt(0), vmcnt / expcnt untouched
    constexpr int kWaitVm8 = 0x0F78;          // vmcnt(8)
    constexpr int kWaitVm0 = 0x0F70;          // vmcnt(0)
    // PROBE 7: s_memtime stamps (inline asm: the values are read only after one of the explicit lgkmcnt(0)
    // below); the cycles of the six segments of every phase are summed in scalar registers.
    uint64_t stp[6] = {0, 0, 0, 0, 0, 0};
    uint32_t seg_sum[4][6] = {};
    auto stamp_to = [](uint64_t& dst) { asm volatile("s_memtime %0" : "=s"(dst)); };
    auto stamp = [&](auto ic) {
        if constexpr (PROBE == 7) stamp_to(stp[decltype(ic)::value]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using S4 = std::integral_constant<int, 4>;
    using S5 = std::integral_constant<int, 5>;

    // the load section's DMA part: next class of the stream, then all but the 4 youngest classes landed
    auto stream = [&](auto jc) {
        if (iss_c < total_classes) {
            if constexpr (PROBE == 1) { ++iss_c; return; }
            issue_class(jc);
            stamp(S1{});
            __builtin_amdgcn_s_waitcnt(kWaitVm8);
        } else {
            stamp(S1{});
            __builtin_amdgcn_s_waitcnt(kWaitVm0);                // stream ran dry: everything has landed
        }
        stamp(S2{});
    };

    // ---- MFMA fragments: one base register per operand and k-step, everything else immediate offsets
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * kRowBytes + ((fq ^ ((fr >> 1) & 7)) << 4);   // k-step 0; k-step 1 = foff ^ 64
    const char* a_frag[2];
    const char* w_frag[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_frag[kk] = smem + grp * (128 * kRowBytes) + (foff ^ (kk << 6));
        w_frag[kk] = smem + kWRegion + wn * 64 * kRowBytes + (foff ^ (kk << 6));
    }

    bf16x8 af[4][2], wf0[2][2], wf1[2][2];
    f32x4 acc[4][8];
    int my_mt = 8;                            // m-tiles of this wave's group in the current output tile (4..8)

    auto load_w = [&](bf16x8 (&wf)[2][2], auto stc, int nh) {
        constexpr int stage = decltype(stc)::value;
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf[n2][kk] = *reinterpret_cast<const bf16x8*>(w_frag[kk] + stage * kAStage + (nh * 2 + n2) * 16 * kRowBytes);
    };
    auto load_a = [&](auto stc, int mh, bool masked) {
        constexpr int stage = decltype(stc)::value;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4)
            if (!masked || mh * 4 + m4 < my_mt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    af[m4][kk] = *reinterpret_cast<const bf16x8*>(a_frag[kk] + stage * kAStage + (mh * 4 + m4) * 16 * kRowBytes);
            }
    };
    // one quadrant: m-tiles mh*4 .. mh*4+3, n-tiles nh*2, nh*2+1, both k-steps.  Operands swapped
    // (MFMA row index = n, column index = m), so a lane ends up with 4 consecutive n of one row m.
    auto quadrant = [&](const bf16x8 (&wf)[2][2], auto mhc, auto nhc, bool masked, auto firstc) {
        constexpr int mh = decltype(mhc)::value, nh = decltype(nhc)::value;
        constexpr bool first = decltype(firstc)::value;      // first K tile of an output tile: accumulate onto 0
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        stamp(S3{});
        __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
        stamp(S4{});
        __builtin_amdgcn_s_setprio(1);
        if (!masked) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2)
                        acc[nh * 2 + n2][mh * 4 + m4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            wf[n2][kk], af[m4][kk], (first && kk == 0) ? zero : acc[nh * 2 + n2][mh * 4 + m4], 0, 0, 0);
        } else {
            // m-tiles past the wave's share are skipped (wave-uniform, monotonic: an early-exit chain)
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
                if (mh * 4 + m4 < my_mt) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2)
                            acc[nh * 2 + n2][mh * 4 + m4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                wf[n2][kk], af[m4][kk], (first && kk == 0) ? zero : acc[nh * 2 + n2][mh * 4 + m4], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (PROBE == 7) {
            // segments: [barrier after the previous M, L issue, vmcnt wait, barrier after L, lgkmcnt wait] of this
            // phase, then its MFMA issue
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int ph = mh * 2 + (mh ^ nh);
            const uint32_t t5 = (uint32_t)stp[5], t0 = (uint32_t)stp[0], t1 = (uint32_t)stp[1], t2 = (uint32_t)stp[2],
                           t3 = (uint32_t)stp[3], t4 = (uint32_t)stp[4];
            seg_sum[ph][0] += t0 - t5;
            seg_sum[ph][1] += t1 - t0;
            seg_sum[ph][2] += t2 - t1;
            seg_sum[ph][3] += t3 - t2;
            seg_sum[ph][4] += t4 - t3;
            __builtin_amdgcn_sched_barrier(0);
            stamp(S5{});
            __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
            seg_sum[ph][5] += (uint32_t)stp[5] - t4;
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // the four phases of one K tile held in LDS stage `stc`
    auto k_tile = [&](auto stc, auto firstc) {
        // phase 0: upper m-tiles x first n-tiles
        stamp(S0{});
        load_w(wf0, stc, 0);
        load_a(stc, 0, false);
        stream(J2{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        quadrant(wf0, C0{}, C0{}, false, firstc);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // phase 1: upper m-tiles x last n-tiles
        stamp(S0{});
        load_w(wf1, stc, 1);
        stream(J3{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        quadrant(wf1, C0{}, C1{}, false, firstc);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // phase 2: lower m-tiles x last n-tiles
        stamp(S0{});
        load_a(stc, 1, true);
        stream(J0{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        quadrant(wf1, C1{}, C1{}, true, firstc);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // phase 3: lower m-tiles x first n-tiles (fragments all in registers)
        stamp(S0{});
        stream(J1{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        quadrant(wf0, C1{}, C0{}, true, firstc);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: classes 0..5 of the stream (K tile 0 and half of K tile 1; K/64 >= 2)
    issue_class(J0{});
    issue_class(J1{});
    issue_class(J2{});
    issue_class(J3{});
    issue_class(J0{});
    issue_class(J1{});
    __builtin_amdgcn_s_waitcnt(kWaitVm8);
    __builtin_amdgcn_s_barrier();

    char* win = smem + 2 * kStageBytes + wave * kEpiBytesPerWave;
    for (int t = 0; t < my_tiles; ++t) {
        int m0, mtot_t, tn;
        tile_geo(t, m0, mtot_t, tn);
        const int n0 = tn * BN;
        const int mt0 = (mtot_t + 1) >> 1;                       // m-tiles of group 0
        my_mt = grp ? mtot_t - mt0 : mt0;
        const int my_row0 = grp ? mt0 * 16 : 0;
        if (grp) __builtin_amdgcn_s_barrier();                   // group 1 falls one barrier behind

        // K/64 is even, so every output tile starts in stage 0.  The first K tile's first k-step takes 0 as
        // its C operand: no accumulator clearing, and m-tiles past the wave's share stay untouched.
        if constexpr (PROBE == 3) {                              // timing probe: clear the accumulators instead
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < ktiles; kt += 2) {
                k_tile(C0{}, std::false_type{});
                k_tile(C1{}, std::false_type{});
            }
        } else {
            k_tile(C0{}, std::true_type{});
            k_tile(C1{}, std::false_type{});
            for (int kt = 2; kt < ktiles; kt += 2) {
                k_tile(C0{}, std::false_type{});
                k_tile(C1{}, std::false_type{});
            }
        }

        if (!grp) __builtin_amdgcn_s_barrier();                  // group 0 waits for group 1's last quadrant

        if constexpr (PROBE == 7) {
            if (blockIdx.x == 0 && (wave & 3) == 0 && t == my_tiles - 1 && my_tiles > 1 && lane == 0) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(p.out) + (wave >> 2) * 32;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int i = 0; i < 6; ++i) dst[ph * 6 + i] = seg_sum[ph][i];
                dst[24] = (uint32_t)(my_tiles * ktiles);
            }
        }

        // ---- epilogue, both groups together (the next tile's first K tiles are in flight or landed)
        const int mw = m0 + my_row0, nw = n0 + wn * 64;
        // rows end at M, and in the two-height plan at the end of the XCD's row range (its last panel may reach past it)
        int m_end = p.M;
        if (pl.mixed) m_end = m_end < (xcd + 1) * pl.xcd_units * 16 ? m_end : (xcd + 1) * pl.xcd_units * 16;
        int row_limit = m_end - mw;
        row_limit = row_limit < my_mt * 16 ? row_limit : my_mt * 16;
        // An opaque copy of the lane id: everything the epilogue derives from it is then computed
        // here, per tile, instead of being hoisted above the K loop and held (or spilled) across it.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if constexpr (PROBE == 2) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) asm volatile("" ::"v"(acc[nt][mt]));
        } else {
            epilogue_tile<EPI, true, 2>(p, acc, win, mw, nw, my_mt, row_limit, lane_e);
        }
    }
}

// K/64 must be even (the K loop is unrolled over the two LDS stages) and the operand is a plain
// row-major matrix: the caller routes everything else to the one-group pipelined kernel.
bool gemm_pp_supports(const GemmArgs& a, bool gather) { return !gather && (a.K / BK) % 2 == 0 && a.K >= 2 * BK; }

template <int EPI>
static hipError_t launch_pp(const GemmArgs& a, hipStream_t s) {
    const int cus = num_cus();
    GemmArgs b = a;
    if (((b.order >> 8) & 31) >= 1 && ((b.order >> 8) & 31) < 8) b.order = (b.order & ~(31 << 8)) | (8 << 8);
    const PipePlan pl = make_plan(b, cus, true);
    const int tiles = pl.tiles_m * pl.tiles_n;
    const int grid = tiles < cus ? tiles : cus;
#ifdef VITK_PROBES                                   // timing probes (wrong or no results): diagnostic builds only
    const int probe = (b.order >> 17) & 7;           // bits 17..19 select a timing probe (plain bf16 epilogue only)
    if constexpr (EPI == EPI_BIAS_BF16) {
        if (probe == 1) hipLaunchKernelGGL((gemm_bf16_pp<EPI, 1>), dim3(grid), dim3(kGemmThreads), 0, s, b, pl);
        if (probe == 7) hipLaunchKernelGGL((gemm_bf16_pp<EPI, 7>), dim3(grid), dim3(kGemmThreads), 0, s, b, pl);
        if (probe == 2) hipLaunchKernelGGL((gemm_bf16_pp<EPI, 2>), dim3(grid), dim3(kGemmThreads), 0, s, b, pl);
        if (probe == 3) hipLaunchKernelGGL((gemm_bf16_pp<EPI, 3>), dim3(grid), dim3(kGemmThreads), 0, s, b, pl);
        if (probe == 1 || probe == 2 || probe == 3 || probe == 7) return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((gemm_bf16_pp<EPI>), dim3(grid), dim3(kGemmThreads), 0, s, b, pl);
    return hipGetLastError();
}

hipError_t launch_gemm_pp(const GemmArgs& a, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case EPI_BIAS_BF16: return launch_pp<EPI_BIAS_BF16>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_pp<EPI_BIAS_GELU_BF16>(a, s);
        case EPI_BIAS_GELU_DUAL_BF16: return launch_pp<EPI_BIAS_GELU_DUAL_BF16>(a, s);
        case EPI_BIAS_RESID_F32: return launch_pp<EPI_BIAS_RESID_F32>(a, s);
        case EPI_QKV_BF16: return launch_pp<EPI_QKV_BF16>(a, s);
        case EPI_BIAS_F32: return launch_pp<EPI_BIAS_F32>(a, s);
        case EPI_BIAS_TANH_BF16: return launch_pp<EPI_BIAS_TANH_BF16>(a, s);
        case EPI_MUL_DGELU_BF16: return launch_pp<EPI_MUL_DGELU_BF16>(a, s);
        case EPI_MUL_DTANH_BF16: return launch_pp<EPI_MUL_DTANH_BF16>(a, s);
        case EPI_RESID_LN_F32: return launch_pp<EPI_RESID_LN_F32>(a, s);
        case EPI_LN_GELU_BF16: return launch_pp<EPI_LN_GELU_BF16>(a, s);
        case EPI_LN_QKV_BF16: return launch_pp<EPI_LN_QKV_BF16>(a, s);
        case EPI_RESID_LN_FP: return launch_pp<EPI_RESID_LN_FP>(a, s);
        case EPI_RESID_LN_PP: return launch_pp<EPI_RESID_LN_PP>(a, s);
        case EPI_RESID_PF: return launch_pp<EPI_RESID_PF>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vitk
