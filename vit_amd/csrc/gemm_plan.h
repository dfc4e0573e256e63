// Tile plan shared by the persistent GEMM kernels: tile height, tile order, per-XCD chunks.
#pragma once
#include <atomic>

#include "gemm_common.h"

namespace vitk {

constexpr int kPipeLdsBytes = 2 * kStageBytes + 8 * kEpiBytesPerWave;   // 160 KiB

struct PipePlan {
    int mtot;        // m-tiles (16 rows) per output tile
    int tiles_m;     // ceil(M / (16 mtot))
    int tiles_n;     // N / 256
    int gn;          // n-group width of the tile order (0 = plain n-fastest)
    // Two-height plan (mixed != 0; plain order only): every XCD owns xcd_units * 16 consecutive rows, cut into na
    // panels of ha m-tiles followed by nb panels of hb m-tiles (tiles_n output tiles per panel).  Its workgroups
    // take the XCD's tiles round-robin, so each gets tall tiles first and short ones last and the round count need
    // not be a whole number of equal tiles: 3 tiles of 13 become 13 + 13 + 11 where 2.85 rounds of 13 were needed.
    int mixed, xcd_units, ha, na, hb, nb;
};

// tile `local` of XCD `xcd` in the two-height plan -> first row, height (m-tiles), n-tile
__device__ __forceinline__ void mixed_tile(const PipePlan& pl, int xcd, int local, int& m0, int& mtot, int& tile_n) {
    const int panel = local / pl.tiles_n;
    tile_n = local - panel * pl.tiles_n;
    const bool tall = panel < pl.na;
    mtot = tall ? pl.ha : pl.hb;
    m0 = (xcd * pl.xcd_units + (tall ? panel * pl.ha : pl.na * pl.ha + (panel - pl.na) * pl.hb)) * 16;
}

// position `idx` of the tile order -> (tile_m, tile_n).  Plain: ids walk the n tiles of one
// activation panel.  Grouped: n tiles in groups of gn; ids walk n inside the group, then m, then
// the next group, so a group's weight panels stay in the XCD's L2 while activations stream past.
__device__ __forceinline__ void tile_coords(const PipePlan& pl, int idx, int& tile_m, int& tile_n) {
    if (pl.gn > 0) {
        const int per_group = pl.gn * pl.tiles_m;
        const int full = pl.tiles_n / pl.gn;
        int g = idx / per_group, rem = idx - g * per_group, width = pl.gn;
        if (g >= full) { g = full; rem = idx - full * per_group; width = pl.tiles_n - full * pl.gn; }
        tile_m = rem / width;
        tile_n = g * pl.gn + (rem - tile_m * width);
    } else {
        tile_m = idx / pl.tiles_n;
        tile_n = idx - tile_m * pl.tiles_n;
    }
}

// Cost of one output tile of h m-tiles, in thousandths of an m-tile of the tallest tile: short tiles cost more per
// row (per-tile pipeline fill, epilogue latencies, more weight bytes per flop through the LDS-DMA): measured
// +2.4 % per m-tile below 16 on the ViT shapes (M = 73,856, N = 3072, K = 768: 328 us at 16 m-tiles and 14 rounds,
// 361 us at 11 m-tiles and 20 rounds, although 14 x 16 > 20 x 11).
inline long tile_cost(int h) { return (long)h * (1000 + 24 * (16 - h)); }

// A last round that only part of the workgroups take runs faster than a full one (less contention for HBM and
// L2): thousandths of a full round for a share of `rem` of `cus` workgroups (measured 0.57-0.64 at a tenth of the
// workgroups, 0.84 at 30 %, 0.91 at half, 1.0 from 80 %).
inline long tail_round(long rem, int cus) {
    if (rem == 0) return 0;
    const long t = 550 + 700 * rem / cus;
    return t < 1000 ? t : 1000;
}

// Two-height plan for `units` m-tiles per XCD, tiles_n n-tiles and `slots` workgroups per XCD: the (ha, na, hb) whose
// busiest workgroup has the least tile cost (workgroup s takes local tiles s, s + slots, ...); hmin <= hb <= ha <= 16.
// Returns that maximum (in tile_cost units).
inline int plan_two_heights(int units, int tiles_n, int slots, int hmin, int& ha, int& na, int& hb, int& nb) {
    int best = 1 << 30;
    for (int a = 16; a >= hmin; --a)
        for (int b = a; b >= hmin; --b)
            for (int n1 = 0; n1 * a <= units + a - 1; ++n1) {
                const int rest = units - n1 * a;
                const int n2 = rest > 0 ? (rest + b - 1) / b : 0;
                if (a == b && n1 > 0) continue;                      // the uniform plans: once, as (a, 0, a, n2)
                const int tiles = (n1 + n2) * tiles_n;
                int worst = 0;
                for (int s = 0; s < slots && s < tiles; ++s) {
                    int load = 0;
                    for (int l = s; l < tiles; l += slots) load += (int)tile_cost((l / tiles_n) < n1 ? a : b);
                    worst = load > worst ? load : worst;
                }
                if (worst < best) { best = worst; ha = a; na = n1; hb = b; nb = n2; }
            }
    return best;
}

// tile height (in 16-row units) with the shortest estimated critical path: (full rounds + tail round) x tile cost
inline PipePlan make_plan(const GemmArgs& a, int cus, bool allow_mixed = false) {
    PipePlan pl;
    pl.mixed = 0; pl.xcd_units = 0; pl.ha = pl.na = pl.hb = pl.nb = 0;
    pl.tiles_n = a.N / BN;
    int best = 16;
    long best_cost = -1;
    for (int mtot = 16; mtot >= 10; --mtot) {
        const long tiles = (long)((a.M + 16 * mtot - 1) / (16 * mtot)) * pl.tiles_n;
        const long cost = ((tiles / cus) * 1000 + tail_round(tiles % cus, cus)) * tile_cost(mtot);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mtot; }
    }
    const int forced = (a.order >> 8) & 31;          // debug: bits 8..12 force the tile height
    if (forced >= 1 && forced <= 16) best = forced;
    pl.mtot = best;
    pl.tiles_m = (a.M + 16 * best - 1) / (16 * best);
    // n-group width: keep the group's weight panels (256 x K bf16 each) within ~1.6 MB of the XCD's L2
    pl.gn = 0;
    if (pl.tiles_n > 4) {
        const long panel = 256L * a.K * 2;
        long gn = 1600000 / panel;
        gn = gn < 2 ? 2 : gn;
        pl.gn = gn >= pl.tiles_n ? 0 : (int)gn;
    }
    if ((a.order & 15) == 1) pl.gn = 0;
    if ((a.order & 15) == 3) pl.gn = (a.order >> 4) & 15;
    // Plain tile order (few n-tiles, or small K) and at least one tile per workgroup: the two-height plan, where it leaves the busiest
    // workgroup fewer m-tiles than equal tiles do.  Found by exhaustive search, so remembered per shape.
    if (allow_mixed && pl.gn == 0 && !(forced >= 1 && forced <= 16) && ((a.order >> 13) & 1) == 0 && cus % 8 == 0 &&
        (long)pl.tiles_m * pl.tiles_n >= cus && a.M <= (1 << 17)) {
        struct Memo { int M, tiles_n, cus, worst, ha, na, hb, nb; };
        static thread_local Memo memo[8];
        static thread_local int memo_n = 0;
        const Memo* hit = nullptr;
        for (int i = 0; i < memo_n; ++i)
            if (memo[i].M == a.M && memo[i].tiles_n == pl.tiles_n && memo[i].cus == cus) hit = &memo[i];
        const int units = ((a.M + 15) / 16 + 7) / 8;                 // m-tiles per XCD
        if (!hit) {
            Memo m = {a.M, pl.tiles_n, cus, 0, 0, 0, 0, 0};
            m.worst = plan_two_heights(units, pl.tiles_n, cus / 8, 10, m.ha, m.na, m.hb, m.nb);   // heights as the equal-tile plan's: below 160 rows the LDS-DMA feed binds
            const int at = memo_n < 8 ? memo_n++ : 7;
            memo[at] = m;
            hit = &memo[at];
        }
        if ((long)hit->worst * 1000 < best_cost) {
            pl.mixed = 1; pl.xcd_units = units;
            pl.ha = hit->ha; pl.na = hit->na; pl.hb = hit->hb; pl.nb = hit->nb;
        }
    }
    return pl;
}

// Measurement hook (vitk_set_cu_limit): size the persistent grids and tile plans for this many CUs (0 = all).
extern std::atomic<int> g_cu_limit;

inline int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int lim = g_cu_limit.load(std::memory_order_relaxed);
    return lim > 0 && lim < cus ? lim : cus;
}

}  // namespace vitk
