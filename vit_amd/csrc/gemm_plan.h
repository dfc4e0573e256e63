// Tile plan shared by the persistent GEMM kernels: tile height, tile order, per-XCD chunks.
#pragma once
#include "gemm_common.h"

namespace vitk {

constexpr int kPipeLdsBytes = 2 * kStageBytes + 8 * kEpiBytesPerWave;   // 160 KiB

struct PipePlan {
    int mtot;        // m-tiles (16 rows) per output tile
    int tiles_m;     // ceil(M / (16 mtot))
    int tiles_n;     // N / 256
    int gn;          // n-group width of the tile order (0 = plain n-fastest)
};

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

// tile height (in 16-row units) that minimises rounds x height, rounds = ceil(tiles / cus)
inline PipePlan make_plan(const GemmArgs& a, int cus) {
    PipePlan pl;
    pl.tiles_n = a.N / BN;
    int best = 16;
    long best_cost = -1;
    for (int mtot = 16; mtot >= 10; --mtot) {
        const long tiles_m = (a.M + 16 * mtot - 1) / (16 * mtot);
        const long rounds = (tiles_m * pl.tiles_n + cus - 1) / cus;
        const long cost = rounds * mtot;
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
    return pl;
}

inline int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace vitk
