// Weight-gradient GEMM of the training step (gfx950 / CDNA4):
//
//   C[m][n] = sum_k G[k][m] * X[k][n]        G bf16 [R, M], X bf16 [R, N] (row-major: k is the slow index)
//
// i.e. dW = dY^T X for a Linear layer (dY = G: rows x out, X: rows x in).  R = batch * tokens is ~50 k while
// M x N is at most 3072 x 768, so the 256 x 256 output tiles alone would occupy a tenth of the chip: the k
// range is split over S workgroups per tile, each writes an f32 partial tile, and a second kernel adds the
// S partials in a fixed order (reproducible; no atomics).
//
// COLSUM: the column sums of G (the bias gradient of the same layer, sum_k G[k][m]) come out of the same pass:
// the workgroups of n-tile 0 multiply every G fragment once more, with an all-ones operand, wave (wm, wn) for
// the m-tiles 2 wn and 2 wn + 1 of its 8 (2 more MFMA per 32), and write a partial vector per split.
//
// Both operands are k-major in memory, the wrong way round for an MFMA fragment (8 consecutive k per
// lane).  A K tile (64 rows of k) of each operand goes to LDS as it lies in memory (LDS-DMA, 512-byte
// rows of 256 m or n), and the fragments come out through ds_read_b64_tr_b16, which hands each lane of a
// 16-lane group the 4 k values of one column out of a 4 x 16 block: two such reads per operand
// fragment of v_mfma_f32_16x16x32_bf16.  32-byte unit u (16 columns = one MFMA tile) of row k is stored
// at unit u ^ f(k), f(k) = (k & 3) | (((k >> 3) & 1) << 2): the 8 rows that 32 lanes of a transposed
// read touch then sit in 8 different bank groups.  LDS-DMA writes linearly, so the XOR is applied to
// each lane's source column.
//
// 8 waves, wave (wm, wn) owns 128 m x 64 n of the tile (8 x 4 MFMA tiles, 128 accumulator registers).
// Two LDS stages; per K tile: wait for this wave's DMA, barrier, issue the next K tile's DMA into the other
// stage, then 16 steps of 4 MFMA whose transposed reads run two steps ahead behind counted lgkmcnt waits.
// The transposed reads are inline asm: issued through the builtin, the compiler would put
// s_waitcnt vmcnt(0) in front of them while the next tile's LDS-DMA is in flight.
#include <type_traits>

#include "gemm_plan.h"

namespace vitk {

// A row of zeros: what the rows past the end of G read as in the last K tile when the row count is not a multiple of 64
__device__ __attribute__((aligned(16))) uint32_t g_tn_zero_row[128 + 4];

constexpr int kTnStage = 64 * 512;            // one operand K tile: 64 k-rows x 256 columns bf16
constexpr int kTnLds = 4 * kTnStage;          // [G stage 0][G stage 1][X stage 0][X stage 1]

template <bool COLSUM>
__global__ __launch_bounds__(512, 2) void gemm_tn_partial(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                          float* __restrict__ part, float* __restrict__ colsum_part,
                                                          int M, int N, int ldg, int ldx, int ktiles_total, int splits,
                                                          int tail_rows) {
    __shared__ __attribute__((aligned(1024))) char smem[kTnLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N >> 8, tiles_m = M >> 8;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): renumber them so that an XCD owns
    // a contiguous run of (split, tile) ids.  The ~32 workgroups it runs at a time then belong to one k range:
    // a K tile of G is fetched once for the tiles_n workgroups that read it and a K tile of X once for tiles_m,
    // instead of each operand coming tiles_n / tiles_m times from beyond L2.
    int idx;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
        idx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    const int tn = idx % tiles_n;
    idx /= tiles_n;
    const int tm = idx % tiles_m;
    const int split = idx / tiles_m;
    const int kt0 = (int)((long)split * ktiles_total / splits), kt1 = (int)((long)(split + 1) * ktiles_total / splits);
    const int ktiles = kt1 - kt0;
    const int m0 = tm << 8, n0 = tn << 8;

    // ---- LDS-DMA: instruction t (0..31) of an operand tile carries k-rows 2t, 2t+1; wave w issues t = 4w..4w+3
    uint32_t g_off[4], x_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 2 * (wave * 4 + i) + (lane >> 5);
        const int p = lane & 31;
        const int f = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((p >> 1) ^ f) << 4) + (p & 1) * 8;
        g_off[i] = ((uint32_t)k * ldg + m0 + col) * 2u;
        x_off[i] = ((uint32_t)k * ldx + n0 + col) * 2u;
    }
    const char* g_base = reinterpret_cast<const char*>(G) + (size_t)kt0 * 64 * ldg * 2;
    const char* x_base = reinterpret_cast<const char*>(X) + (size_t)kt0 * 64 * ldx * 2;
    const size_t g_step = (size_t)64 * ldg * 2, x_step = (size_t)64 * ldx * 2;
    // piece i (0..7) of this wave's share of K tile kt: i < 4 a G piece, else an X piece
    // tail_rows > 0: the last K tile of all holds only that many rows.  Its other rows read zeros for G (so they add
    // nothing to any product or column sum) and re-read the last existing row for X (finite values).
    const int kt_tail = tail_rows > 0 ? ktiles_total - 1 - kt0 : -1;        // this split's index of that K tile, if it has it
    auto issue_piece = [&](int stage, int kt, auto ic) {
        constexpr int i = decltype(ic)::value & 3;
        if (kt == kt_tail) {                                     // wave-uniform, once per launch at most
            const int k = 2 * (wave * 4 + i) + (lane >> 5);
            const int p = lane & 31;
            const int f = (k & 3) | (((k >> 3) & 1) << 2);
            const int col = (((p >> 1) ^ f) << 4) + (p & 1) * 8;
            if constexpr (decltype(ic)::value < 4) {
                char* dg = smem + stage * kTnStage + (wave * 4 + i) * 1024;
                const char* src = k < tail_rows ? g_base + kt * g_step + g_off[i]
                                                : reinterpret_cast<const char*>(g_tn_zero_row) + (col & 127) * 2;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dg, 16, 0, 0);
            } else {
                char* dx = smem + 2 * kTnStage + stage * kTnStage + (wave * 4 + i) * 1024;
                const int kc = k < tail_rows ? k : tail_rows - 1;
                const char* src = x_base + kt * x_step + ((uint32_t)kc * ldx + n0 + col) * 2u;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dx, 16, 0, 0);
            }
            return;
        }
        if constexpr (decltype(ic)::value < 4) {
            char* dg = smem + stage * kTnStage + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void*)(g_base + kt * g_step + g_off[i]), (lds_void*)dg, 16, 0, 0);
        } else {
            char* dx = smem + 2 * kTnStage + stage * kTnStage + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void*)(x_base + kt * x_step + x_off[i]), (lds_void*)dx, 16, 0, 0);
        }
    };
    auto issue = [&](int stage, int kt) {
        issue_piece(stage, kt, std::integral_constant<int, 0>{}); issue_piece(stage, kt, std::integral_constant<int, 4>{});
        issue_piece(stage, kt, std::integral_constant<int, 1>{}); issue_piece(stage, kt, std::integral_constant<int, 5>{});
        issue_piece(stage, kt, std::integral_constant<int, 2>{}); issue_piece(stage, kt, std::integral_constant<int, 6>{});
        issue_piece(stage, kt, std::integral_constant<int, 3>{}); issue_piece(stage, kt, std::integral_constant<int, 7>{});
    };

    // ---- fragment addresses: lane (q = lane / 16: the 8 k of a k-step, tq: row within 4, tp: 8-byte piece)
    const int q = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int fl = tq | ((q & 1) << 2);                        // f(k) of every row this lane addresses
    const int wm = wave >> 2, wn = wave & 3;                   // 2 x 4 waves: 128 m x 64 n each
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    const uint32_t row_b = (uint32_t)((8 * q + tq) * 512 + tp * 8);
    uint32_t ga[8], xa[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] = lds0 + row_b + (uint32_t)(((wm * 8 + j) ^ fl) << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = lds0 + 2 * kTnStage + row_b + (uint32_t)(((wn * 4 + j) ^ fl) << 5);

    f32x4 acc[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // COLSUM: the m-tile pair whose column sums this wave accumulates (none outside n-tile 0)
    const int cs_j0 = (COLSUM && tn == 0) ? 2 * wn : -2;
    f32x4 cs_acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    // A K tile is 16 steps (t = 8 kk + j: k-step kk, m-tile j) of 4 MFMA each.  The fragments of step t + 2
    // (2 transposed reads for the G fragment, and 8 for the X fragments when t + 2 opens a k-step) are
    // requested before the MFMAs of step t, and the wait in front of step t is counted: it leaves the reads
    // issued after G(t) in flight (LDS returns in order), so the LDS latency hides under the matrix work.
    bf16x4 xl[2][4], xh[2][4];         // X fragments of the two k-steps: [k 8q..8q+3 | k 8q+4..8q+7] of the lane's column
    bf16x4 gl[3], gh[3];               // ring of G fragments: step t uses slot t % 3
    auto rd_x = [&](auto stage_c, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        constexpr int off = decltype(stage_c)::value * kTnStage + kk * 32 * 512;
        auto& xl_ = xl; auto& xh_ = xh; auto& xa_ = xa;         // named outside the asm operands: a generic lambda captures only what it names there
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xl_[kk][j]) : "v"(xa_[j]), "n"(off));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xh_[kk][j]) : "v"(xa_[j]), "n"(off + 4 * 512));
        }
    };
    auto rd_g = [&](auto stage_c, auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int off = decltype(stage_c)::value * kTnStage + (t >> 3) * 32 * 512;
        auto& gl_ = gl; auto& gh_ = gh; auto& ga_ = ga;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gl_[t % 3]) : "v"(ga_[t & 7]), "n"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(gh_[t % 3]) : "v"(ga_[t & 7]), "n"(off + 4 * 512));
    };
    auto step = [&](auto stage_c, auto t_c, bool more, int kt) {
        constexpr int t = decltype(t_c)::value, kk = t >> 3, j = t & 7;
        // the next K tile's LDS-DMA, one instruction per step of the first k-step: issued in one burst behind the
        // barrier it would hold every wave's matrix work back for its whole issue time
        if constexpr (t < 8) {
            if (more) issue_piece(decltype(stage_c)::value ^ 1, kt + 1, t_c);
        }
        if constexpr (t + 2 < 16) {
            if constexpr (((t + 2) & 7) == 0) rd_x(stage_c, std::integral_constant<int, (t + 2) / 8>{});
            rd_g(stage_c, std::integral_constant<int, t + 2>{});
        }
        // reads issued after G(t): G(t+1), G(t+2), and the X fragments of the next k-step if it opens at t+1 or t+2
        constexpr int after = (t + 1 < 16 ? 2 + (((t + 1) & 7) == 0 ? 8 : 0) : 0) +
                              (t + 2 < 16 ? 2 + (((t + 2) & 7) == 0 ? 8 : 0) : 0);
        // the wait carries the fragment registers as operands, so the MFMAs below cannot move above it
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(gl[t % 3]), "+v"(gh[t % 3]) : "n"(after));
        if constexpr (j == 0) {
            asm volatile("" : "+v"(xl[kk][0]), "+v"(xl[kk][1]), "+v"(xl[kk][2]), "+v"(xl[kk][3]), "+v"(xh[kk][0]), "+v"(xh[kk][1]),
                              "+v"(xh[kk][2]), "+v"(xh[kk][3]));
        }
        const bf16x8 gf = __builtin_shufflevector(gl[t % 3], gh[t % 3], 0, 1, 2, 3, 4, 5, 6, 7);
        // MFMA row index = n, column index = m: a lane ends up with 4 consecutive n of one row m
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 xf = __builtin_shufflevector(xl[kk][i], xh[kk][i], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf, acc[j][i], 0, 0, 0);
        }
        if constexpr (COLSUM) {
            // every row of the result holds sum_k G[k][m] for the lane's column m
            if ((j & ~1) == cs_j0) cs_acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, gf, cs_acc[j & 1], 0, 0, 0);
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    auto k_tile = [&](auto stage_c, int kt) {
        constexpr int stage = decltype(stage_c)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's pieces of K tile kt have landed
        __builtin_amdgcn_s_barrier();                          // everyone's have; everyone is done with the other stage
        const bool more = kt + 1 < ktiles;
        rd_x(stage_c, C0{});
        rd_g(stage_c, C0{});
        rd_g(stage_c, C1{});
#define VITK_TN_STEP(T) step(stage_c, std::integral_constant<int, T>{}, more, kt);
        VITK_TN_STEP(0) VITK_TN_STEP(1) VITK_TN_STEP(2) VITK_TN_STEP(3) VITK_TN_STEP(4) VITK_TN_STEP(5) VITK_TN_STEP(6) VITK_TN_STEP(7)
        VITK_TN_STEP(8) VITK_TN_STEP(9) VITK_TN_STEP(10) VITK_TN_STEP(11) VITK_TN_STEP(12) VITK_TN_STEP(13) VITK_TN_STEP(14) VITK_TN_STEP(15)
#undef VITK_TN_STEP
    };

    if (ktiles > 0) issue(0, 0);
    for (int kt = 0; kt < ktiles; kt += 2) {
        k_tile(C0{}, kt);
        if (kt + 1 < ktiles) k_tile(C1{}, kt + 1);
    }

    // ---- partial tile: acc[j][i][r] = C[m0 + 128 wm + 16 j + lane % 16][n0 + 64 wn + 16 i + 4 (lane / 16) + r]
    float* out = part + ((size_t)split * M + m0 + wm * 128 + (lane & 15)) * N + n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(out + (size_t)j * 16 * N + i * 16) = acc[j][i];
    if constexpr (COLSUM) {
        if (cs_j0 >= 0 && lane < 16) {
            float* cs = colsum_part + (size_t)split * M + m0 + wm * 128 + cs_j0 * 16 + lane;
            cs[0] = cs_acc[0][0];
            cs[16] = cs_acc[1][0];
        }
    }
}

// C = sum over the S partials, in split order
// (and, with colsum_part, the column-sum vector [M] = sum over the S partial vectors; the last blocks take it)
__global__ __launch_bounds__(256) void gemm_tn_reduce(const float* __restrict__ part, float* __restrict__ C, size_t mn4,
                                                      int splits, const float* __restrict__ colsum_part,
                                                      float* __restrict__ colsum, int m4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn4; i += stride) {
        f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
        for (int s = 1; s < splits; ++s) a = a + reinterpret_cast<const f32x4*>(part)[(size_t)s * mn4 + i];
        reinterpret_cast<f32x4*>(C)[i] = a;
    }
    if (colsum_part) {
        const int i = (int)(gridDim.x - 1 - blockIdx.x) * 256 + (int)threadIdx.x;
        if (i < m4) {
            f32x4 a = reinterpret_cast<const f32x4*>(colsum_part)[i];
            for (int s = 1; s < splits; ++s) a = a + reinterpret_cast<const f32x4*>(colsum_part)[(size_t)s * m4 + i];
            reinterpret_cast<f32x4*>(colsum)[i] = a;
        }
    }
}

// number of k splits: one workgroup per CU where the k range allows (at least 4 K tiles per split)
int gemm_tn_splits(int M, int N, int R) {
    const int tiles = (M >> 8) * (N >> 8), ktiles = (R + 63) / 64;
    if (tiles <= 0 || ktiles <= 0) return 0;
    int s = num_cus() / tiles;
    if (s < 1) s = 1;
    if (s > ktiles / 4) s = ktiles / 4 > 0 ? ktiles / 4 : 1;
    return s;
}

// colsum (optional): f32 [M] = column sums of G; with splits > 1 its partial vectors follow the splits * M * N
// floats of partial tiles in `part`
hipError_t launch_gemm_tn(const void* G, const void* X, float* C, float* colsum, float* part, int M, int N, int R,
                          int ldg, int ldx, hipStream_t s) {
    const int splits = gemm_tn_splits(M, N, R);
    if (splits <= 0) return hipErrorInvalidValue;
    const int grid = (M >> 8) * (N >> 8) * splits;
    float* cs_part = colsum ? (splits == 1 ? colsum : part + (size_t)splits * M * N) : nullptr;
    if (colsum)
        hipLaunchKernelGGL(gemm_tn_partial<true>, dim3(grid), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(G),
                           reinterpret_cast<const bf16_t*>(X), splits == 1 ? C : part, cs_part, M, N, ldg, ldx, (R + 63) / 64, splits, R & 63);
    else
        hipLaunchKernelGGL(gemm_tn_partial<false>, dim3(grid), dim3(512), 0, s, reinterpret_cast<const bf16_t*>(G),
                           reinterpret_cast<const bf16_t*>(X), splits == 1 ? C : part, cs_part, M, N, ldg, ldx, (R + 63) / 64, splits, R & 63);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const size_t mn4 = (size_t)M * N / 4;
    size_t blocks = (mn4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < (size_t)(M / 4 + 255) / 256) blocks = (M / 4 + 255) / 256;
    hipLaunchKernelGGL(gemm_tn_reduce, dim3((unsigned)blocks), dim3(256), 0, s, part, C, mn4, splits, cs_part, colsum, M / 4);
    return hipGetLastError();
}

}  // namespace vitk
