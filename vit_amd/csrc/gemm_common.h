// Shared pieces of the tiled bf16 GEMM kernels: tile constants, the LDS operand image, and the
// epilogue that turns a wave's MFMA accumulators into whole-line global stores.
#pragma once
#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kGemmThreads = 512;
constexpr int kTileBytes = BM * BK * 2;       // one operand tile, 32 KiB
constexpr int kStageBytes = 2 * kTileBytes;   // A tile + W tile
constexpr int kRowBytes = BK * 2;             // 128 B per tile row
constexpr int kEpiBytesPerWave = 4096;        // wave-private epilogue window

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// LDS image of an operand tile: row r (0..255) occupies bytes [128r, 128r+128); its 16-byte
// chunk c (8 bf16 of K) is stored at slot c ^ ((r >> 1) & 7).  With this XOR the 16 lanes of
// every ds_read_b128 lane group of an MFMA fragment read (rows r..r+15, one chunk column) land
// on 16 distinct 16-byte slots of the 256-byte bank row.  LDS-DMA writes linearly (wave base +
// lane*16), so the XOR is applied to the per-lane SOURCE chunk and again on the read.

// rows advance by `step` inside one image of `per` rows: (b, t) -> (b', t')
__device__ __forceinline__ void advance_row(int& b, int& t, int step, int per) {
    t += step;
    while (t >= per) { t -= per; ++b; }
}

// sum over the 16 lanes of a DPP row (lanes 16 i .. 16 i + 15): four rotate-and-add steps, every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
    return v;
}

// Epilogue of one wave's sub-tile: acc[nt][mt][j] = C[m = mw + 16 mt + fr][n = nw + 16 nt + 4 fq + j]
// for mt < my_mt (fr = lane & 15, fq = lane >> 4).  Accumulators (+bias, +GELU/tanh) go through the
// wave-private LDS window `win` and are read back as 16-byte row pieces, so every global store
// (and residual / pos-embed load) instruction covers whole 128-byte lines instead of 32-byte
// fragments of 16 different rows.  Rows at or past `row_limit` (relative to mw) are not stored;
// MASKED = false promises my_mt == 8 and row_limit >= 128 and drops every per-row branch.  AHEAD (1 or 2)
// is the number of addend batches requested ahead of their use (residual / pos-embed epilogues).
template <int EPI, bool MASKED, int AHEAD = 1>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, const f32x4 (&acc)[4][8], char* win, int mw,
                                              int nw, int my_mt, int row_limit, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr bool AUXMUL = (EPI == EPI_MUL_DGELU_BF16 || EPI == EPI_MUL_DTANH_BF16);   // out = bf16(acc) * f'(aux)
    constexpr bool LNIN = (EPI == EPI_LN_GELU_BF16 || EPI == EPI_LN_QKV_BF16);      // LayerNorm applied to the accumulators
    constexpr bool QKV = (EPI == EPI_QKV_BF16 || EPI == EPI_LN_QKV_BF16);
    constexpr bool PIN = (EPI == EPI_RESID_LN_PP || EPI == EPI_RESID_PF);           // residual stream read as two 16-bit planes
    constexpr bool POUT = (EPI == EPI_RESID_LN_FP || EPI == EPI_RESID_LN_PP);       // ... and written as such
    constexpr bool LNOUT = (EPI == EPI_RESID_LN_F32 || POUT);                       // row statistics + bf16 copy of the new rows
    constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_TANH_BF16 ||
                               QKV || EPI == EPI_BIAS_GELU_DUAL_BF16 || AUXMUL || LNIN);
    constexpr bool DUAL = EPI == EPI_BIAS_GELU_DUAL_BF16;       // two trips through the window: pre-activation, then GELU
    if constexpr (OUT_BF16) {
        // window = 32 rows x 128 B.  8-byte piece (row r, chunk c, half h) lives at
        // 128 r + 16 (c ^ (r & 7)) + 8 (h ^ ((r >> 3) & 1)): conflict-free for the MFMA-layout
        // writes (16 rows, one column) and for the row-wise 16-byte reads.
        f32x4 b4[4];                                 // bias in MFMA layout (it precedes GELU / tanh)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            b4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4[nt] = *reinterpret_cast<const f32x4*>(p.bias + nw + nt * 16 + fq * 4);
        }
        // LNIN: value = rstd[m] acc + (-mean[m] rstd[m]) s[n] + c[n]: the column sums s beside the bias (here c), and
        // the two row factors of this lane's eight rows
        f32x4 s4[LNIN ? 4 : 1];
        f32x2 ab8[LNIN ? 8 : 1];
        if constexpr (LNIN) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) s4[nt] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + nt * 16 + fq * 4);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                if (MASKED && mt >= my_mt) break;
                int row = mw + mt * 16 + fr;
                row = row < p.M ? row : p.M - 1;
                ab8[mt] = *reinterpret_cast<const f32x2*>(p.ln_ab + 2 * (size_t)row);
            }
        }
        const int rc = lane >> 3, cc = lane & 7;     // read-back role: row within 8, 16-byte chunk
        uint32_t col_off;                            // element offset contributed by the column
        int rb0 = 0, rt0 = 0;                        // QKV: (image, token) of row mw
        if constexpr (QKV) {
            // column n of the fused projection = which*D + h*hd + e  (reshape(B,N,3,H,hd), transformer.py:42).
            // The divisions run on wave-uniform values (scalar unit); per lane only when the wave's 64
            // columns straddle a head boundary (head_dim not a multiple of 64).
            const int D = p.heads * p.hd;
            const int which_u = nw / D, rem_u = nw - which_u * D;
            const int h_u = rem_u / p.hd, e_u = rem_u - h_u * p.hd;
            if (e_u + 64 <= p.hd) {
                col_off = (((uint32_t)which_u * p.batch * p.heads + h_u) * p.ntok) * p.hd + e_u + cc * 8;
            } else {
                const int n = nw + cc * 8;
                const int which = n / D, rem = n - which * D;
                const int h = rem / p.hd, e = rem - h * p.hd;
                col_off = (((uint32_t)which * p.batch * p.heads + h) * p.ntok) * p.hd + e;
            }
            rb0 = mw / p.ntok;
            rt0 = mw - rb0 * p.ntok;
        } else {
            col_off = (uint32_t)(nw + cc * 8);
        }
        // AUXMUL: the whole sub-tile's pieces of aux are requested up front (16 loads per lane in flight: one memory
        // round trip per tile instead of one per pass; the K loop's fragment registers are free here)
        // (AHEAD == 2; with AHEAD == 1 a pass requests its own pieces: for kernels that hold fragments across the epilogue)
        constexpr bool AUX_UPFRONT = AUXMUL && AHEAD == 2;
        u32x4 aux_all[AUX_UPFRONT ? 4 : 1][4];
        auto load_aux = [&](u32x4 (&dst)[4], int pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = mw + pass * 32 + i * 8 + rc;
                row = row < p.M ? row : p.M - 1;         // loads stay in bounds; such rows are not stored
                dst[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux) +
                                                         ((uint32_t)row * p.ldo + col_off) * 2u);
            }
        };
        if constexpr (AUX_UPFRONT) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (MASKED && pass * 2 >= my_mt) break;
                load_aux(aux_all[pass], pass);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (MASKED && pass * 2 >= my_mt) break;
#pragma unroll
            for (int rep = 0; rep < (DUAL ? 2 : 1); ++rep) {
            const bool gelu_now = EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_LN_GELU_BF16 || (DUAL && rep == 1);
            char* dst = reinterpret_cast<char*>((DUAL && rep == 0) ? p.out2 : p.out);
            u32x4 (&auxv)[4] = aux_all[AUX_UPFRONT ? pass : 0];
            if constexpr (AUXMUL && !AUX_UPFRONT) load_aux(auxv, pass);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                if (MASKED && pass * 2 + m2 >= my_mt) break;
                const int r = m2 * 16 + fr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 a = acc[nt][pass * 2 + m2];
                    float v0, v1, v2, v3;
                    if constexpr (LNIN) {
                        // on packed f32 arithmetic (v_pk_fma_f32): two instructions per pair of values
                        const f32x2 ra = {ab8[pass * 2 + m2][0], ab8[pass * 2 + m2][0]};
                        const f32x2 rb = {ab8[pass * 2 + m2][1], ab8[pass * 2 + m2][1]};
                        const f32x2 lo = f32x2{a[0], a[1]} * ra + (rb * f32x2{s4[nt][0], s4[nt][1]} + f32x2{b4[nt][0], b4[nt][1]});
                        const f32x2 hi = f32x2{a[2], a[3]} * ra + (rb * f32x2{s4[nt][2], s4[nt][3]} + f32x2{b4[nt][2], b4[nt][3]});
                        v0 = lo[0]; v1 = lo[1]; v2 = hi[0]; v3 = hi[1];
                    } else {
                        v0 = a[0] + b4[nt][0]; v1 = a[1] + b4[nt][1]; v2 = a[2] + b4[nt][2]; v3 = a[3] + b4[nt][3];
                    }
                    if (gelu_now) {
                        f32x2 g01 = {v0, v1}, g23 = {v2, v3};
                        gelu_erf_x4(g01, g23);
                        v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
                    }
                    if constexpr (EPI == EPI_BIAS_TANH_BF16) {
                        v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
                    }
                    const int c = nt * 2 + (fq >> 1), h = fq & 1;
                    u32x2 o = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
                    *reinterpret_cast<u32x2*>(win + r * 128 + ((c ^ (r & 7)) << 4) + ((h ^ ((r >> 3) & 1)) << 3)) = o;
                }
            }
            int rb = rb0, rt = rt0;
            if constexpr (QKV) advance_row(rb, rt, pass * 32 + rc, p.ntok);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 8 + rc;
                u32x4 v = *reinterpret_cast<const u32x4*>(win + r * 128 + ((cc ^ (r & 7)) << 4));
                if (i & 1) v = u32x4{v[2], v[3], v[0], v[1]};     // rows 8..15 and 24..31 hold swapped halves
                if constexpr (AUXMUL) {
                    // the product's bf16 value times the activation's derivative at the saved tensor: what
                    // autograd computes as gelu_backward / tanh_backward on the bf16 matmul output
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 ax = {bf16_lo(auxv[i][j]), bf16_hi(auxv[i][j])};
                        f32x2 d;
                        if constexpr (EPI == EPI_MUL_DGELU_BF16) d = dgelu_erf_x2(ax);
                        else d = 1.0f - ax * ax;
                        v[j] = pack_bf16(bf16_lo(v[j]) * d[0], bf16_hi(v[j]) * d[1]);
                    }
                }
                const int lr = pass * 32 + r;                     // row inside the wave's sub-tile
                uint32_t off;                                     // element offset (< 2^31: checked by the launcher)
                if constexpr (QKV) {
                    off = ((uint32_t)rb * p.heads * p.ntok + rt) * p.hd + col_off;
                    advance_row(rb, rt, 8, p.ntok);
                } else {
                    off = (uint32_t)(mw + lr) * p.ldo + col_off;
                }
#ifdef VITK_PROBES
                if ((p.order >> 22) & 1) continue;                // diagnostic builds: order bit 22 = no stores (timing probe)
#endif
                if (!MASKED || lr < row_limit) *reinterpret_cast<u32x4*>(dst + off * 2u) = v;
            }
            }
        }
    } else {
        // window = 16 rows x 256 B; 16-byte chunk c of row r lives at 256 r + 16 (c ^ (r & 15)).
        // The residual / pos-embed addend is requested AHEAD batches (of 2 m-tiles = 8 loads per lane)
        // ahead of its use, so that one memory round trip covers a quarter of the sub-tile instead of
        // each 16 bytes; a batch's loads are issued before the stores of the batch that follows its
        // predecessor (out may alias resid, but never within rows still to be loaded).
        const int rc = lane >> 4, cc = lane & 15;    // read-back role: row within 4, 16-byte chunk
        const int ncol = nw + cc * 4;
        f32x4 bias_rb = {0.f, 0.f, 0.f, 0.f};       // bias in read-back layout: added after the transpose
        if (p.bias) bias_rb = *reinterpret_cast<const f32x4*>(p.bias + ncol);
        constexpr bool ADDEND = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_PATCH_F32 || LNOUT || PIN);
        constexpr int QB = (EPI == EPI_PATCH_F32) ? 1 : 2;     // m-tiles per addend batch (register budget)
        // 32-bit byte offsets from the (uniform) tensor bases keep the address registers few; the
        // launcher guarantees that every tensor touched here is smaller than 4 GiB.
        const char* resid_b = reinterpret_cast<const char*>(p.resid);
        const char* pos_b = reinterpret_cast<const char*>(p.pos);
        char* out_b = reinterpret_cast<char*>(p.out);
        constexpr int NB = 8 / QB;                   // addend batches
        f32x4 add[AHEAD][QB][4];                     // AHEAD batches of addend loads in flight
        uint32_t orow[AHEAD][QB][4];                     // output row of each piece (PATCH only; else it is mw + local row)
        auto load_batch = [&](int batch, f32x4 (&ad)[QB][4], uint32_t (&orw)[QB][4]) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int mt = batch * QB + q;
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
                        orw[q][i] = (uint32_t)(rb * p.ntok + 1 + rt);
                        ad[q][i] = *reinterpret_cast<const f32x4*>(pos_b + ((uint32_t)(1 + rt) * p.N + ncol) * 4u);
                        advance_row(rb, rt, 4, p.npatch);
                    } else if constexpr (PIN) {
                        // four elements of plane H and of plane L, kept as they are until the batch is used
                        const uint32_t off = ((uint32_t)m * p.ldo + ncol) * 2u;
                        const u32x2 hp = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(p.aux) + off);
                        const u32x2 lp = *reinterpret_cast<const u32x2*>(resid_b + off);
                        ad[q][i] = __builtin_bit_cast(f32x4, u32x4{hp[0], hp[1], lp[0], lp[1]});
                    } else if constexpr (EPI == EPI_BIAS_RESID_F32 || LNOUT) {
                        ad[q][i] = *reinterpret_cast<const f32x4*>(resid_b + ((uint32_t)m * p.ldo + ncol) * 4u);
                    }
                }
            }
        };
        if constexpr (ADDEND) {
            load_batch(0, add[0], orow[0]);
            if constexpr (AHEAD == 2) {
                if (!MASKED || QB < my_mt) load_batch(1, add[1], orow[1]);
            }
        }
#pragma unroll
        for (int batch = 0; batch < NB; ++batch) {
            if (MASKED && batch * QB >= my_mt) break;
            f32x4 (&ad)[QB][4] = add[batch % AHEAD];
            uint32_t (&orw)[QB][4] = orow[batch % AHEAD];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int mt = batch * QB + q;
                if (MASKED && mt >= my_mt) break;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    *reinterpret_cast<f32x4*>(win + fr * 256 + (((nt * 4 + fq) ^ fr) << 4)) = acc[nt][mt];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 4 + rc;
                    f32x4 v = *reinterpret_cast<const f32x4*>(win + r * 256 + ((cc ^ r) << 4));
                    v = v + bias_rb;
                    if constexpr (PIN) {
                        // bits(x) = (H << 16) + sext16(L): H is rounded half up, so L's top bit says whether H was bumped
                        const u32x4 raw = __builtin_bit_cast(u32x4, ad[q][i]);
                        const u32x4 xb = {(raw[0] << 16) + (uint32_t)((int32_t)(raw[2] << 16) >> 16),
                                          (raw[0] & 0xFFFF0000u) + (uint32_t)((int32_t)raw[2] >> 16),
                                          (raw[1] << 16) + (uint32_t)((int32_t)(raw[3] << 16) >> 16),
                                          (raw[1] & 0xFFFF0000u) + (uint32_t)((int32_t)raw[3] >> 16)};
                        v = v + __builtin_bit_cast(f32x4, xb);
                    } else if constexpr (ADDEND) {
                        v = v + ad[q][i];
                    }
                    const int lr = mt * 16 + r;
                    const uint32_t orow_i = (EPI == EPI_PATCH_F32) ? orw[q][i] : (uint32_t)(mw + lr);
                    if constexpr (POUT) {
                        const u32x4 vb = __builtin_bit_cast(u32x4, v);
                        const u32x2 hpk = {((vb[0] + 0x8000u) >> 16) | ((vb[1] + 0x8000u) & 0xFFFF0000u),
                                           ((vb[2] + 0x8000u) >> 16) | ((vb[3] + 0x8000u) & 0xFFFF0000u)};
                        const u32x2 lpk = {(vb[0] & 0xFFFFu) | (vb[1] << 16), (vb[2] & 0xFFFFu) | (vb[3] << 16)};
                        if (!MASKED || lr < row_limit) {
                            const uint32_t off = (orow_i * p.ldo + ncol) * 2u;
                            *reinterpret_cast<u32x2*>(out_b + off) = lpk;
                            *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(p.out2) + off) = hpk;
                        }
                    } else {
                        if (!MASKED || lr < row_limit)
                            *reinterpret_cast<f32x4*>(out_b + (orow_i * p.ldo + ncol) * 4u) = v;
                    }
                    if constexpr (LNOUT) {
                        // the new row's bf16 copy (the next GEMM's operand) and this strip's share of its LayerNorm
                        // statistics: 16 lanes hold the row's 64 columns of this wave
                        const f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};      // packed f32: 5 instructions for both sums
                        const f32x2 s2 = vlo + vhi, q2 = vhi * vhi + vlo * vlo;
                        const float ps = row16_sum(s2[0] + s2[1]);
                        const float pq = row16_sum(q2[0] + q2[1]);
                        if (!MASKED || lr < row_limit) {
                            if constexpr (!POUT) {
                                const u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                                *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(p.out2) + (orow_i * p.ldo + ncol) * 2u) = pk;
                            }
                            if (cc == 0)     // 32-bit offsets: the launcher keeps every tensor below 4 GiB
                                *reinterpret_cast<f32x2*>(reinterpret_cast<char*>(p.ln_stats) +
                                                          (orow_i * (uint32_t)(p.N >> 6) + (uint32_t)(nw >> 6)) * 8u) = f32x2{ps, pq};
                        }
                    }
                }
            }
            // a later batch goes into the registers this one has just released
            if constexpr (ADDEND) {
                if (batch + AHEAD < NB && (!MASKED || (batch + AHEAD) * QB < my_mt)) load_batch(batch + AHEAD, ad, orw);
            }
        }
    }
}

hipError_t launch_gemm_pipe(const GemmArgs& a, int epilogue, bool gather, hipStream_t s);
bool gemm_pp_supports(const GemmArgs& a, bool gather);
hipError_t launch_gemm_pp(const GemmArgs& a, int epilogue, hipStream_t s);

}  // namespace vitk
