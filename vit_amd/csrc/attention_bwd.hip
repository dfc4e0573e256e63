// Backward of the multi-head self-attention of the ViT blocks (models/transformer.py:42-46) for the
// training step, head_dim 64 (up to 608 tokens) and 80 / 96 (up to 288 tokens: the two-pass kernels, four waves),
// i.e. token counts whose K and V (or Q and dO) of one head fit in LDS (gfx950 / CDNA4).
//
//   S = scale Q K^T,  P = softmax(S),  O = P V          (forward, attention.hip; it also leaves
//                                                          L = log2 sum_k exp2(S log2 e) per query row)
//   dP = dO V^T,  D = rowsum(dO * O),  dS = P * (dP - D)
//   dQ = scale dS K,   dK = scale dS^T Q,   dV = P^T dO
//
// Two passes over a head, one workgroup per (image, head).  Up to 288 tokens both run in one kernel from one LDS image
// of Q, K, V and dO (attention_bwd_fused: every matrix fetched once, D handed over through LDS); above, as two
// launches of one kernel template that keeps two of the four matrices resident and reads the others' rows from global
// memory (attention_bwd), with D in a caller's buffer:
//   MODE 0 (dQ):     K and V resident in LDS; a wave owns 32 query rows (Q, dO fragments in registers), walks
//                    the keys in blocks of 32:  S^T = K Q^T,  dP^T = V dO^T  (key on the MFMA row, query on
//                    the lane, so L and D are per-lane scalars),  dQ^T += K^T dS^T.  Also computes D for its
//                    rows (from dO and O) and stores it for the second pass.
//   MODE 1 (dK, dV): Q and dO resident; a wave owns 32 key rows (K, V fragments in registers), walks the
//                    queries in blocks of 32:  S = Q K^T,  dP = dO V^T  (query on the MFMA row: L and D come
//                    from LDS per accumulator row),  dK^T += Q^T dS,  dV^T += dO^T P.
// In both modes the exponentiated / differentiated score block, packed to bf16, is directly the B operand of the
// output products (the same trick as the forward's P.V), and the transposed A operands (K^T, Q^T, dO^T) come
// from the row-major LDS image through ds_read_b64_tr_b16.  The probabilities are rebuilt as
// exp2(S log2 e - L): no running maximum, no second softmax pass.
//
// LDS rows are 128 B (head_dim 64) or 256 B (80 / 96: chunks past the head's last re-read it and are never used);
// 16-byte chunk c of row r sits at slot c ^ swz(r), swz(r) = ((r >> 1) & 1) << 2 | ((r >> 2) & 3) for 8 slots and
// (r & 3) << 2 | ((r >> 2) & 3) for 16: conflict-free both for the ds_read_b128 of 16 rows x one chunk column and for
// the transposed read of 4 rows x 2 chunks.
// Gradients are written in the layout [batch, ntok, 3, heads, head_dim]: the [rows, 3 D] matrix the input
// projection's backward GEMMs take.
#include <type_traits>

#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_b;
typedef __attribute__((address_space(3))) void lds_void_b;
typedef __attribute__((address_space(1))) const void gbl_void_b;

// Per head_dim: k-steps of 16 columns in the S / dP products, 32-column blocks of the outputs, LDS row bytes.
template <int HD>
struct BwdGeom {
    static_assert(HD == 64 || HD == 80 || HD == 96, "head_dim 64, 80 or 96");
    static constexpr int KS = (HD + 15) / 16, DB = (HD + 31) / 32, RB = HD == 64 ? 128 : 256;
    static constexpr int RPP = 1024 / RB;                 // rows per 1 KiB LDS-DMA instruction
    static constexpr int CPR = RB / 16;                   // 16-byte slots per row
    __host__ __device__ static constexpr int swz(int row) {
        return HD == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
    }
};

// One owner block (32 queries in MODE 0, 32 keys in MODE 1; the lane's row is orow, its operands t1 / t2) against all
// resident blocks: accumulates and stores dQ (MODE 0) or dK and dV (MODE 1).  r1 / r2: the resident matrices in LDS
// (MODE 0: K, V; MODE 1: Q, dO); MODE 0 takes L and D of the lane's query as scalars, MODE 1 reads those of the resident
// queries from LDS.
template <int MODE, int HD>
__device__ __forceinline__ void attn_bwd_owner_block(const char* r1, const char* r2, const float* l_lds, const float* d_lds,
                                                     const bf16x8 (&t1)[BwdGeom<HD>::KS], const bf16x8 (&t2)[BwdGeom<HD>::KS], float Lc, float Dc,
                                                     int blocks, int ntok, float c, float scale, int orow, int b, int h,
                                                     int heads, bf16_t* __restrict__ dqkv) {
    using G = BwdGeom<HD>;
    constexpr int KS = G::KS, DB = G::DB, RB = G::RB;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int tsub = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const bool ragged = (ntok & 31) != 0;
    auto swz = [](int row) { return G::swz(row); };
    // ---- per-lane LDS offsets inside a resident block of 32 rows (32 RB bytes); the swizzle of a row depends on its bits
    // 0..3 only, so none of them changes from block to block and a block costs one add per address
    uint32_t off_row[KS];                  // the lane's row r, chunk 2 ks + hh: the ds_read_b128 of the S / dP products
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) off_row[ks] = (uint32_t)(r * RB + (((ks * 2 + hh) ^ swz(r)) << 4));
    uint32_t off_tr[2][DB][2];             // transposed reads of the output products: k-step st, column block d, rows ka / ka + 8
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int ka = 16 * st + 4 * hh + tq, kb = ka + 8;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int chunk = d * 4 + tsub * 2 + (tp >> 1);
            off_tr[st][d][0] = (uint32_t)(ka * RB + ((chunk ^ swz(ka)) << 4) + (tp & 1) * 8);
            off_tr[st][d][1] = (uint32_t)(kb * RB + ((chunk ^ swz(kb)) << 4) + (tp & 1) * 8);
        }
    }
    // MODE 0, last block of a ragged token count: bit i set if accumulator element i is a real key
    uint32_t last_valid = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((blocks - 1) * 32 + 8 * (i >> 2) + 4 * hh + (i & 3) < ntok) last_valid |= 1u << i;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 out1[DB], out2[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            out1[d] = zero16;
            out2[d] = zero16;
        }
        const f32x2 c2 = {c, c}, nl2 = {-Lc, -Lc}, nd2 = {-Dc, -Dc};

        // the resident block's fragments for the S / dP products are requested one block ahead: the reads of block
        // rb + 1 go out as soon as the MFMAs of block rb have taken their operands and land under the softmax arithmetic
        bf16x8 a1[KS], a2[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a1[ks] = *reinterpret_cast<const bf16x8*>(r1 + off_row[ks]);
            a2[ks] = *reinterpret_cast<const bf16x8*>(r2 + off_row[ks]);
        }
        for (int rb = 0; rb < blocks; ++rb) {
            const char* b1 = r1 + rb * 32 * RB;
            const char* b2 = r2 + rb * 32 * RB;
            // the first MFMA of each chain takes a literal zero as its C operand: no registers to clear
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], t1[0], zero16, 0, 0, 0);
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[0], t2[0], zero16, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], t1[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks], t2[ks], dp, 0, 0, 0);
            }
            {
                // (past the last block this reads the first again: in bounds, never used)
                const int nb = rb + 1 < blocks ? (rb + 1) * 32 * RB : 0;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a1[ks] = *reinterpret_cast<const bf16x8*>(r1 + nb + off_row[ks]);
                    a2[ks] = *reinterpret_cast<const bf16x8*>(r2 + nb + off_row[ks]);
                }
            }
            // the transposed fragments of the output products of this block, requested before the arithmetic as well
            bf16x4 v0[2][DB], v1[2][DB], w0[2][DB], w1[2][DB];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    v0[st][d] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_b*)(b1 + off_tr[st][d][0]));
                    v1[st][d] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_b*)(b1 + off_tr[st][d][1]));
                    if constexpr (MODE == 1) {
                        w0[st][d] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_b*)(b2 + off_tr[st][d][0]));
                        w1[st][d] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_b*)(b2 + off_tr[st][d][1]));
                    }
                }
            // element i of a lane: resident row rb*32 + 8 (i >> 2) + 4 hh + (i & 3), owner row = the lane's.
            // Two elements per instruction (packed f32): p = exp2(s c - L), dS = p (dP - D); the factor `scale` of
            // dQ and dK is applied once to the finished sums.
            f32x16 pr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x2 nl[2] = {nl2, nl2}, nd[2] = {nd2, nd2};
                if constexpr (MODE == 1) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(l_lds + rb * 32 + 8 * g + 4 * hh);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(d_lds + rb * 32 + 8 * g + 4 * hh);
                    nl[0] = f32x2{-l4[0], -l4[1]}; nl[1] = f32x2{-l4[2], -l4[3]};
                    nd[0] = f32x2{-d4[0], -d4[1]}; nd[1] = f32x2{-d4[2], -d4[3]};
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int i = 4 * g + 2 * h2;
                    const f32x2 e = f32x2{s[i], s[i + 1]} * c2 + nl[h2];
                    const f32x2 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                    const f32x2 ds = p * (f32x2{dp[i], dp[i + 1]} + nd[h2]);
                    pr[i] = p[0]; pr[i + 1] = p[1];
                    s[i] = ds[0]; s[i + 1] = ds[1];            // dS: s is reused
                }
            }
            if (MODE == 0 && ragged && rb == blocks - 1) {      // keys past the last token: probability 0
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (!((last_valid >> i) & 1)) { pr[i] = 0.f; s[i] = 0.f; }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 dk = {pack_bf16(s[8 * st + 0], s[8 * st + 1]), pack_bf16(s[8 * st + 2], s[8 * st + 3]),
                            pack_bf16(s[8 * st + 4], s[8 * st + 5]), pack_bf16(s[8 * st + 6], s[8 * st + 7])};
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, dk);
                u32x4 pk = {pack_bf16(pr[8 * st + 0], pr[8 * st + 1]), pack_bf16(pr[8 * st + 2], pr[8 * st + 3]),
                            pack_bf16(pr[8 * st + 4], pr[8 * st + 5]), pack_bf16(pr[8 * st + 6], pr[8 * st + 7])};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const bf16x8 af = __builtin_shufflevector(v0[st][d], v1[st][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    out1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf, out1[d], 0, 0, 0);
                    if constexpr (MODE == 1) {
                        const bf16x8 bf = __builtin_shufflevector(w0[st][d], w1[st][d], 0, 1, 2, 3, 4, 5, 6, 7);
                        out2[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, pf, out2[d], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) out1[d] = out1[d] * scale;   // dQ = scale dS K, dK = scale dS^T Q

        // ---- store: lane holds owner row orow, columns d = 32 blk + 8 g + 4 hh + (0..3)
        if (orow < ntok) {
            bf16_t* base = dqkv + (((size_t)b * ntok + orow) * 3 * heads + h) * HD + 4 * hh;
            const size_t which_stride = (size_t)heads * HD;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (d * 32 + 8 * g >= HD) continue;            // head_dim 80: columns 80..95 do not exist
                    u32x2 a = {pack_bf16(out1[d][4 * g], out1[d][4 * g + 1]), pack_bf16(out1[d][4 * g + 2], out1[d][4 * g + 3])};
                    *reinterpret_cast<u32x2*>(base + (MODE == 0 ? 0 : 1) * which_stride + d * 32 + 8 * g) = a;
                    if constexpr (MODE == 1) {
                        u32x2 v2 = {pack_bf16(out2[d][4 * g], out2[d][4 * g + 1]), pack_bf16(out2[d][4 * g + 2], out2[d][4 * g + 3])};
                        *reinterpret_cast<u32x2*>(base + 2 * which_stride + d * 32 + 8 * g) = v2;
                    }
                }
        }
}

template <int MODE, int NW, int HD>
__global__ __launch_bounds__(NW * 64) void attention_bwd(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ V, const bf16_t* __restrict__ O,
                                                         const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                                         float* __restrict__ Dbuf, bf16_t* __restrict__ dqkv, int ntok,
                                                         int heads, float scale) {
    using G = BwdGeom<HD>;
    constexpr int KS = G::KS, RB = G::RB, RPP = G::RPP, CPR = G::CPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x;
    const int b = bh / heads, h = bh - b * heads;
    const int blocks = (ntok + 31) >> 5, npad = blocks * 32;
    const float c = scale * 1.44269504088896340736f;
    char* r1 = smem;                         // MODE 0: K, MODE 1: Q
    char* r2 = smem + npad * RB;             // MODE 0: V, MODE 1: dO
    float* l_lds = reinterpret_cast<float*>(smem + 2 * npad * RB);      // MODE 1: L and D of every query row
    float* d_lds = l_lds + npad;

    const size_t head_rows = (size_t)bh * ntok;              // first row of this head in Q / K / V
    const size_t ctx_stride = (size_t)heads * HD;            // row stride of O / dO (heads concatenated)
    const bf16_t* o_head = O + (size_t)b * ntok * ctx_stride + h * HD;
    const bf16_t* do_head = dO + (size_t)b * ntok * ctx_stride + h * HD;

    // ---- stage the two resident matrices: RPP rows per LDS-DMA instruction (1 KiB), rows past ntok re-read the last,
    // slots past the head's last chunk re-read that chunk
    {
        const bf16_t* s1 = (MODE == 0 ? K : Q) + head_rows * HD;
        const bf16_t* s2 = MODE == 0 ? V + head_rows * HD : do_head;
        const size_t st2 = MODE == 0 ? HD : ctx_stride;
        const int pieces = npad / RPP;
        for (int pi = wave; pi < 2 * pieces; pi += NW) {      // wave-uniform
            const bool second = pi >= pieces;
            const int lrow = (second ? pi - pieces : pi) * RPP + lane / CPR;
            int chunk = (lane % CPR) ^ G::swz(lrow);
            chunk = chunk < HD / 8 ? chunk : HD / 8 - 1;
            const int grow = lrow < ntok ? lrow : ntok - 1;
            const bf16_t* src = second ? s2 + (size_t)grow * st2 + chunk * 8 : s1 + (size_t)grow * HD + chunk * 8;
            __builtin_amdgcn_global_load_lds((gbl_void_b*)src, (lds_void_b*)(smem + pi * 1024), 16, 0, 0);
        }
        if constexpr (MODE == 1) {
            for (int i = tid; i < npad; i += NW * 64) {
                l_lds[i] = i < ntok ? LSE[head_rows + i] : 1.0e30f;        // padded queries: probability 0
                d_lds[i] = i < ntok ? Dbuf[head_rows + i] : 0.f;
            }
        }
    }

    const int r = lane & 31, hh = lane >> 5;
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    __syncthreads();

    for (int o0 = wave * 32; o0 < ntok; o0 += NW * 32) {     // the wave's 32 owner rows (queries / keys)
        const int orow = o0 + r;
        const int oc = orow < ntok ? orow : ntok - 1;         // padded rows re-read the last row; never stored
        bf16x8 t1[KS], t2[KS];
        float Lc = 0.f, Dc = 0.f;
        {
            const bf16_t* p1 = (MODE == 0 ? Q : K) + (head_rows + oc) * HD + hh * 8;
            const bf16_t* p2 = MODE == 0 ? do_head + (size_t)oc * ctx_stride + hh * 8 : V + (head_rows + oc) * HD + hh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                t1[ks] = *reinterpret_cast<const bf16x8*>(p1 + ks * 16);
                t2[ks] = *reinterpret_cast<const bf16x8*>(p2 + ks * 16);
            }
            if constexpr (MODE == 0) {
                const bf16_t* po = o_head + (size_t)oc * ctx_stride + hh * 8;
                float acc = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(po + ks * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += (float)t2[ks][j] * (float)ov[j];
                }
                Dc = acc + __shfl_xor(acc, 32, 64);           // the other half-wave holds the other 32 columns
                Lc = LSE[head_rows + oc];
                if (hh == 0 && orow < ntok) Dbuf[head_rows + orow] = Dc;
            }
        }
        attn_bwd_owner_block<MODE, HD>(r1, r2, l_lds, d_lds, t1, t2, Lc, Dc, blocks, ntok, c, scale, orow, b, h, heads, dqkv);
    }
}

// Both passes in one workgroup per (image, head) for token counts whose Q, K, V and dO fit in LDS together (<= 288):
// the four matrices are fetched once (8 matrix transfers per head with O and the three gradients, instead of 12), the
// owner rows' operands come from LDS instead of from global memory, and D goes from the dQ pass to the dK/dV pass through
// LDS.  Phase A: every wave takes 32 queries against K, V (MODE 0 above); phase B: 32 keys against Q, dO (MODE 1).
template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_bwd_fused(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, const bf16_t* __restrict__ O,
                                                               const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                                               bf16_t* __restrict__ dqkv, int ntok, int heads, float scale) {
    constexpr int HD = 64, KS = 4, RB = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x;
    const int b = bh / heads, h = bh - b * heads;
    const int blocks = (ntok + 31) >> 5, npad = blocks * 32;
    const float c = scale * 1.44269504088896340736f;
    char* qs = smem;
    char* ks_ = smem + npad * RB;
    char* vs = smem + 2 * npad * RB;
    char* dos = smem + 3 * npad * RB;
    float* l_lds = reinterpret_cast<float*>(smem + 4 * npad * RB);
    float* d_lds = l_lds + npad;
    auto swz = [](int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); };
    const size_t head_rows = (size_t)bh * ntok;
    const size_t ctx_stride = (size_t)heads * HD;
    const bf16_t* o_head = O + (size_t)b * ntok * ctx_stride + h * HD;
    const bf16_t* do_head = dO + (size_t)b * ntok * ctx_stride + h * HD;

    // L first: in front of an LDS store the compiler waits for every LDS-DMA in flight
    for (int i = tid; i < npad; i += NW * 64) l_lds[i] = i < ntok ? LSE[head_rows + i] : 1.0e30f;   // padded queries: probability 0
    {
        const int pieces = npad / 8;                          // per matrix: 8 rows per LDS-DMA instruction (1 KiB)
        for (int pi = wave; pi < 4 * pieces; pi += NW) {      // wave-uniform
            const int m = pi / pieces;                        // 0: Q, 1: K, 2: V, 3: dO
            const int lrow = (pi - m * pieces) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(lrow);
            const int grow = lrow < ntok ? lrow : ntok - 1;   // rows past ntok re-read the last
            const bf16_t* src = m == 3 ? do_head + (size_t)grow * ctx_stride + chunk * 8
                                       : (m == 0 ? Q : m == 1 ? K : V) + (head_rows + grow) * HD + chunk * 8;
            __builtin_amdgcn_global_load_lds((gbl_void_b*)src, (lds_void_b*)(smem + pi * 1024), 16, 0, 0);
        }
    }
    const int r = lane & 31, hh = lane >> 5;
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    __syncthreads();

    // ---- phase A: dQ, and D for phase B
    for (int o0 = wave * 32; o0 < ntok; o0 += NW * 32) {
        const int orow = o0 + r;
        const int oc = orow < ntok ? orow : ntok - 1;
        const int sw = swz(orow);
        bf16x8 t1[KS], t2[KS];
        float acc = 0.f;
        const bf16_t* po = o_head + (size_t)oc * ctx_stride + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            t1[ks] = *reinterpret_cast<const bf16x8*>(qs + orow * RB + (((ks * 2 + hh) ^ sw) << 4));
            t2[ks] = *reinterpret_cast<const bf16x8*>(dos + orow * RB + (((ks * 2 + hh) ^ sw) << 4));
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(po + ks * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)t2[ks][j] * (float)ov[j];
        }
        const float Dc = acc + __shfl_xor(acc, 32, 64);       // the other half-wave holds the other 32 columns
        const float Lc = l_lds[orow];
        if (hh == 0) d_lds[orow] = orow < ntok ? Dc : 0.f;
        attn_bwd_owner_block<0, 64>(ks_, vs, l_lds, d_lds, t1, t2, Lc, Dc, blocks, ntok, c, scale, orow, b, h, heads, dqkv);
    }
    __syncthreads();                                          // D of every query is in LDS

    // ---- phase B: dK, dV
    for (int o0 = wave * 32; o0 < ntok; o0 += NW * 32) {
        const int orow = o0 + r;
        const int sw = swz(orow);
        bf16x8 t1[KS], t2[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            t1[ks] = *reinterpret_cast<const bf16x8*>(ks_ + orow * RB + (((ks * 2 + hh) ^ sw) << 4));
            t2[ks] = *reinterpret_cast<const bf16x8*>(vs + orow * RB + (((ks * 2 + hh) ^ sw) << 4));
        }
        attn_bwd_owner_block<1, 64>(qs, dos, l_lds, d_lds, t1, t2, 0.f, 0.f, blocks, ntok, c, scale, orow, b, h, heads, dqkv);
    }
}

// Q, K, V, dO and L, D of a head within 160 KiB
static bool attn_bwd_fits_fused(int ntok) {
    const int npad = ((ntok + 31) / 32) * 32;
    return (size_t)4 * npad * 128 + (size_t)2 * npad * 4 <= 160 * 1024;
}

template <int NW>
static hipError_t attn_bwd_fused_launch(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                        const float* lse, void* dqkv, int batch, int heads, int ntok, float scale, hipStream_t s) {
    const int npad = ((ntok + 31) / 32) * 32;
    const size_t lds = (size_t)4 * npad * 128 + (size_t)2 * npad * 4;
    auto kf = &attention_bwd_fused<NW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kf, dim3(batch * heads), dim3(NW * 64), lds, s, reinterpret_cast<const bf16_t*>(q),
                       reinterpret_cast<const bf16_t*>(k), reinterpret_cast<const bf16_t*>(v), reinterpret_cast<const bf16_t*>(o),
                       reinterpret_cast<const bf16_t*>(d_o), lse, reinterpret_cast<bf16_t*>(dqkv), ntok, heads, scale);
    return hipGetLastError();
}

// two resident [ntok, row bytes] bf16 matrices + L, D within 160 KiB: 608 tokens at 128-byte rows (155,648 + 4,864 B),
// 288 at 256-byte rows (147,456 + 2,304 B)
int attention_bwd_max_tokens(int hd) { return hd == 64 ? 608 : (hd == 80 || hd == 96) ? 288 : 0; }

template <int NW, int HD>
static hipError_t attn_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                  const float* lse, float* dbuf, void* dqkv, int batch, int heads, int ntok, float scale,
                                  hipStream_t s) {
    const int npad = ((ntok + 31) / 32) * 32;
    const size_t lds0 = (size_t)2 * npad * BwdGeom<HD>::RB, lds1 = lds0 + (size_t)2 * npad * 4;
    if (lds1 > 160 * 1024) return hipErrorInvalidValue;
    auto k0 = &attention_bwd<0, NW, HD>;
    auto k1 = &attention_bwd<1, NW, HD>;
    if (lds1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(k);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(v);
    const bf16_t* O = reinterpret_cast<const bf16_t*>(o);
    const bf16_t* DO = reinterpret_cast<const bf16_t*>(d_o);
    bf16_t* G = reinterpret_cast<bf16_t*>(dqkv);
    hipLaunchKernelGGL(k0, dim3(batch * heads), dim3(NW * 64), lds0, s, Q, K, V, O, DO, lse, dbuf, G, ntok, heads, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k1, dim3(batch * heads), dim3(NW * 64), lds1, s, Q, K, V, O, DO, lse, dbuf, G, ntok, heads, scale);
    return hipGetLastError();
}

hipError_t launch_attention_backward(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                     const float* lse, float* dbuf, void* dqkv, int batch, int heads, int ntok, int hd,
                                     float scale, int variant, hipStream_t s) {
    if (ntok < 1 || ntok > attention_bwd_max_tokens(hd)) return hipErrorInvalidValue;
    // head_dim 80 / 96: three 32-column output blocks (two outputs in the dK / dV pass) need the 512 registers of one
    // wave per SIMD
    if (hd == 80) return attn_bwd_launch<4, 80>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
    if (hd == 96) return attn_bwd_launch<4, 96>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
    const int tiles = (ntok + 31) / 32;
    const int nw = attention_pick_waves(tiles, 8);
    if (attn_bwd_fits_fused(ntok) && !(variant & 0x200)) {   // attention debug variant bit 9: the two-pass kernels at any size
        switch (nw) {
            case 4: return attn_bwd_fused_launch<4>(q, k, v, o, d_o, lse, dqkv, batch, heads, ntok, scale, s);
            case 5: return attn_bwd_fused_launch<5>(q, k, v, o, d_o, lse, dqkv, batch, heads, ntok, scale, s);
            case 6: return attn_bwd_fused_launch<6>(q, k, v, o, d_o, lse, dqkv, batch, heads, ntok, scale, s);
            case 7: return attn_bwd_fused_launch<7>(q, k, v, o, d_o, lse, dqkv, batch, heads, ntok, scale, s);
            default: return attn_bwd_fused_launch<8>(q, k, v, o, d_o, lse, dqkv, batch, heads, ntok, scale, s);
        }
    }
    switch (nw) {
        case 4: return attn_bwd_launch<4, 64>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
        case 5: return attn_bwd_launch<5, 64>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
        case 6: return attn_bwd_launch<6, 64>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
        case 7: return attn_bwd_launch<7, 64>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
        default: return attn_bwd_launch<8, 64>(q, k, v, o, d_o, lse, dbuf, dqkv, batch, heads, ntok, scale, s);
    }
}

}  // namespace vitk
