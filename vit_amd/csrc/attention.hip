// Flash-style multi-head self-attention forward for ViT token counts
// (197 / 577 tokens, head_dim 64 or 96) on gfx950 / CDNA4.
//
// Reference semantics (models/transformer.py:42-46): per (image, head),
//   P = softmax(q k^T * scaling) over keys, no mask;  ctx = P v,
// with heads concatenated head-major along the channel dim of the output.
//
// One workgroup = one (image, head) and NW q-tiles of 32 rows, one wave per
// q-tile.  The head's whole K and V ([ntok, head_dim] bf16 each) are staged once
// in LDS (rows of 128 B at head_dim 64, of 256 B with 64 B unused at head_dim 96);
// every wave then walks the keys in blocks of 32 with a wave-private online softmax:
//   S^T[key][q] = K_blk Q^T    (mfma 32x32x16, key on the MFMA row, q on the lane)
//   running max / sum per q live in that lane's registers (both half-waves hold
//   the same q, they exchange one value per block),
//   O^T[d][q] += V_blk^T P^T   where the exponentiated S^T accumulator is, after
//   packing to bf16, already the B operand of this product (its k order is
//   key = 16s + 8(j>>2) + 4*(lane>>5) + (j&3) for element j of k-step s).
// The [ntok, ntok] score matrix never exists in memory.
#include <type_traits>

#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr float kMaskedScore = -1.0e30f;
static int g_attn_probe = 0;   // timing probe (debug variant bit 2): all workgroups read head 0

// VTR = true : V row-major in LDS, transposed fragments via ds_read_b64_tr_b16.
// VTR = false: V stored transposed ([d][key]) in LDS, fragments via ds_read_b64.
// HD = 64 or 96 (VTR only).  LDS slot of 16-byte chunk c of row r:
//   HD 64: K c ^ ((r >> 1) & 7), V c ^ (((r >> 1) & 1) << 2) in 128-byte rows;
//   HD 96: K c ^ (r & 15),       V c ^ ((r & 3) << 2)        in 256-byte rows (12 of 16 slots used),
// which makes the ds_read_b128 of 32 rows x one chunk column (K) and the transposed read of 4 rows x 64 B
// (V) conflict-free in both geometries.
template <int HD, int NW, bool VTR>
__global__ __launch_bounds__(NW * 64) void attention_fwd(const bf16_t* __restrict__ Q,
                                                         const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ V,
                                                         bf16_t* __restrict__ O, int ntok, int heads,
                                                         float scale_log2e, int probe) {
    static_assert(HD == 64 || (HD == 96 && VTR), "head_dim 64, or 96 with transposed LDS reads");
    constexpr int KS = HD / 16;                  // k-steps of q k^T
    constexpr int DB = HD / 32;                  // 32-row blocks of O^T
    constexpr int RB = HD == 64 ? 128 : 256;     // bytes per K/V row in LDS
    constexpr int RPP = 1024 / RB;               // rows per LDS-DMA piece (one wave instruction = 1 KiB)
    constexpr int CPR = RB / 16;                 // 16-byte slots per LDS row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int bh = (probe & 1) ? 0 : blockIdx.x;   // image * heads + head (probe: every workgroup reads head 0)
    const int kblocks = (ntok + 31) >> 5;
    const int kpad = kblocks * 32;
    const int vts = kpad + 4;    // V^T row stride in elements (VTR = false)

    char* k_lds = smem;
    char* v_lds = smem + kpad * RB;

    // ---- stage K and V of this head
    auto k_swz = [](int row) { return HD == 64 ? ((row >> 1) & 7) : (row & 15); };
    auto v_swz = [](int row) { return HD == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); };
    const bf16_t* kg = K + (size_t)bh * ntok * HD;
    const bf16_t* vg = V + (size_t)bh * ntok * HD;
    if constexpr (VTR) {
        // LDS-DMA: one wave instruction moves 1 KiB = RPP LDS rows; LDS is written linearly, so the slot
        // XOR is applied to each lane's SOURCE chunk.  Rows past ntok re-read the last row (finite
        // values: such keys are masked out of the softmax, and their P is 0 in P.V); at head_dim 96 the
        // slots whose chunk would be 12..15 re-read chunk 11 and are never read back.
        const int pieces = kpad / RPP;                       // per operand
        for (int pi = wave; pi < 2 * pieces; pi += NW) {     // wave-uniform
            const bool is_v = pi >= pieces;
            const int row = (is_v ? pi - pieces : pi) * RPP + lane / CPR;
            const int slot = lane % CPR;
            int c = is_v ? slot ^ v_swz(row) : slot ^ k_swz(row);
            c = c < HD / 8 ? c : HD / 8 - 1;
            const int grow = row < ntok ? row : ntok - 1;
            const bf16_t* src = (is_v ? vg : kg) + (size_t)grow * HD + c * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + pi * 1024), 16, 0, 0);
        }
    } else if constexpr (HD == 64) {
        for (int cid = tid; cid < kpad * 8; cid += NW * 64) {
            const int row = cid >> 3, c = cid & 7;
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (row < ntok) {
                kv = *reinterpret_cast<const u32x4*>(kg + (size_t)row * HD + c * 8);
                vv = *reinterpret_cast<const u32x4*>(vg + (size_t)row * HD + c * 8);
            }
            *reinterpret_cast<u32x4*>(k_lds + row * RB + ((c ^ ((row >> 1) & 7)) << 4)) = kv;
            unsigned short* vt = reinterpret_cast<unsigned short*>(v_lds);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[(c * 8 + 2 * j) * vts + row] = (unsigned short)(vv[j] & 0xFFFFu);
                vt[(c * 8 + 2 * j + 1) * vts + row] = (unsigned short)(vv[j] >> 16);
            }
        }
    }

    // ---- this wave's 32 query rows: B-operand fragments straight from global memory.
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = (blockIdx.y * NW + wave) * 32;
    const int qrow = q0 + r;
    const int qld = qrow < ntok ? qrow : ntok - 1;   // padded rows re-read the last row; never stored
    bf16x8 qf[KS];
    {
        const bf16_t* qg = Q + ((size_t)bh * ntok + qld) * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qg + ks * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
    __syncthreads();

    f32x16 o_acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[d][i] = 0.f;
    // Online softmax with a lazy running max (log2 units): the accumulators are rescaled only when a
    // block's max exceeds the running one by more than kLazy, so probabilities stay below 2^kLazy
    // (exact in fp32 sums, same relative rounding in bf16) and most blocks skip the 32-register rescale.
    constexpr float kLazy = 8.0f;
    float m_run = kMaskedScore, l_run = 0.f;

    const int kswz = k_swz(r);              // key rows of a block start at multiples of 32: swizzle of r alone
    // transposed-read lane roles: 16-lane group -> d sub-block, lane -> (key row, 4-column piece)
    const int tsub = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const bool ragged = (ntok & 31) != 0;

    // One step of the key loop over NB (1 or 2) blocks of 32 keys.  Two blocks give two independent
    // MFMA chains and share one max exchange and one rescale decision.
    auto key_step = [&](int kb, auto nb_c) {
        constexpr int NB = decltype(nb_c)::value;
        f32x16 s[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[u][i] = 0.f;
        }
        // S^T blocks: 32 keys x 32 queries each, K = head_dim in steps of 16
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const char* krow = k_lds + ((kb + u) * 32 + r) * RB;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((ks * 2 + hh) ^ kswz) << 4));
                s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[u], 0, 0, 0);
            }
        }
        if (ragged && kb + NB == kblocks) {     // wave-uniform: keys past ntok exist only in the last block
            const int key0 = (kb + NB - 1) * 32 + 4 * hh;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (key0 + (i & 3) + 8 * (i >> 2) >= ntok) s[NB - 1][i] = kMaskedScore;
        }
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, s[0][i]), s[0][i + 1]);
        mx = fmaxf(mx, s[0][15]);
        if constexpr (NB == 2) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[1][i]), s[1][i + 1]);
        }
        // The other half-wave holds the other 16 keys of each block for the same query.  The swap
        // exchanges lanes 32..63 of its first operand with lanes 0..31 of its second: each lane finds
        // its partner's value in one of the two results and its own in the other.
        {
            const unsigned mxu = __builtin_bit_cast(unsigned, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mxu, mxu, false, false);
            mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])) * scale_log2e;
        }                                                              // scale > 0: max commutes with it
        if (__any(mx > m_run + kLazy)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) o_acc[d][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[u][i] = __builtin_amdgcn_exp2f(fmaf(s[u][i], scale_log2e, -m_run));
                psum += s[u][i];
            }
        }
        l_run += psum;
        // O^T[d][q] += V^T[d][key] P^T[key][q]; the packed P^T accumulators are the B operand
#pragma unroll
        for (int u = 0; u < NB; ++u) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 pk = {pack_bf16(s[u][8 * st + 0], s[u][8 * st + 1]), pack_bf16(s[u][8 * st + 2], s[u][8 * st + 3]),
                            pack_bf16(s[u][8 * st + 4], s[u][8 * st + 5]), pack_bf16(s[u][8 * st + 6], s[u][8 * st + 7])};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    bf16x4 v0, v1;
                    const int keyb = (kb + u) * 32 + 16 * st + 4 * hh;   // + 8t + (0..3)
                    if constexpr (VTR) {
                        const int chunk = d * 4 + tsub * 2 + (tp >> 1);
                        const int ka = keyb + tq, kb2 = keyb + 8 + tq;
                        const char* a0 = v_lds + ka * RB + ((chunk ^ v_swz(ka)) << 4) + (tp & 1) * 8;
                        const char* a1 = v_lds + kb2 * RB + ((chunk ^ v_swz(kb2)) << 4) + (tp & 1) * 8;
                        v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
                        v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
                    } else {
                        const bf16_t* vt = reinterpret_cast<const bf16_t*>(v_lds) + (d * 32 + r) * vts + keyb;
                        v0 = *reinterpret_cast<const bf16x4*>(vt);
                        v1 = *reinterpret_cast<const bf16x4*>(vt + 8);
                    }
                    const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[d], 0, 0, 0);
                }
            }
        }
    };

    if (q0 < ntok) {   // wave-uniform: a wave whose whole q-tile is padding only took part in staging
        int kb = 0;
        const int kend = (probe & 2) ? 0 : kblocks;      // probe bit 1: skip the key loop (timing only)
        for (; kb + 2 <= kend; kb += 2) key_step(kb, std::integral_constant<int, 2>{});
        if (kb < kend) key_step(kb, std::integral_constant<int, 1>{});

        // ---- normalise and store: lane holds q = qrow, d = 32*blk + 8g + 4*hh + (0..3)
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (qrow < ntok) {
            const int b = bh / heads, h = bh - b * heads;
            bf16_t* og = O + (((size_t)b * ntok + qrow) * heads + h) * HD + 4 * hh;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk = {pack_bf16(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv),
                                pack_bf16(o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(og + d * 32 + 8 * g) = pk;
                }
        }
    }
}

template <int HD, int NW, bool VTR>
static hipError_t attn_launch(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                              int ntok, float scale, hipStream_t s) {
    const int kpad = ((ntok + 31) / 32) * 32;
    const int rb = HD == 64 ? 128 : 256;
    const size_t lds = VTR ? (size_t)kpad * rb * 2 : (size_t)kpad * 128 + (size_t)64 * (kpad + 4) * 2;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit the kernel must be told
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd<HD, NW, VTR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int qtiles = (ntok + 31) / 32;
    dim3 grid(batch * heads, (qtiles + NW - 1) / NW), block(NW * 64);
    hipLaunchKernelGGL((attention_fwd<HD, NW, VTR>), grid, block, lds, s,
                       reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k),
                       reinterpret_cast<const bf16_t*>(v), reinterpret_cast<bf16_t*>(out), ntok, heads,
                       scale * 1.44269504088896340736f, g_attn_probe);
    return hipGetLastError();
}

// token counts whose K and V fit the 160 KiB of LDS: 608 at head_dim 64, 288 at head_dim 96
int attention_max_tokens(int hd) { return hd == 64 ? 608 : hd == 96 ? 288 : 0; }

hipError_t launch_attention(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                            int ntok, int hd, float scale, int variant, hipStream_t s) {
    if ((hd != 64 && hd != 96) || ntok < 1) return hipErrorInvalidValue;
    const int qtiles = (ntok + 31) / 32;
    // Waves per workgroup (one 32-row q-tile each): a single workgroup per head when the
    // q-tiles fit 7 waves (K/V staged once), else the width that leaves the fewest idle
    // wave slots, larger on ties.
    const bool tr = (variant & 1) == 0 || hd != 64;
    g_attn_probe = (variant >> 2) & 3;
    int nw = 4;
    if (qtiles > 4 && qtiles <= 7) {
        nw = 7;
    } else if (qtiles > 7) {
        int best = 1 << 30;
        for (int c = 4; c <= 8; ++c) {
            const int idle = ((qtiles + c - 1) / c) * c - qtiles;
            if (idle <= best) { best = idle; nw = c; }
        }
    }
#define VITK_ATTN_CASE(NWV)                                                                        \
    case NWV:                                                                                      \
        if (hd == 96) return attn_launch<96, NWV, true>(q, k, v, out, batch, heads, ntok, scale, s); \
        return tr ? attn_launch<64, NWV, true>(q, k, v, out, batch, heads, ntok, scale, s)         \
                  : attn_launch<64, NWV, false>(q, k, v, out, batch, heads, ntok, scale, s);
    switch (nw) {
        VITK_ATTN_CASE(4)
        VITK_ATTN_CASE(5)
        VITK_ATTN_CASE(6)
        VITK_ATTN_CASE(7)
        VITK_ATTN_CASE(8)
    }
#undef VITK_ATTN_CASE
    return hipErrorInvalidValue;
}

}  // namespace vitk
