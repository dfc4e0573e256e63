// Flash-style multi-head self-attention forward for ViT token counts
// (197 / 257 / 577 tokens, head_dim 64, 80 or 96) on gfx950 / CDNA4.
//
// Reference semantics (models/transformer.py:42-46): per (image, head),
//   P = softmax(q k^T * scaling) over keys, no mask;  ctx = P v,
// with heads concatenated head-major along the channel dim of the output.
//
// One workgroup = one (image, head) and NW q-tiles of 32 rows, one wave per
// q-tile.  The head's whole K and V ([ntok, head_dim] bf16 each) are staged once
// in LDS (rows of 128 B at head_dim 64, of 256 B with the tail unused at head_dim 80 and 96);
// every wave then walks the keys in blocks of 32 with a wave-private online softmax:
//   S^T[key][q] = K_blk Q^T    (mfma 32x32x16, key on the MFMA row, q on the lane)
//   running max / sum per q live in that lane's registers (both half-waves hold
//   the same q, they exchange one value per block),
//   O^T[d][q] += V_blk^T P^T   where the exponentiated S^T accumulator is, after
//   packing to bf16, already the B operand of this product (its k order is
//   key = 16s + 8(j>>2) + 4*(lane>>5) + (j&3) for element j of k-step s).
// The [ntok, ntok] score matrix never exists in memory.
#include <type_traits>

#include "gemm_plan.h"
#include "vitk_common.h"
#include "vitk_kernels.h"

namespace vitk {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr float kMaskedScore = -1.0e30f;

// f(integral_constant<int, I>) for I = First .. Last - 1, unrolled at compile time
template <int First, int Last, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (First < Last) {
        f(std::integral_constant<int, First>{});
        static_for<First + 1, Last>(f);
    }
}
// Timing probes (K/V of head 0 for every head, no key loop, no stores: wrong or no results) exist in
// diagnostic builds only; elsewhere the kernel's `probe` argument is ignored and the checks fold away.
#ifdef VITK_PROBES
#define VITK_ATTN_PROBE(p) (p)
// probe bit 3 (diagnostic builds): workgroup 0 writes cycle stamps of its first 12 heads over the start of the
// output tensor, uint32 [head][wave][8]: 0 head start (before the wait for the staged K/V), 1 after the barrier,
// 2 next head's LDS-DMA and query loads issued, 3..6 after each key step, 7 after normalising and storing
#define VITK_ATTN_STAMP(slot)                                                                         \
    if ((probe & 8) && blockIdx.x == 0 && it < 12 && lane == 0)                                        \
        reinterpret_cast<uint32_t*>(O)[(it * NW + wave) * 8 + (slot)] = (uint32_t)clock64();
#else
#define VITK_ATTN_PROBE(p) 0
#define VITK_ATTN_STAMP(slot)
#endif

// VTR = true : V row-major in LDS, transposed fragments via ds_read_b64_tr_b16.
// VTR = false: V stored transposed ([d][key]) in LDS, fragments via ds_read_b64.
// HD = 64, or 80 / 96 (VTR only).  LDS slot of 16-byte chunk c of row r:
//   HD 64: K c ^ ((r >> 1) & 7), V c ^ (((r >> 1) & 1) << 2) in 128-byte rows;
//   HD 80, 96: K c ^ (r & 15),   V c ^ ((r & 3) << 2)        in 256-byte rows (10 or 12 of 16 slots used),
// which makes the ds_read_b128 of 32 rows x one chunk column (K) and the transposed read of 4 rows x 64 B
// (V) conflict-free in both geometries.
//
// STREAM = false: the head's whole K and V are staged once (any of the staging variants).
// STREAM = true (VTR only): K and V pass through LDS in chunks of 64 keys, double-buffered: while the
// waves work on chunk c, the LDS-DMA of chunk c+1 is in flight; one barrier per chunk.  32 KiB (64 KiB at
// head_dim 96) of LDS per workgroup whatever the token count, so several workgroups share a CU and the
// first MFMA waits for 16 KiB instead of the whole head.
//
// PD = true (VTR, not STREAM): persistent workgroups, one per CU, each walking heads blockIdx.x,
// blockIdx.x + gridDim.x, ... with two K/V buffers: the LDS-DMA of the next head (and the loads of its
// queries) run under the current head's key loop, so the memory system never waits for a workgroup to
// retire and the next one to be launched.  Needs two heads' K and V in LDS.
//
// LW = true (PD only): one more wave, the last, does nothing but issue the LDS-DMA of the next head.  Issuing a
// head's ~56 DMA instructions costs the waves that share them 2,000-3,300 cycles of a 15,000-cycle head at 197
// tokens (each instruction waits for the address path, and nothing else of that wave runs meanwhile); the extra
// wave sits on the SIMD that has only one q-tile of the seven, and the compute waves go straight to their keys.
template <int HD, int NW, bool VTR, bool STREAM, bool PD = false, bool LW = false>
__global__ __launch_bounds__((NW + (LW ? 1 : 0)) * 64) void attention_fwd(const bf16_t* __restrict__ Q,
                                                         const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ V,
                                                         bf16_t* __restrict__ O, int ntok, int heads,
                                                         float scale_log2e, int probe, int nbh,
                                                         float* __restrict__ lse, int qrows) {
    static_assert(HD == 64 || ((HD == 96 || HD == 80) && VTR), "head_dim 64, or 80 / 96 with transposed LDS reads");
    static_assert(VTR || !STREAM, "streaming needs the transposed LDS reads");
    static_assert(!PD || (VTR && !STREAM), "the persistent variant stages whole heads by LDS-DMA");
    static_assert(!LW || PD, "the loader wave belongs to the persistent variant");
    constexpr int KS = HD / 16;                  // k-steps of q k^T
    constexpr int DB = (HD + 31) / 32;           // 32-row blocks of O^T (the last one half used at head_dim 80:
                                                 // its upper V columns are copies of chunk 9, never stored)
    constexpr int RB = HD == 64 ? 128 : 256;     // bytes per K/V row in LDS
    constexpr int RPP = 1024 / RB;               // rows per LDS-DMA piece (one wave instruction = 1 KiB)
    constexpr int CPR = RB / 16;                 // 16-byte slots per LDS row
    constexpr int KC = 64;                       // keys per streamed chunk
    constexpr int CHUNK = 2 * KC * RB;           // bytes of one chunk buffer: K rows, then V rows
    constexpr bool OVL = PD || STREAM;           // LDS-DMA in flight during the key loop
    // 32-key blocks per key step: what the registers hold besides O^T (two waves per SIMD: 256 registers)
#ifndef VITK_ATTN_NBMAX
#define VITK_ATTN_NBMAX 2
#endif
    constexpr int NBMAX = (PD && HD == 64 && NW + (LW ? 1 : 0) <= 8) ? VITK_ATTN_NBMAX : 2;
    constexpr bool MSUM = OVL || NW > 8;         // row sums on the matrix pipe (16 more registers: not where 4 waves share a SIMD)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int bh = blockIdx.x;                           // image * heads + head
    const int kblocks = (ntok + 31) >> 5;
    const int kpad = kblocks * 32;
    const int vts = kpad + 4;    // V^T row stride in elements (VTR = false)

    const char* k_lds = smem;                      // STREAM: re-pointed at the current chunk buffer
    const char* v_lds = smem + (STREAM ? KC : kpad) * RB;

    auto k_swz = [](int row) { return HD == 64 ? ((row >> 1) & 7) : (row & 15); };
    auto v_swz = [](int row) { return HD == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); };

    // LDS-DMA of `pieces` K pieces followed by `pieces` V pieces (RPP rows each) starting at key row
    // `row0`, into `dst`.  One wave instruction moves 1 KiB; LDS is written linearly, so the slot XOR is
    // applied to each lane's SOURCE chunk.  Rows past ntok re-read the last row (finite values: such keys
    // are masked out of the softmax, and their P is 0 in P.V); at head_dim 96 the slots whose chunk would
    // be 12..15 re-read chunk 11 and are never read back.
    auto stage = [&](char* dst, int row0, int pieces, int head, int first = -1, int stride = NW) {
        if (VITK_ATTN_PROBE(probe) & 1) head = 0;                            // timing probe: K/V from one head (cache-resident)
        const bf16_t* kg = K + (size_t)head * ntok * HD;
        const bf16_t* vg = V + (size_t)head * ntok * HD;
        for (int pi = first < 0 ? wave : first; pi < 2 * pieces; pi += stride) {     // wave-uniform
            const bool is_v = pi >= pieces;
            const int lrow = (is_v ? pi - pieces : pi) * RPP + lane / CPR;
            const int slot = lane % CPR;
            int c = is_v ? slot ^ v_swz(lrow) : slot ^ k_swz(lrow);     // row0 is a multiple of 64: same swizzle
            c = c < HD / 8 ? c : HD / 8 - 1;
            const int grow = row0 + lrow < ntok ? row0 + lrow : ntok - 1;
            const bf16_t* src = (is_v ? vg : kg) + (size_t)grow * HD + c * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst + pi * 1024), 16, 0, 0);
        }
    };

    if constexpr (STREAM) {
        stage(smem, 0, KC / RPP, bh);
    } else if constexpr (VTR) {
        stage(smem, 0, kpad / RPP, bh, wave, NW + (LW ? 1 : 0));
    } else if constexpr (HD == 64) {
        char* kw = smem;
        char* vw = smem + kpad * RB;
        const bf16_t* kg = K + (size_t)bh * ntok * HD;
        const bf16_t* vg = V + (size_t)bh * ntok * HD;
        for (int cid = tid; cid < kpad * 8; cid += NW * 64) {
            const int row = cid >> 3, c = cid & 7;
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (row < ntok) {
                kv = *reinterpret_cast<const u32x4*>(kg + (size_t)row * HD + c * 8);
                vv = *reinterpret_cast<const u32x4*>(vg + (size_t)row * HD + c * 8);
            }
            *reinterpret_cast<u32x4*>(kw + row * RB + ((c ^ ((row >> 1) & 7)) << 4)) = kv;
            unsigned short* vt = reinterpret_cast<unsigned short*>(vw);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[(c * 8 + 2 * j) * vts + row] = (unsigned short)(vv[j] & 0xFFFFu);
                vt[(c * 8 + 2 * j + 1) * vts + row] = (unsigned short)(vv[j] >> 16);
            }
        }
    }

    // ---- this wave's 32 query rows (B-operand fragments straight from global memory).  With resident K/V a
    // wave works through q-tiles wave, wave + NW, ... of the head; with streamed K/V it has the one q-tile
    // (blockIdx.y * NW + wave).
    const int r = lane & 31, hh = lane >> 5;
    int q0 = (STREAM ? blockIdx.y * NW + wave : wave) * 32;
    int qrow = q0 + r;
    bf16x8 qf[KS];
    auto load_q = [&](bf16x8 (&dst)[KS], int head, int row) {
        const int qld = row < ntok ? row : ntok - 1;     // padded rows re-read the last row; never stored
        const bf16_t* qg = Q + ((size_t)head * ntok + qld) * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(qg + ks * 16);
    };
    load_q(qf, bh, qrow);

    f32x16 o_acc[DB];
    auto clear_o = [&]() {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[d][i] = 0.f;
    };
    clear_o();
    // Online softmax with a lazy running max (log2 units): the accumulators are rescaled only when a
    // block's max exceeds the running one by more than kLazy, so probabilities stay below 2^kLazy
    // (exact in fp32 sums, same relative rounding in bf16) and most blocks skip the 32-register rescale.
    constexpr float kLazy = 8.0f;
    float m_run = kMaskedScore, l_run = 0.f;
    // MSUM: the row sums come from the matrix pipe (the softmax is bound by vector issue, not by MFMA): one more
    // MFMA per 16-key slice multiplies the packed probabilities with an all-ones V^T block, so every element
    // of l_acc holds the lane's row sum (of the bf16-rounded probabilities, the same values P.V uses).
    f32x16 l_acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) l_acc[i] = 0.f;
    const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    const int kswz = k_swz(r);              // key rows of a block start at multiples of 32: swizzle of r alone
    // transposed-read lane roles: 16-lane group -> d sub-block, lane -> (key row, 4-column piece)
    const int tsub = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const bool ragged = (ntok & 31) != 0;
    // Per-lane byte offsets of the fragment reads inside a 32-key block, fixed for the whole kernel (blocks start at
    // multiples of 32 rows, so the row swizzles depend on the lane alone): a key step adds one block base per
    // operand and reaches everything else through immediate offsets.
    int kfrag[KS], vfrag[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfrag[ks] = r * RB + (((ks * 2 + hh) ^ kswz) << 4);
#pragma unroll
    for (int d = 0; d < DB; ++d)
        vfrag[d] = (4 * hh + tq) * RB + (((d * 4 + tsub * 2 + (tp >> 1)) ^ v_swz(tq)) << 4) + (tp & 1) * 8;
    // keys of the last block that exist (wave-uniform): a 16-key slice with none of them is skipped
    const int tail_keys = ntok - (kblocks - 1) * 32;

    // One step of the key loop over NB (1 or 2) blocks of 32 keys: two blocks give two independent MFMA
    // chains and share one max exchange and one rescale decision.  lb = first block's index in the LDS
    // image (k_lds / v_lds), gb = its index in the sequence.
    auto key_step = [&](int lb, int gb, auto nb_c) {
        constexpr int NB = decltype(nb_c)::value;
        f32x16 s[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[u][i] = 0.f;
        }
        // S^T blocks: 32 keys x 32 queries each, K = head_dim in steps of 16
        const char* kblk = k_lds + lb * (32 * RB);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* kp = kblk + kfrag[ks];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + u * (32 * RB));
                s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[u], 0, 0, 0);
            }
        }
        if (ragged && gb + NB == kblocks) {     // wave-uniform: keys past ntok exist only in the last block
            const int key0 = (gb + NB - 1) * 32 + 4 * hh;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (key0 + (i & 3) + 8 * (i >> 2) >= ntok) s[NB - 1][i] = kMaskedScore;
        }
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, s[0][i]), s[0][i + 1]);
        mx = fmaxf(mx, s[0][15]);
#pragma unroll
        for (int u = 1; u < NB; ++u) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[u][i]), s[u][i + 1]);
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
            if constexpr (MSUM) {
#pragma unroll
                for (int i = 0; i < 16; ++i) l_acc[i] *= alpha;
            } else {
                l_run *= alpha;
            }
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) o_acc[d][i] *= alpha;
        }
        // OVL: V^T fragments of the first 16-key slice are requested before the exponentials (the inline asm is a
        // scheduling barrier: issued after them, the first P.V of every step would wait out the LDS latency)
        constexpr int G = NB * 2;
        bf16x4 vv[2][DB][2];
        uint32_t vblk[DB];                  // this step's first V row for each 32-column block of O^T
        if constexpr (OVL) {
            const uint32_t vb = (uint32_t)(uintptr_t)(lds_void*)(v_lds + lb * (32 * RB));
#pragma unroll
            for (int d = 0; d < DB; ++d) vblk[d] = vb + (uint32_t)vfrag[d];
        }
        auto issue = [&](auto gc, bf16x4 (&dst)[DB][2]) {
            constexpr int g = decltype(gc)::value;
            constexpr int off = ((g >> 1) * 32 + 16 * (g & 1)) * RB;      // 16-key slice g of the step
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const uint32_t addr = vblk[d];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[d][0]) : "v"(addr), "n"(off));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[d][1]) : "v"(addr), "n"(off + 8 * RB));
            }
        };
        auto landed = [&](bf16x4 (&w)[DB][2], auto last_c) {
            constexpr bool last = decltype(last_c)::value;       // no younger group in flight
            if constexpr (DB == 2) {
                if constexpr (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
                else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
            } else {
                if constexpr (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]));
                else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]));
            }
        };
        // the last block's second 16-key slice holds no key at all when at most 16 of its keys exist (wave-uniform)
        const bool skip_tail = gb + NB == kblocks && tail_keys <= 16;
        if constexpr (OVL) {
            issue(std::integral_constant<int, 0>{}, vv[0]);
            if (G > 2 || !skip_tail) issue(std::integral_constant<int, 1>{}, vv[1]);
        }
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (OVL && u == NB - 1 && skip_tail) {
                // keys 16..31 of the last block do not exist: their probabilities are never used
#pragma unroll
                for (int i = 0; i < 8; ++i) s[u][i] = __builtin_amdgcn_exp2f(fmaf(s[u][i], scale_log2e, -m_run));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[u][i] = __builtin_amdgcn_exp2f(fmaf(s[u][i], scale_log2e, -m_run));
                    if constexpr (!MSUM) psum += s[u][i];
                }
            }
        }
        if constexpr (!MSUM) l_run += psum;
        // O^T[d][q] += V^T[d][key] P^T[key][q]; the packed P^T accumulators are the B operand
        if constexpr (OVL) {
            // The transposed LDS reads go through inline asm with hand-placed lgkmcnt waits.  Issued through
            // the builtin, the compiler puts an s_waitcnt vmcnt(0) in front of them for as long as any
            // LDS-DMA is in flight (it cannot tell that the DMA fills the other buffer), i.e. the first
            // P.V of a head would wait for the whole staging of the next head.  Groups of 2 DB reads
            // (one 16-key slice) are issued two groups ahead of the MFMAs that use them (the first two before the
            // exponentials); the wait is tied to the group's registers so that the MFMAs cannot move above it.
            auto slice = [&](auto gc, auto last_c, auto gs_c) {
                constexpr int g = decltype(gc)::value;
                constexpr int GS = decltype(gs_c)::value;            // slices this step works through
                constexpr int u = g >> 1, st = g & 1;
                u32x4 pk = {pack_bf16(s[u][8 * st + 0], s[u][8 * st + 1]), pack_bf16(s[u][8 * st + 2], s[u][8 * st + 3]),
                            pack_bf16(s[u][8 * st + 4], s[u][8 * st + 5]), pack_bf16(s[u][8 * st + 6], s[u][8 * st + 7])};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
                landed(vv[g & 1], last_c);
                // copies: the slot is refilled (slice g + 2) while these MFMAs are still being issued
                bf16x8 vf[DB];
#pragma unroll
                for (int d = 0; d < DB; ++d) vf[d] = __builtin_shufflevector(vv[g & 1][d][0], vv[g & 1][d][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int d = 0; d < DB; ++d) o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d], pf, o_acc[d], 0, 0, 0);
                l_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, l_acc, 0, 0, 0);
                if constexpr (g + 2 < GS) issue(std::integral_constant<int, g + 2>{}, vv[g & 1]);
            };
            if (skip_tail) {
                static_for<0, G - 1>([&](auto gc) {
                    slice(gc, std::integral_constant<bool, decltype(gc)::value + 1 == G - 1>{}, std::integral_constant<int, G - 1>{});
                });
            } else {
                static_for<0, G>([&](auto gc) {
                    slice(gc, std::integral_constant<bool, decltype(gc)::value + 1 == G>{}, std::integral_constant<int, G>{});
                });
            }
        } else {
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
                        const int keyb = (lb + u) * 32 + 16 * st + 4 * hh;   // + 8t + (0..3)
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
                    if constexpr (MSUM) l_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, l_acc, 0, 0, 0);
                }
            }
        }
    };
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>;
    const int kend = (VITK_ATTN_PROBE(probe) & 2) ? 0 : kblocks;      // probe bit 1: skip the key loop (timing only)

    // normalise and store one q-tile: lane holds q = qrow, d = 32*blk + 8g + 4*hh + (0..3)
    auto finish = [&]() {
        const float l_tot = MSUM ? l_acc[0] : l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        const int b = bh / heads, h = bh - b * heads;
        // training: log2 of the row's softmax denominator in the kernel's units (scores * scale * log2 e),
        // which is all the backward needs to rebuild the probabilities
        if (lse && hh == 0 && qrow < ntok) lse[(size_t)bh * ntok + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
        if constexpr (PD && HD == 64) {
            // Through a wave-private 32 x 128 B window, so that every store instruction writes 8 whole rows
            // of the head's 128-byte output slice instead of 8 bytes of 64 different rows.  8-byte piece
            // (row q, chunk c, half hh) lives at 128 q + 16 (c ^ (q & 7)) + 8 (hh ^ ((q >> 3) & 1)):
            // conflict-free for these writes (16 rows, one column) and for the row-wise 16-byte reads.
            char* win = smem + 4 * kpad * RB + wave * 4096;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk = {pack_bf16(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv),
                                pack_bf16(o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(win + r * 128 + (((d * 4 + g) ^ (r & 7)) << 4) + ((hh ^ ((r >> 3) & 1)) << 3)) = pk;
                }
            const int rc = lane >> 3, cc = lane & 7;
            if (!(VITK_ATTN_PROBE(probe) & 4)) {                                  // probe bit 2: no output stores (timing only)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + rc;
                    u32x4 v = *reinterpret_cast<const u32x4*>(win + row * 128 + ((cc ^ (row & 7)) << 4));
                    if (i & 1) v = u32x4{v[2], v[3], v[0], v[1]};  // rows 8..15 and 24..31 hold swapped halves
                    if (q0 + row < ntok)
                        *reinterpret_cast<u32x4*>(O + (((size_t)b * ntok + q0 + row) * heads + h) * HD + cc * 8) = v;
                }
            }
        } else if (qrow < ntok && !(VITK_ATTN_PROBE(probe) & 4)) {
            bf16_t* og = O + (((size_t)b * ntok + qrow) * heads + h) * HD + 4 * hh;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (d * 32 + 8 * g >= HD) continue;          // head_dim 80: columns 80..95 do not exist
                    u32x2 pk = {pack_bf16(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv),
                                pack_bf16(o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(og + d * 32 + 8 * g) = pk;
                }
        }
    };

    if constexpr (STREAM) {
        const bool active = q0 < qrows;  // wave-uniform: a wave whose whole q-tile is padding (or not asked for) only takes part in staging
#pragma unroll
        for (int i = 0; i < 16; ++i) l_acc[i] = 0.f;
        const int nchunks = (kblocks + 1) >> 1;
        for (int c = 0; c < nchunks; ++c) {
            // chunk c has landed (every wave waits for its own pieces, the barrier publishes them), and
            // every wave is done reading chunk c-1, whose buffer chunk c+1 now overwrites
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
            __syncthreads();
            if (c + 1 < nchunks) stage(smem + ((c + 1) & 1) * CHUNK, (c + 1) * KC, KC / RPP, bh);
            k_lds = smem + (c & 1) * CHUNK;
            v_lds = k_lds + KC * RB;
            if (active && 2 * c < kend) {
                if (2 * c + 2 <= kend) key_step(0, 2 * c, C2{});
                else key_step(0, 2 * c, C1{});
            }
        }
        if (active) finish();
    } else if constexpr (PD) {
        const int buf_bytes = 2 * kpad * RB;
        for (int it = 0; bh < nbh; bh += gridDim.x, ++it) {
            // head bh has landed in buffer it & 1 (every wave waits for its own pieces, the barrier publishes
            // them), and every wave is done with head bh - gridDim.x, whose buffer the next head overwrites
            VITK_ATTN_STAMP(0)
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), through the builtin: the compiler's wait bookkeeping sees it
            __syncthreads();
            VITK_ATTN_STAMP(1)
            const int nxt = bh + gridDim.x;
            bf16x8 qn[KS];
            if constexpr (LW) {
                if (wave == NW) {                          // the loader wave: the next head's K and V, nothing else
                    if (nxt < nbh) stage(smem + ((it + 1) & 1) * buf_bytes, 0, kpad / RPP, nxt, 0, 1);
                    continue;
                }
                if (nxt < nbh) load_q(qn, nxt, wave * 32 + r);
            } else if (nxt < nbh) {                        // workgroup-uniform
                stage(smem + ((it + 1) & 1) * buf_bytes, 0, kpad / RPP, nxt);
                load_q(qn, nxt, wave * 32 + r);
            }
            VITK_ATTN_STAMP(2)
            k_lds = smem + (it & 1) * buf_bytes;
            v_lds = k_lds + kpad * RB;
            q0 = wave * 32;
            qrow = q0 + r;
            while (q0 < qrows) {                           // wave-uniform
                clear_o();
                m_run = kMaskedScore;
#pragma unroll
                for (int i = 0; i < 16; ++i) l_acc[i] = 0.f;
                // Steps of up to four 32-key blocks: a step's time is set by its dependent chain (scores -> max -> exp ->
                // P.V: a one-block step takes as long as a two-block one), so fewer, larger steps; four blocks of scores
                // are what the registers of two waves per SIMD hold.
                int kb = 0;
                if constexpr (NBMAX == 4) {
                    for (; kb + 4 <= kend; kb += 4) {
                        key_step(kb, kb, C4{});
                        VITK_ATTN_STAMP(3 + (kb >> 2 < 2 ? kb >> 2 : 2))
                    }
                } else if constexpr (NBMAX == 3) {
                    // (a remainder of one block joins the last full step's neighbour: 7 blocks run as 3 + 2 + 2)
                    for (; kb + 3 <= kend && kend - kb != 4; kb += 3) {
                        key_step(kb, kb, C3{});
                        VITK_ATTN_STAMP(3 + (kb / 3 < 2 ? kb / 3 : 2))
                    }
                }
                for (; kb + 2 <= kend; kb += 2) {
                    key_step(kb, kb, C2{});
                    VITK_ATTN_STAMP(3 + (kb >> 1 < 2 ? kb >> 1 : 2))
                }
                if (kb < kend) key_step(kb, kb, C1{});
                VITK_ATTN_STAMP(6)
                finish();
                VITK_ATTN_STAMP(7)
                q0 += NW * 32;
                qrow = q0 + r;
                if (q0 < qrows) {
                    // (more q-tiles than waves.)  Waiting here, not at the first use, keeps the key loop free
                    // of vector-memory waits, which would also wait for the next head's LDS-DMA.
                    load_q(qf, bh, qrow);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            }
            if (nxt < nbh) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
            }
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
        __syncthreads();
        while (q0 < qrows) {                               // wave-uniform
            int kb = 0;
            for (; kb + 2 <= kend; kb += 2) key_step(kb, kb, C2{});
            if (kb < kend) key_step(kb, kb, C1{});
            finish();
            q0 += NW * 32;
            qrow = q0 + r;
            if (q0 < qrows) {
                load_q(qf, bh, qrow);
                clear_o();
                m_run = kMaskedScore;
                l_run = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) l_acc[i] = 0.f;
            }
        }
    }
}

template <int HD, int NW, bool VTR, bool STREAM, bool PD = false, bool LW = false>
static hipError_t attn_launch(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                              int ntok, float scale, hipStream_t s, float* lse, int probe, int qrows) {
    const int kpad = ((ntok + 31) / 32) * 32;
    const int rb = HD == 64 ? 128 : 256;
    const size_t lds = STREAM ? (size_t)4 * 64 * rb
                              : VTR ? (size_t)kpad * rb * (PD ? 4 : 2) + (PD && HD == 64 ? NW * 4096 : 0)
                                    : (size_t)kpad * 128 + (size_t)64 * (kpad + 4) * 2;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit the kernel must be told
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd<HD, NW, VTR, STREAM, PD, LW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int qtiles = (ntok + 31) / 32;
    const int nbh = batch * heads;
    dim3 grid(PD ? (nbh < num_cus() ? nbh : num_cus()) : nbh, STREAM ? (qtiles + NW - 1) / NW : 1), block((NW + (LW ? 1 : 0)) * 64);
    hipLaunchKernelGGL((attention_fwd<HD, NW, VTR, STREAM, PD, LW>), grid, block, lds, s,
                       reinterpret_cast<const bf16_t*>(q), reinterpret_cast<const bf16_t*>(k),
                       reinterpret_cast<const bf16_t*>(v), reinterpret_cast<bf16_t*>(out), ntok, heads,
                       scale * 1.44269504088896340736f, probe, nbh, lse, qrows);
    return hipGetLastError();
}

// token counts the kernels take: any with the streamed K/V; the resident variants need K and V of one
// head in the 160 KiB of LDS (608 tokens at head_dim 64, 288 at head_dim 96)
int attention_max_tokens(int hd) { return (hd == 64 || hd == 80 || hd == 96) ? 16384 : 0; }

// Waves of a workgroup that owns one head and deals its q-tiles (or, in the backward, its owner-row tiles) to its
// waves round-robin.  Wave w sits on SIMD w % 4 and takes tiles w, w + c, ...: the launch takes as long as the SIMD
// with the most tiles (the key loop is bound by vector issue).  Fewest tiles on the busiest SIMD, then fewest rounds
// (the sequential chain of one wave), then most waves at work (a third wave on a SIMD hides more of the score ->
// max -> exp -> P.V chain), then fewest idle waves.  Measured at N = 577 (19 q-tiles; 5, 5, 6 tiles on the busiest
// SIMD): 12 waves 217 us, 8 waves 225, 10 waves 242; N = 485: 12 waves 166, 8 waves 178; N = 401: 12 or 10 waves
// 132, 7 waves 142.
int attention_pick_waves(int tiles, int max_waves) {
    long best = -1;
    int nw = 4;
    for (int c : {4, 5, 6, 7, 8, 10, 12}) {
        if (c > max_waves) continue;
        int simd[4] = {0, 0, 0, 0};
        for (int w = 0; w < c && w < tiles; ++w) simd[w & 3] += (tiles - w + c - 1) / c;
        int crit = simd[0];
        for (int i = 1; i < 4; ++i) crit = simd[i] > crit ? simd[i] : crit;
        const int rounds = (tiles + c - 1) / c, active = c < tiles ? c : tiles;
        const long cost = (((long)crit * 64 + rounds) * 16 + (15 - active)) * 16 + (c - active);
        if (best < 0 || cost < best) { best = cost; nw = c; }
    }
    return nw;
}

hipError_t launch_attention(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                            int ntok, int hd, float scale, int variant, hipStream_t s, float* lse, int qrows) {
    if ((hd != 64 && hd != 80 && hd != 96) || ntok < 1) return hipErrorInvalidValue;
    // qrows: only the query rows below it are computed (whole 32-row tiles); the tiling, and so every computed row's
    // arithmetic, is that of the full launch
    if (qrows <= 0 || qrows > ntok) qrows = ntok;
    const int qtiles = (ntok + 31) / 32;
    // variant: 0 = automatic (persistent workgroups with two resident K/V buffers when two heads fit in LDS,
    // one head per workgroup when one fits, else K/V streamed in 64-key chunks); 1 = resident, V stored
    // transposed (head_dim 64); 2 = streamed; 3 = resident, one head per workgroup
    int mode = variant & 3;
    const bool fits = ntok <= (hd == 64 ? 608 : 288);      // 128-byte rows at head_dim 64, 256-byte rows else
    const bool fits2 = ntok <= (hd == 64 ? 256 : 160);     // two heads (and the output windows at head_dim 64)
    if ((mode == 1 && (hd != 64 || !fits)) || (mode == 3 && !fits)) return hipErrorInvalidValue;
    if (mode == 0 && !fits) mode = 2;
    if (mode == 0 && !fits2) mode = 3;
    const int probe = ((variant >> 2) & 7) | (((variant >> 10) & 1) << 3);       // read by diagnostic builds only
    // Waves per workgroup (one 32-row q-tile at a time each).  Resident K/V: one workgroup per head, its
    // waves take q-tiles round-robin (the width is chosen below).  Streamed: one q-tile per wave, fewest
    // workgroups per head, then fewest idle slots.
    int nw = 4;
    if (mode != 2) {
        // two resident heads and the output windows: at most 8 waves; 12 are built for head_dim 64 only
        nw = attention_pick_waves(qtiles, mode == 0 ? 8 : (mode == 3 && hd == 64) ? 12 : 10);
    } else if (qtiles > 4 && qtiles <= 7) {
        nw = 7;
    } else if (qtiles > 7) {
        int best_wgs = 1 << 30, best_idle = 1 << 30;
        for (int c = 4; c <= 8; ++c) {
            const int wgs = (qtiles + c - 1) / c, idle = wgs * c - qtiles;
            if (wgs < best_wgs || (wgs == best_wgs && idle < best_idle)) { best_wgs = wgs; best_idle = idle; nw = c; }
        }
    }
    {   // test hook: variant bits 5..8 force the wave count (resident modes: 4..8 or 10; streamed: 4..8)
        const int forced = (variant >> 5) & 15;
        if ((forced >= 4 && forced <= 8) || ((forced == 10 || forced == 12) && mode != 2)) nw = forced;
    }
    const bool no_loader = (variant >> 11) & 1;     // test hook: the persistent variant without its loader wave
#define VITK_ATTN_CASE(NWV)                                                                                   \
    case NWV:                                                                                                 \
        if (hd == 96 && mode == 2) return attn_launch<96, NWV, true, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (hd == 96 && mode == 0) return attn_launch<96, NWV, true, false, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (hd == 96) return attn_launch<96, NWV, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);   \
        if (hd == 80 && mode == 2) return attn_launch<80, NWV, true, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (hd == 80 && mode == 0) return attn_launch<80, NWV, true, false, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (hd == 80) return attn_launch<80, NWV, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);   \
        if (mode == 2) return attn_launch<64, NWV, true, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);   \
        if (mode == 0 && NWV < 8 && !no_loader) return attn_launch<64, NWV, true, false, true, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (mode == 0) return attn_launch<64, NWV, true, false, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        if (mode == 3) return attn_launch<64, NWV, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);  \
        return attn_launch<64, NWV, false, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
    switch (nw) {
        VITK_ATTN_CASE(4)
        VITK_ATTN_CASE(5)
        VITK_ATTN_CASE(6)
        VITK_ATTN_CASE(7)
        VITK_ATTN_CASE(8)
        case 10:      // resident modes only
            // (head_dim 80 / 96 with two resident heads has at most 5 q-tiles: never 10 waves)
            if (hd == 96) return attn_launch<96, 10, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
            if (hd == 80) return attn_launch<80, 10, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
            if (mode == 0) return attn_launch<64, 10, true, false, true>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
            if (mode == 3) return attn_launch<64, 10, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
            return attn_launch<64, 10, false, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
        case 12:      // one resident head per workgroup, head_dim 64 only
            if (hd == 64 && mode == 3) return attn_launch<64, 12, true, false>(q, k, v, out, batch, heads, ntok, scale, s, lse, probe, qrows);
            return hipErrorInvalidValue;
    }
#undef VITK_ATTN_CASE
    return hipErrorInvalidValue;
}

}  // namespace vitk
