// Issue-rate probe for one CDNA4 SIMD: how many cycles a loop of 32x32x16 bf16 MFMAs, of exponentials, and of both
// interleaved takes with one and with two waves per SIMD.  It answers whether the matrix pipe and the vector ALU
// overlap within a wave and across the waves of a SIMD.
//     hipcc -O2 --offload-arch=gfx950 tools/probe_issue.hip -o tools/probe_issue && tools/probe_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define FM(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t[i]) : "v"(e[i]), "v"(c), "v"(m))
#define EX(i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(t[i]))
#define LD128(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[i & 3]) : "v"(la), "n"((i) * 1024))
#define LDTR(i) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[i & 3]) : "v"(lb), "n"((i) * 512))
#define DEP(i) asm volatile("v_fma_f32 %0, %1, %2, %3\n v_exp_f32 %0, %0" : "=&v"(e[i]) : "v"(t[i]), "v"(c), "v"(m))
#define DPAIR(i) DEP(i); DEP(i + 1)
#define PAIR(i) FM(i); FM(i + 1); EX(i); EX(i + 1)

template <int MODE>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int iters) {
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) a0[i] = a1[i] = a2[i] = a3[i] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (threadIdx.x + i)); y[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float e[16], t[16];
    for (int i = 0; i < 16; ++i) { e[i] = 0.01f * (threadIdx.x + i); t[i] = 0.f; }
    const float c = 0.5f, m = -0.25f;
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.f;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    i32x4 w[4] = {};
    i32x2 u[4] = {};
    const int lane = threadIdx.x & 63;
    // K-fragment pattern of the attention kernel: lane -> row (lane & 31) of 128 B, 16-byte chunk swizzled by the row
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (4 * (lane >> 5) + ((lane & 15) >> 2)) * 128 + (((((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) ^ (((((lane & 15) >> 2) >> 1) & 1) << 2)) << 4) + (lane & 1) * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // ten MFMAs, three accumulators
            MF(a0); MF(a1); MF(a2); MF(a0); MF(a1); MF(a2); MF(a0); MF(a1); MF(a2); MF(a0);
        } else if constexpr (MODE == 1) {   // sixteen (fma, exp)
            PAIR(0); PAIR(2); PAIR(4); PAIR(6); PAIR(8); PAIR(10); PAIR(12); PAIR(14);
        } else if constexpr (MODE == 2) {   // interleaved: one MFMA, two (fma, exp)
            MF(a0); PAIR(0); MF(a1); PAIR(2); MF(a2); PAIR(4); MF(a0); PAIR(6); MF(a1); PAIR(8);
            MF(a2); PAIR(10); MF(a0); PAIR(12); MF(a1); PAIR(14); MF(a2); MF(a0);
        } else if constexpr (MODE == 3) {   // the same work, MFMAs first
            MF(a0); MF(a1); MF(a2); MF(a0); MF(a1); MF(a2); MF(a0); MF(a1); MF(a2); MF(a0);
            PAIR(0); PAIR(2); PAIR(4); PAIR(6); PAIR(8); PAIR(10); PAIR(12); PAIR(14);
        } else if constexpr (MODE == 4) {   // sixteen exponentials alone
            EX(0); EX(1); EX(2); EX(3); EX(4); EX(5); EX(6); EX(7); EX(8); EX(9); EX(10); EX(11); EX(12); EX(13); EX(14); EX(15);
        } else if constexpr (MODE == 5) {   // sixteen fma alone
            FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15);
        } else if constexpr (MODE == 7) {   // sixteen 128-bit LDS reads (1 KiB per wave each)
            LD128(0); LD128(1); LD128(2); LD128(3); LD128(4); LD128(5); LD128(6); LD128(7);
            LD128(8); LD128(9); LD128(10); LD128(11); LD128(12); LD128(13); LD128(14); LD128(15);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
        } else if constexpr (MODE == 8) {   // sixteen transposed 64-bit LDS reads (512 B per wave each)
            LDTR(0); LDTR(1); LDTR(2); LDTR(3); LDTR(4); LDTR(5); LDTR(6); LDTR(7);
            LDTR(8); LDTR(9); LDTR(10); LDTR(11); LDTR(12); LDTR(13); LDTR(14); LDTR(15);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
        } else if constexpr (MODE == 9) {   // one attention block's mix: 10 MFMA, 16 (fma, exp), 4 + 8 LDS reads
            LDTR(0); LDTR(1); LDTR(2); LDTR(3);
            MF(a0); PAIR(0); MF(a1); PAIR(2); MF(a2); PAIR(4); MF(a0); PAIR(6);
            LDTR(4); LDTR(5); LDTR(6); LDTR(7);
            MF(a1); PAIR(8); MF(a2); PAIR(10); MF(a0); PAIR(12);
            LD128(0); LD128(1); LD128(2); LD128(3);
            MF(a1); PAIR(14); MF(a2); MF(a0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
        } else if constexpr (MODE == 10) {  // ten MFMAs on one accumulator
            MF(a0); MF(a0); MF(a0); MF(a0); MF(a0); MF(a0); MF(a0); MF(a0); MF(a0); MF(a0);
        } else if constexpr (MODE == 11) {  // the attention block's chains: 4 dependent, then 3 accumulators twice
            MF(a0); PAIR(0); MF(a0); PAIR(2); MF(a0); PAIR(4); MF(a0); PAIR(6); MF(a1); PAIR(8);
            MF(a2); PAIR(10); MF(a3); PAIR(12); MF(a1); PAIR(14); MF(a2); MF(a3);
        } else if constexpr (MODE == 12) {  // the same with each exponential right behind its fma
            MF(a0); DPAIR(0); MF(a0); DPAIR(2); MF(a0); DPAIR(4); MF(a0); DPAIR(6); MF(a1); DPAIR(8);
            MF(a2); DPAIR(10); MF(a3); DPAIR(12); MF(a1); DPAIR(14); MF(a2); MF(a3);
        } else if constexpr (MODE == 6) {   // interleaved: one MFMA, four fma (no transcendental)
            MF(a0); FM(0); FM(1); FM(2); FM(3); MF(a1); FM(4); FM(5); FM(6); FM(7); MF(a2); FM(8); FM(9); FM(10); FM(11);
            MF(a0); FM(12); FM(13); FM(14); FM(15); MF(a1); FM(0); FM(1); FM(2); FM(3); MF(a2); FM(4); FM(5); FM(6); FM(7);
            MF(a0); FM(8); FM(9); FM(10); FM(11); MF(a1); FM(12); FM(13); FM(14); FM(15); MF(a2); MF(a0);
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i] + e[i] + t[i];
    for (int i = 0; i < 4; ++i) s += (float)(w[i][0] + w[i][3] + u[i][0] + u[i][1]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 20000;
    for (int waves : {4, 8}) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, 100);
        hipEventRecord(a);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        printf("%-44s %d wave(s)/SIMD, %3d CUs: %8.1f counter ticks/iter  %8.2f ns/iter\n", name, waves / 4, blocks, mean / iters,
               ms * 1e6 / iters);
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    for (int blocks : {1, 256}) {
        run<0>("10 MFMA 32x32x16", out, cyc, blocks);
        run<4>("16 v_exp_f32", out, cyc, blocks);
        run<5>("16 v_fma_f32", out, cyc, blocks);
        run<1>("16 (fma, exp)", out, cyc, blocks);
        run<2>("10 MFMA interleaved with 16 (fma, exp)", out, cyc, blocks);
        run<3>("10 MFMA, then 16 (fma, exp)", out, cyc, blocks);
        run<6>("10 MFMA interleaved with 32 fma", out, cyc, blocks);
        run<10>("10 MFMA on one accumulator", out, cyc, blocks);
        run<11>("block chains (4 dependent + 3x2), 16 (fma, exp)", out, cyc, blocks);
        run<12>("block chains, exp right behind its fma", out, cyc, blocks);
        run<7>("16 ds_read_b128 (16 KiB per wave)", out, cyc, blocks);
        run<8>("16 ds_read_b64_tr_b16 (8 KiB per wave)", out, cyc, blocks);
        run<9>("attention block mix (8 KiB LDS per wave)", out, cyc, blocks);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return 0;
}
