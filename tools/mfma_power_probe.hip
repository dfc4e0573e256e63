// Sustained matrix rate of the two bf16 MFMA shapes on random operands (the chip is power-limited under dense MFMA
// work: the shape that needs fewer register-file reads per flop may hold a higher clock).
// 256 workgroups x 8 waves, operands loaded once from a random buffer, `iters` x 32 independent MFMAs per wave.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_power_probe.hip -o tools/mfma_power_probe && tools/mfma_power_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ src, int iters, float* sink) {
    const int tid = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(blockIdx.x * 512 + tid) * 8 + i]);
        b[i] = __builtin_bit_cast(bf16x8, src[(blockIdx.x * 512 + tid) * 8 + 4 + i]);
    }
    float total = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i & 15], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) total += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) total += acc[i][0] + acc[i][15];
    }
    if (total == 12345.f) sink[0] = total;
}

template <int SHAPE>
void run(const char* label, const u32x4* src, float* sink, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(256), dim3(512), 0, 0, src, iters / 10, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(256), dim3(512), 0, 0, src, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)iters * 32 * 16384.0 * 8 * 256;      // either shape: 32 x 16x16x32 = 16 x 32x32x16 per iteration
    const double tf = flops / (ms * 1e-3) / 1e12;
    // 256 CUs x 4 SIMDs x 16,384 flop per 16 clocks: the clock this rate implies if the pipe never idles
    printf("%-28s %8.2f ms  %7.1f TFLOP/s  (matrix pipe saturated at %4.2f GHz)\n", label, ms, tf, tf * 1e12 / (1024.0 * 1024.0) / 1e9);
}

int main() {
    const size_t n = (size_t)256 * 512 * 8;
    std::vector<uint32_t> host(n * 4);
    srand(1);
    for (auto& v : host) {
        // two bf16 values of magnitude 1..2 with random signs and mantissas
        const uint32_t lo = (0x3F80u ^ ((rand() & 1u) << 15)) | (rand() & 0x7Fu), hi = (0x3F80u ^ ((rand() & 1u) << 15)) | (rand() & 0x7Fu);
        v = lo | (hi << 16);
    }
    u32x4* src;
    float* sink;
    (void)hipMalloc(&src, n * 16);
    (void)hipMemcpy(src, host.data(), n * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&sink, 64);
    const int iters = 200000;   // ~0.4 s per run: long enough for the clock to settle
    for (int rep = 0; rep < 2; ++rep) {
        run<16>("v_mfma_f32_16x16x32_bf16", src, sink, iters);
        run<32>("v_mfma_f32_32x32x16_bf16", src, sink, iters);
    }
    (void)hipMemset(src, 0, n * 16);
    run<16>("16x16x32 on zeros", src, sink, iters);
    run<32>("32x32x16 on zeros", src, sink, iters);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
