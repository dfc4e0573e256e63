// Shared device helpers for the ViT forward kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitk {

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// Round-to-nearest-even f32 -> bf16 pair packed in one dword (lo = a, hi = b).
// A plain conversion lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs.
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exact (erf-based) GELU, as nn.GELU() computes it: 0.5 x (1 + erf(x / sqrt 2)).
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 rounding of the
// result), arranged branch-free:  gelu(x) = max(x, 0) - 0.5 |x| q,  q = 1 - erf(|x| / sqrt 2)
// = poly(t) * exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2).  12 VALU ops + rcp + exp2 per value,
// against ~40 with divergent branches for libm's erff.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    q = q * t * e;
    return fmaf(-0.5f * ax, q, fmaxf(x, 0.0f));
}

// d/dx gelu(x) = Phi(x) + x phi(x), with the same erf approximation: Phi(|x|) = 1 - q / 2 and
// phi(x) = exp(-x^2 / 2) / sqrt(2 pi) share the exponential.  (GELU's backward in the training step.)
__device__ __forceinline__ float dgelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    q = 0.5f * q * t * e;                                  // Phi(-|x|)
    const float cdf = x >= 0.f ? 1.0f - q : q;
    return fmaf(x * e, 0.39894228040143267794f, cdf);
}

// The same function on packed f32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: twice the throughput of scalar
// VALU code, and no transcendental) for the fused GEMM epilogue, where GELU of 128 accumulators per lane is
// what the epilogue's time goes to.
//   gelu(x) = x Phi(x),  Phi(u) - 1/2 = u P(u^2) for u = clamp(x, -4, 4)
// P is a degree-6 minimax polynomial (weighted for the error of x Phi) with u P(u^2) pinned to 1/2 at u = 4,
// so that gelu(x) = x (or 0) exactly beyond the clamp, where the true value differs from that by at most
// 4 Phi(-4) = 1.3e-4.  |error of gelu| <= 1.9e-4 over all x in f32 arithmetic: below half a bf16 ulp of any
// result larger than 0.1 in size.
// Two pairs run in lockstep: dependent v_pk_fma_f32 need a wait state between them, which the second chain fills.
__device__ __forceinline__ void gelu_erf_x4(f32x2& a, f32x2& b) {
    const f32x2 ua = {__builtin_amdgcn_fmed3f(a[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(a[1], -4.0f, 4.0f)};
    const f32x2 ub = {__builtin_amdgcn_fmed3f(b[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(b[1], -4.0f, 4.0f)};
    const f32x2 sa = ua * ua, sb = ub * ub;
    f32x2 pa = sa * 2.258807628e-08f + -1.588820034e-06f;
    f32x2 pb = sb * 2.258807628e-08f + -1.588820034e-06f;
    pa = pa * sa + 4.776374102e-05f;  pb = pb * sb + 4.776374102e-05f;
    pa = pa * sa + -8.121860839e-04f; pb = pb * sb + -8.121860839e-04f;
    pa = pa * sa + 8.763685335e-03f;  pb = pb * sb + 8.763685335e-03f;
    pa = pa * sa + -6.455440912e-02f; pb = pb * sb + -6.455440912e-02f;
    pa = pa * sa + 3.978702896e-01f;  pb = pb * sb + 3.978702896e-01f;
    pa = ua * pa + 0.5f;              pb = ub * pb + 0.5f;
    a = a * pa;
    b = b * pb;
}

// d/dx gelu(x) = Phi(x) + x phi(x) for two values on packed arithmetic, with Phi from the polynomial of
// gelu_erf_x4 (|error| <= 3.3e-4 absolute: below a bf16 ulp of the gradients it scales wherever the derivative
// exceeds 0.1) and phi(x) = exp2(-x^2 log2(e) / 2) / sqrt(2 pi).  The GEMM epilogue of the training step's
// GELU backward evaluates 128 of these per lane.
__device__ __forceinline__ f32x2 dgelu_erf_x2(f32x2 a) {
    const f32x2 u = {__builtin_amdgcn_fmed3f(a[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(a[1], -4.0f, 4.0f)};
    const f32x2 s = u * u;
    f32x2 p = s * 2.258807628e-08f + -1.588820034e-06f;
    p = p * s + 4.776374102e-05f;
    p = p * s + -8.121860839e-04f;
    p = p * s + 8.763685335e-03f;
    p = p * s + -6.455440912e-02f;
    p = p * s + 3.978702896e-01f;
    p = u * p + 0.5f;                                              // Phi
    const f32x2 t = (a * a) * (-0.5f * 1.44269504088896340736f);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    return (a * e) * 0.39894228040143267794f + p;
}

}  // namespace vitk
