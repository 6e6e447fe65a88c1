// Device-side scalar helpers shared by the kernels.
#pragma once

#include <hip/hip_runtime.h>

// sin(u)^2 for the snake activation (reference l3ac/layers.py:29-33: torch.sin(alpha * x).pow(2)).
// Quadrant reduction n = rint(u * 2/pi), r = u - n * pi/2 with a two-term Cody-Waite constant (the fma makes the
// first step exact), then the cephes single-precision sine kernel on [-pi/4, pi/4] (|err| < 1 ulp there), and
// sin(u)^2 = s^2 for even n, 1 - s^2 for odd n (no cancellation: s^2 <= 1/2).  Absolute error ~1e-7 for
// |u| <= SIN2_ARG_MAX — the accuracy of sinf()^2 — at about a third of the instructions of the OCML sinf, which carries a
// Payne-Hanek branch the activations of this network never need.
//
// Guard.  The reduction is exact while n = rint(u * 2/pi) is the right integer, i.e. far beyond any activation this
// network produces but not for all floats, so the argument is CLAMPED to +-SIN2_ARG_MAX first (one v_med3_f32; a branch
// to an accurate slow path at every activation site was measured to cost the fused kernels their registers and their
// MFMA/VALU interleave).  What the clamp means: snake(x) = x + sin^2(alpha x) / alpha with sin^2 in [0, 1], so for
// |alpha x| > 1e5 the clamped term is still a value in [0, 1] and the result differs from the exact one by at most
// 1 / alpha on a magnitude above 1e5 / alpha: a relative error <= 1e-5, finite, never an integer overflow.  NaN stays NaN
// (through x itself).  tests/test_gpu_blocks.py::test_sin_squared_range checks the accuracy inside the range against fp64
// and this bound beyond it.
constexpr float SIN2_ARG_MAX = 1.0e5f;

__device__ __forceinline__ float sin_squared(float u) {
    u = __builtin_amdgcn_fmed3f(u, -SIN2_ARG_MAX, SIN2_ARG_MAX);
    const float n = rintf(u * 0.636619772367581343f);
    float r = fmaf(n, -1.57079637050628662109375f, u);
    r = fmaf(n, 4.37113900018624283e-8f, r);
    const float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    const float s = fmaf(p * z, r, r);
    const float s2 = s * s;
    return ((int)n & 1) ? 1.0f - s2 : s2;
}

// snake(h) = h + (alpha + 1e-8)^-1 * sin(alpha * h)^2
__device__ __forceinline__ float snake_act(float h, float alpha, float inv_alpha) {
    return fmaf(inv_alpha, sin_squared(alpha * h), h);
}

// Packed (two elements per lane) forms: gfx950 executes v_pk_fma_f32 / v_pk_mul_f32 on register pairs, which nearly
// halves the VALU instruction count of the activation — what co-limits the narrow-stage kernels next to the MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 sin_squared2(f32x2 u) {
    u.x = __builtin_amdgcn_fmed3f(u.x, -SIN2_ARG_MAX, SIN2_ARG_MAX);
    u.y = __builtin_amdgcn_fmed3f(u.y, -SIN2_ARG_MAX, SIN2_ARG_MAX);
    const f32x2 n = __builtin_elementwise_rint(u * 0.636619772367581343f);
    f32x2 r = __builtin_elementwise_fma(n, (f32x2)(-1.57079637050628662109375f), u);
    r = __builtin_elementwise_fma(n, (f32x2)(4.37113900018624283e-8f), r);
    const f32x2 z = r * r;
    f32x2 p = __builtin_elementwise_fma(z, (f32x2)(-1.9515295891e-4f), (f32x2)(8.3321608736e-3f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-1.6666654611e-1f));
    const f32x2 s = __builtin_elementwise_fma(p * z, r, r);                              // sin(r), |r| <= pi/4
    const f32x2 c = __builtin_elementwise_fma(s * s, (f32x2)(-2.0f), (f32x2)(1.0f));      // cos(2r)
    const i32x2 cb = __builtin_bit_cast(i32x2, c) ^ (__builtin_convertvector(n, i32x2) << 31);  // cos(2u) = (-1)^n cos(2r)
    return __builtin_elementwise_fma(__builtin_bit_cast(f32x2, cb), (f32x2)(-0.5f), (f32x2)(0.5f));  // (1 - cos 2u) / 2
}

__device__ __forceinline__ f32x2 snake_act2(f32x2 h, f32x2 alpha, f32x2 inv_alpha) {
    return __builtin_elementwise_fma(inv_alpha, sin_squared2(alpha * h), h);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
