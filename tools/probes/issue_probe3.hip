// Stand-alone probe (not part of the library): throughput of v_cndmask forms and of the library's activation / split code per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -Il3ac_amd/csrc tools/probes/issue_probe3.hip -o /tmp/issue_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "kernels/device_math.hpp"
#include "kernels/split_bf16.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ float sin_squared_b(float u) {
    u = __builtin_amdgcn_fmed3f(u, -SIN2_ARG_MAX, SIN2_ARG_MAX);
    const float t = fmaf(u, 0.636619772367581343f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -1.57079637050628662109375f, u);
    r = fmaf(n, 4.37113900018624283e-8f, r);
    const float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    const float s = fmaf(p * z, r, r);
    const float s2 = s * s;
    const float sign = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, t) << 31) | 0x3f800000u);
    return fmaf(sign, s2, fmaf(sign, -0.5f, 0.5f));
}
// the packed form's arithmetic (device_math.hpp sin_squared2 / snake_act2), one element at a time: the same operations in the same
// order, so the same bits
__device__ __forceinline__ float snake_cosform(float h, float alpha, float inv_alpha) {
    float u = alpha * h;
    u = __builtin_amdgcn_fmed3f(u, -SIN2_ARG_MAX, SIN2_ARG_MAX);
    const float n = __builtin_rintf(u * 0.636619772367581343f);
    float r = fmaf(n, -1.57079637050628662109375f, u);
    r = fmaf(n, 4.37113900018624283e-8f, r);
    const float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    const float s = fmaf(p * z, r, r);
    const float c = fmaf(s * s, -2.0f, 1.0f);
    const int cb = __builtin_bit_cast(int, c) ^ ((int)n << 31);
    const float s2 = fmaf(__builtin_bit_cast(float, cb), -0.5f, 0.5f);
    return fmaf(inv_alpha, s2, h);
}
template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, int reps, float thr) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 1e-6f * threadIdx.x;
    unsigned pacc = 0;
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) {  // e32 with vcc, vcc written once per round
            asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(v[0]), "v"(thr) : "vcc");
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 7]) : "v"(1.0f) : "vcc");
        }
        if (KIND == 1) {  // e64 with an SGPR pair
            unsigned long long m;
            asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(v[0]), "v"(thr));
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(1.0f), "s"(m));
        }
        if (KIND == 2) {  // compiled: 32 x (ok ? x : 0) with per-element conditions, as the conv front writes it
#pragma unroll
            for (int i = 0; i < 256; ++i) v[i & 7] = (v[(i + 3) & 7] < thr + i) ? v[i & 7] * 0.999f : 0.25f;
        }
        if (KIND == 3) {  // compiled: snake_act on 8 values x 4
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = snake_act(v[i], 1.01f, 0.99f) * 0.5f;
        }
        if (KIND == 5) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(0.99f, sin_squared_b(1.01f * v[i]), v[i]) * 0.5f;
        }
        if (KIND == 6) {  // the packed form, two elements per call
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const f32x2 r = snake_act2(f32x2{v[i], v[i + 1]}, f32x2{1.01f, 1.02f}, f32x2{0.99f, 0.98f});
                    v[i] = r.x * 0.5f;
                    v[i + 1] = r.y * 0.5f;
                }
        }
        if (KIND == 7) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = snake_cosform(v[i], 1.01f + 0.01f * (i & 1), 0.99f - 0.01f * (i & 1)) * 0.5f;
        }
        if (KIND == 4) {  // compiled: split2 on 4 pairs x 4
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    unsigned p0, p1, p2;
                    split2(v[i], v[i + 1], p0, p1, p2);
                    pacc ^= p1 ^ p2;
                    v[i] = __builtin_bit_cast(float, p0 & 0xffff0000u) * 0.999f;
                    v[i + 1] = __builtin_bit_cast(float, p0 << 16) * 0.999f;
                }
        }
    }
    float s = (float)pacc;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* out, double units_per_round) {
    const int reps = 500;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-56s", name);
    for (int waves : {4, 8, 16}) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 0, 0, out, reps, 0.7f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(64 * waves), 0, 0, out, reps, 0.7f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d/SIMD: %7.2f ns/unit", waves / 4, ms * 1e6 / (reps * units_per_round * waves / 4));
    }
    printf("\n");
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    run<0>("v_cndmask_b32 e32 (vcc), per instruction", out, 256);
    run<1>("v_cndmask_b32 e64 (sgpr pair), per instruction", out, 256);
    run<2>("compiled select (cmp + mul + cndmask), per element", out, 256);
    run<3>("compiled snake_act (+ 1 mul), per element", out, 32);
    run<5>("snake with magic-number rounding, no select, per element", out, 32);
    run<6>("snake_act2 (packed f32x2), per element", out, 32);
    run<7>("the packed form's arithmetic, scalar, per element", out, 32);
    run<4>("compiled split2 (+ 2 and/shift + 2 mul), per pair", out, 16);
    return 0;
}
