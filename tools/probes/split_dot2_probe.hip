// Is the residual of the bf16x3 split — x - float(bf16(x)) — the same bits when it is computed by v_dot2c_f32_bf16 (x + p.lo * -1 + p.hi * 0:
// ONE instruction per element) as by shift / mask + v_sub_f32 (two), for the whole split2 of split_bf16.hpp?  And what does the instruction cost?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/split_dot2_probe.hip -o /tmp/split_dot2_probe && /tmp/split_dot2_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_ref(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const f32x2_t v = {x0, x1};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = {x0 - __builtin_bit_cast(float, p0 << 16), x1 - __builtin_bit_cast(float, p0 & 0xffff0000u)};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t q = {r.x - __builtin_bit_cast(float, p1 << 16), r.y - __builtin_bit_cast(float, p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
}
__device__ __forceinline__ void split2_dot(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    // (the selectors through SGPRs: given as constants hipcc 7.2 encodes {-1, 0} as the INLINE constant -1.0, which the instruction reads as the
    // 32-bit pattern 0xbf800000 = {0, -1} — the first run of this probe took the wrong half)
    unsigned klo_u = 0x0000bf80u, khi_u = 0xbf800000u;
    asm volatile("" : "+s"(klo_u), "+s"(khi_u));
    const bf16x2_t klo = __builtin_bit_cast(bf16x2_t, klo_u), khi = __builtin_bit_cast(bf16x2_t, khi_u);
    const f32x2_t v = {x0, x1};
    const bf16x2_t b0 = __builtin_convertvector(v, bf16x2_t);
    p0 = __builtin_bit_cast(unsigned, b0);
    const f32x2_t r = {__builtin_amdgcn_fdot2_f32_bf16(b0, klo, x0, false), __builtin_amdgcn_fdot2_f32_bf16(b0, khi, x1, false)};
    const bf16x2_t b1 = __builtin_convertvector(r, bf16x2_t);
    p1 = __builtin_bit_cast(unsigned, b1);
    const f32x2_t q = {__builtin_amdgcn_fdot2_f32_bf16(b1, klo, r.x, false), __builtin_amdgcn_fdot2_f32_bf16(b1, khi, r.y, false)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
}
__global__ void check(const float* x, int n, unsigned* out_ref, unsigned* out_dot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a0, a1, a2, b0, b1, b2;
    split2_ref(x[2 * i], x[2 * i + 1], a0, a1, a2);
    split2_dot(x[2 * i], x[2 * i + 1], b0, b1, b2);
    out_ref[3 * i] = a0; out_ref[3 * i + 1] = a1; out_ref[3 * i + 2] = a2;
    out_dot[3 * i] = b0; out_dot[3 * i + 1] = b1; out_dot[3 * i + 2] = b2;
}
template <int MODE>
__global__ void rate(float* y, int rounds) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a * 1.7f;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            unsigned p0, p1, p2;
            if (MODE == 0) split2_ref(a, b, p0, p1, p2); else split2_dot(a, b, p0, p1, p2);
            acc ^= p0 + p1 + p2;
            a += 0.37f; b -= 0.11f;
        }
    }
    y[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::uniform_int_distribution<int> e(-60, 40);
    for (int i = 0; i < n; ++i) {
        if (i % 1024 == 0) h[i] = 0.f;
        else if (i % 1024 == 1) h[i] = -0.f;
        else if (i % 1024 == 2) { uint32_t bits = 0x3f800000u + (uint32_t)g(); bits &= 0x7fffffffu; std::memcpy(&h[i], &bits, 4); if (!std::isfinite(h[i])) h[i] = 1.f; }
        else h[i] = std::ldexp(u(g), e(g));
    }
    float* dx; unsigned *dr, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, (size_t)n / 2 * 12); hipMalloc(&dd, (size_t)n / 2 * 12);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    check<<<n / 2 / 256, 256>>>(dx, n, dr, dd);
    std::vector<unsigned> r((size_t)n / 2 * 3), d((size_t)n / 2 * 3);
    hipMemcpy(r.data(), dr, r.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < r.size(); ++i) if (r[i] != d[i]) { if (bad < 5) std::printf("differs at pair %zu plane %zu: %08x vs %08x (x = %g, %g)\n", i / 3, i % 3, r[i], d[i], h[2 * (i / 3)], h[2 * (i / 3) + 1]); ++bad; }
    std::printf("split2 by v_dot2c_f32_bf16 against shift / mask / subtract: %zu of %zu plane words differ\n", bad, r.size());
    float* dy; hipMalloc(&dy, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<1024, 256>>>(dy, 2000); else rate<1><<<1024, 256>>>(dy, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) std::printf("%s: %.3f ms for 1024 x 256 threads x 32 000 split2 = %.2f ns per split2 per SIMD-wave\n", mode ? "dot2c" : "shift/mask/sub", ms, ms * 1e6 / (2000.0 * 16) / (1024.0 * 4 / 1024));
        }
    }
    return bad != 0;
}
