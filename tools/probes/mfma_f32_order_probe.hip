// Is v_mfma_f32_16x16x4_f32 the same arithmetic as v_mfma_f32_32x32x2_f32 — per output element a k-ordered chain of fused multiply-adds?
// For random A [M][K], B [N][K] (mixed magnitudes, so that every rounding step matters) computes c = sum_k a.b three ways, feeding the
// k values to the instructions in the SAME sequence, and counts elements whose bits differ:
//   ref   fmaf chain on the vector unit, k = 0, 1, 2, ...
//   m32   32x32x2: lanes 0-31 hold k even, lanes 32-63 k odd (k0 = 2 s, k1 = 2 s + 1 per step s)
//   m16   16x16x4: lane group g = lane >> 4 holds k = 4 s + g
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f32_order_probe mfma_f32_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;

__global__ void ref_kernel(const float* a, const float* b, float* c, int n) {  // c[m][n]: 32 x 32
    const int m = threadIdx.x / 32, nn = threadIdx.x % 32;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(a[m * K + k], b[nn * K + k], acc);
    c[m * 32 + nn] = acc;
}
__global__ void m32_kernel(const float* a, const float* b, float* c) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = {};
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * K + 2 * s + h], b[i * K + 2 * s + h], acc, 0, 0, 0);
    // acc[r]: row (r & 3) + 8 (r >> 2) + 4 h, column i
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ void m16_kernel(const float* a, const float* b, float* c) {  // four 16 x 16 tiles (tm, tn)
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    for (int tm = 0; tm < 2; ++tm)
        for (int tn = 0; tn < 2; ++tn) {
            f32x4 acc = {};
            for (int s = 0; s < K / 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(16 * tm + i) * K + 4 * s + g], b[(16 * tn + i) * K + 4 * s + g], acc, 0, 0, 0);
            // acc[r]: row 4 g + r, column i
            for (int r = 0; r < 4; ++r) c[(16 * tm + 4 * g + r) * 32 + 16 * tn + i] = acc[r];
        }
}

int main() {
    std::vector<float> a(32 * K), b(32 * K);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    int bad32 = 0, bad16 = 0, total = 0;
    float *da, *db, *dc;
    hipMalloc(&da, 32 * K * 4); hipMalloc(&db, 32 * K * 4); hipMalloc(&dc, 3 * 1024 * 4);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : a) v = rnd() * std::pow(10.f, (float)(rand() % 7 - 3));
        for (auto& v : b) v = rnd() * std::pow(10.f, (float)(rand() % 5 - 2));
        hipMemcpy(da, a.data(), 32 * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 32 * K * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(ref_kernel, dim3(1), dim3(1024), 0, 0, da, db, dc, 32);
        hipLaunchKernelGGL(m32_kernel, dim3(1), dim3(64), 0, 0, da, db, dc + 1024);
        hipLaunchKernelGGL(m16_kernel, dim3(1), dim3(64), 0, 0, da, db, dc + 2048);
        std::vector<float> c(3 * 1024);
        hipMemcpy(c.data(), dc, 3 * 1024 * 4, hipMemcpyDeviceToHost);
        for (int e = 0; e < 1024; ++e) {
            ++total;
            bad32 += std::memcmp(&c[e], &c[1024 + e], 4) != 0;
            bad16 += std::memcmp(&c[e], &c[2048 + e], 4) != 0;
        }
    }
    printf("%d outputs, K = %d: 32x32x2 differs from the fmaf chain in %d, 16x16x4 in %d\n", total, K, bad32, bad16);
    return 0;
}
