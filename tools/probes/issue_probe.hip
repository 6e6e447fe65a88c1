// Stand-alone probe (not part of the library): do matrix and vector instructions of DIFFERENT waves on one SIMD overlap?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// One workgroup per CU, 16 waves (4 per SIMD).  Each wave runs `reps` rounds of (M dependent v_mfma_f32_16x16x32_bf16 on one
// accumulator, then V v_fma_f32 on 8 independent registers).  Modes:
//   mfma only (V = 0), valu only (M = 0), both in every wave in phase (all waves start together), both with the odd waves
//   starting on their vector phase (out of phase), and "roles": waves 0-1 of a SIMD pure matrix, waves 2-3 pure vector.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int M, int V, int MODE, bool BIG>  // MODE 0: every wave both, in phase; 1: odd waves start with the vector phase; 2: roles by wave
__global__ __launch_bounds__(1024) void probe(float* out, int reps) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    f32x4 acc = {0, 0, 0, 0};
    f32x16 acc32 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 1e-6f * threadIdx.x;
    const float c0 = 0.999f, c1 = 1e-3f;
    // a SIMD's waves: wave ids w, w + 4, w + 8, w + 12 (round-robin placement assumed; "roles" uses (wave >> 2) & 1)
    const bool do_m = MODE != 2 || ((wave >> 2) & 1) == 0;
    const bool do_v = MODE != 2 || ((wave >> 2) & 1) == 1;
    const bool vec_first = MODE == 1 && (((wave >> 2) & 1) == 1);
    __shared__ __attribute__((aligned(16))) unsigned char lds[36 * 3072];
    if (MODE >= 3) {
        for (int i = threadIdx.x; i < 36 * 3072 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
        __syncthreads();
    }
    int piece = 0;
    auto mphase = [&]() __attribute__((always_inline)) {
        if (M > 0 && do_m) {
            if (MODE == 4) {  // fragments read one group ahead
                bf16x8 fc[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fc[pl] = *reinterpret_cast<const bf16x8*>(lds + (piece % 36) * 3072 + 1024 * pl + 16 * (threadIdx.x & 63));
#pragma unroll
                for (int g = 0; g < M / 6; ++g) {
                    bf16x8 fn[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fn[pl] = *reinterpret_cast<const bf16x8*>(lds + ((piece + g + 1) % 36) * 3072 + 1024 * pl + 16 * (threadIdx.x & 63));
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[2], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[1], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[0], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[1], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[0], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[0], b, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fc[pl] = fn[pl];
                }
                piece = (piece + M / 6) % 36;
                return;
            }
            if (MODE >= 3) {
#pragma unroll
                for (int g = 0; g < M / 6; ++g) {
                    bf16x8 f[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(lds + ((piece + g) % 36) * 3072 + 1024 * pl + 16 * (threadIdx.x & 63));
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], b, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                piece = (piece + M / 6) % 36;
                return;
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (BIG) acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
        }
    };
    auto vphase = [&]() __attribute__((always_inline)) {
        if (V > 0 && do_v) {
#pragma unroll
            for (int i = 0; i < V; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], c0, c1);
        }
    };
    if (vec_first) vphase();
    for (int r = 0; r < reps; ++r) {
        mphase();
        __builtin_amdgcn_sched_barrier(0);
        vphase();
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3] + acc32[0] + acc32[7];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int M, int V, int MODE, bool BIG>
void run(const char* name, float* out, int waves) {
    const int reps = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<M, V, MODE, BIG>), dim3(256), dim3(64 * waves), 0, 0, out, reps);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<M, V, MODE, BIG>), dim3(256), dim3(64 * waves), 0, 0, out, reps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s waves/CU %2d  %8.3f ms  = %7.1f ns per round\n", name, waves, ms, ms * 1e6 / reps);
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    for (int waves : {4, 8, 16}) {
        run<72, 0, 0, false>("72 mfma16 only", out, waves);
        run<0, 205, 0, false>("205 valu only", out, waves);
        run<72, 205, 0, false>("72 mfma16 + 205 valu, in phase", out, waves);
        run<72, 205, 1, false>("72 mfma16 + 205 valu, half the waves out of phase", out, waves);
        run<72, 205, 2, false>("roles: half the waves 72 mfma16, half 205 valu", out, waves);
        run<72, 205, 3, false>("72 mfma16 (fragments from LDS, 3 reads per 6) + 205 valu", out, waves);
        run<72, 0, 3, false>("72 mfma16 (fragments from LDS) only", out, waves);
        run<72, 205, 4, false>("72 mfma16 (fragments one group ahead) + 205 valu", out, waves);
        run<72, 0, 4, false>("72 mfma16 (fragments one group ahead) only", out, waves);
        run<36, 0, 0, true>("36 mfma32 only", out, waves);
        run<36, 205, 0, true>("36 mfma32 + 205 valu, in phase", out, waves);
        run<36, 205, 1, true>("36 mfma32 + 205 valu, half out of phase", out, waves);
        run<36, 205, 2, true>("roles: half 36 mfma32, half 205 valu", out, waves);
    }
    return 0;
}
