// Stand-alone probe (not part of the library): issue cost of instruction kinds on one SIMD with 1 / 2 / 4 waves resident.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/probes/issue_probe2.hip -o /tmp/issue_probe2 && /tmp/issue_probe2
// Each wave runs `reps` rounds of 256 instructions of one kind (8 independent chains), optionally with one scalar instruction
// or one s_waitcnt between every two of them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// KIND 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 v_and_b32, 3 v_cndmask_b32, 4 v_rndne_f32, 5 v_lshlrev_b32, 6 v_sub_f32, 7 v_med3_f32, 8 v_mul_f32
// EXTRA 0 none, 1 one s_add_u32 after every vector instruction, 2 one s_waitcnt lgkmcnt(0) after every vector instruction, 3 s_nop 0
template <int KIND, int EXTRA>
__global__ __launch_bounds__(1024) void probe(float* out, int reps) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 1e-6f * threadIdx.x;
    unsigned sacc = 0;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 256; ++i) {
            float& x = v[i & 7];
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.999f), "v"(1e-3f));
            if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(v[(i + 1) & 7]));
            if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(0xffff0000u));
            if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(1.0f));
            if (KIND == 4) asm volatile("v_rndne_f32 %0, %0" : "+v"(x));
            if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
            if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(1e-3f));
            if (KIND == 7) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(-1e5f), "v"(1e5f));
            if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(0.999f));
            if (EXTRA == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
            if (EXTRA == 2) asm volatile("s_waitcnt lgkmcnt(0)");
            if (EXTRA == 3) asm volatile("s_nop 0");
        }
    }
    float s = (float)sacc;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND, int EXTRA>
void run(const char* name, float* out) {
    const int reps = 500;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-44s", name);
    for (int waves : {4, 8, 16}) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<KIND, EXTRA>), dim3(256), dim3(64 * waves), 0, 0, out, reps);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND, EXTRA>), dim3(256), dim3(64 * waves), 0, 0, out, reps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // ns per vector instruction per SIMD: ms / (reps * 256 * waves / 4)
        printf("  %d/SIMD: %6.2f ns/instr", waves / 4, ms * 1e6 / (reps * 256.0 * waves / 4));
    }
    printf("\n");
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    run<0, 0>("v_fma_f32", out);
    run<1, 0>("v_cvt_pk_bf16_f32", out);
    run<2, 0>("v_and_b32", out);
    run<3, 0>("v_cndmask_b32", out);
    run<4, 0>("v_rndne_f32", out);
    run<5, 0>("v_lshlrev_b32", out);
    run<6, 0>("v_sub_f32", out);
    run<7, 0>("v_med3_f32", out);
    run<8, 0>("v_mul_f32", out);
    run<0, 1>("v_fma_f32 + s_add_u32 each", out);
    run<0, 2>("v_fma_f32 + s_waitcnt each", out);
    run<0, 3>("v_fma_f32 + s_nop 0 each", out);
    run<2, 1>("v_and_b32 + s_add_u32 each", out);
    return 0;
}
