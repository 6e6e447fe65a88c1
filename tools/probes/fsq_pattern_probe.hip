// Stand-alone probe (not part of the library): the quantiser's traffic — 512 B read, 512 + 4 + 24 B written per token — with
// no arithmetic, by HOW the tokens are dealt to blocks, waves and lanes.  Which ordering reaches the box's linear-copy rate?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/fsq_pattern_probe.hip -o tools/probes/fsq_pattern_probe
// ORDER  0: each block walks ONE contiguous token range (fsq_kernel rounds 1-3)
//        1: grid-stride — block b takes token groups b, b + G, b + 2G, ... (the chip sweeps one compact window at a time)
//        2: grid-stride in chunks of 4 consecutive groups
// LINEAR false: 8 lanes per 512-B row, quads round-robin (8 rows x 128 B per wave instruction)
//        true : lane l of a wave loads 16 B at base + 16 l — 1 KiB contiguous per wave instruction (2 tokens), 4 per lane
// DEPTH  groups requested ahead of the one being stored (1 or 2)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                              \
        }                                                              \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// NT bit 0: non-temporal loads, bit 1: non-temporal stores of q, bit 2: non-temporal stores of the side outputs
template <int ORDER, bool LINEAR, int DEPTH, int NT, int SIDE>  // SIDE 0: none, 1: idx + li as fsq_kernel writes them
__global__ __launch_bounds__(256) void traffic(const v4f* __restrict__ x, v4f* __restrict__ q, int* __restrict__ idx, float* __restrict__ li,
                                               long n_groups) {
    // a group = 32 tokens = 1 024 quads; block iteration = one group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long G = gridDim.x;
    long per_block = (n_groups + G - 1) / G;
    auto group_of = [&](long it) -> long {  // the it-th group of this block, or -1
        long g;
        if (ORDER == 0) g = (long)blockIdx.x * per_block + it;
        else if (ORDER == 1) g = it * G + blockIdx.x;
        else if (ORDER == 3) g = (long)(blockIdx.x & 7) * (n_groups / 8) + it * (G / 8) + (blockIdx.x >> 3);  // every XCD sweeps its own eighth
        else g = ((it >> 2) * G + blockIdx.x) * 4 + (it & 3);
        if (ORDER == 0 && it >= per_block) return -1;
        if (ORDER == 3 && it * (G / 8) + (blockIdx.x >> 3) >= n_groups / 8) return -1;
        return g < n_groups ? g : -1;
    };
    auto quad_index = [&](long g, int k) -> long {  // quad of this lane's k-th access in group g
        if (LINEAR) return g * 1024 + wave * 256 + k * 64 + lane;          // wave: 8 tokens = 256 quads, 64 per instruction
        return (g * 32 + (tid >> 3)) * 32 + (tid & 7) + 8 * k;            // row tid/8, quads sub + 8k
    };
    v4f buf[DEPTH][4];
    auto fetch = [&](long it, int slot) {
        const long g = group_of(it);
        if (g >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) buf[slot][k] = (NT & 1) ? __builtin_nontemporal_load(x + quad_index(g, k)) : x[quad_index(g, k)];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, d);
    for (long it = 0;; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long g = group_of(it + d);
            if (g < 0) return;
            v4f cur[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = buf[d][k];
            fetch(it + d + DEPTH, d);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (NT & 2) __builtin_nontemporal_store(cur[k], q + quad_index(g, k));
                else q[quad_index(g, k)] = cur[k];
            }
            if (SIDE == 1) {
                if (LINEAR) {  // wave: tokens g * 32 + 8 wave .. + 7: 32 B of indices, 192 B of level indices, contiguous
                    const long t0 = g * 32 + wave * 8;
                    if (lane < 8) idx[t0 + lane] = __float_as_int(cur[0].x);
                    if (lane < 48) li[t0 * 6 + lane] = cur[0].y;
                } else {
                    const long t = g * 32 + (tid >> 3);
                    const int sub = tid & 7;
                    if (NT & 4) {
                        if (sub == 0) __builtin_nontemporal_store(__float_as_int(cur[0].x), idx + t);
                        if (sub < 6) __builtin_nontemporal_store(cur[0].y, li + t * 6 + sub);
                    } else {
                        if (sub == 0) idx[t] = __float_as_int(cur[0].x);
                        if (sub < 6) li[t * 6 + sub] = cur[0].y;
                    }
                }
            }
        }
    }
}

int main(int argc, char** argv) {
    const long tokens = 1L << 22;
    const long n_groups = tokens / 32;
    v4f *x, *q;
    int* idx;
    float* li;
    CHECK(hipMalloc(&x, tokens * 512));
    CHECK(hipMalloc(&q, tokens * 512));
    CHECK(hipMalloc(&idx, tokens * 4));
    CHECK(hipMalloc(&li, tokens * 24));
    CHECK(hipMemset(x, 1, tokens * 512));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        std::printf("%-64s %8.1f GB/s (%.3f of 8 TB/s)\n", name, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
        std::fflush(stdout);
    };
    const double b_plain = tokens * 1024.0, b_side = tokens * 1052.0;
#define RUN(ORDER, LINEAR, DEPTH, NT, SIDE, PER_CU)                                                                                  \
    {                                                                                                                                \
        char nm[96];                                                                                                                 \
        std::snprintf(nm, sizeof nm, "order %d %s depth %d %s %s, %d blocks/CU", ORDER, LINEAR ? "linear" : "rows8 ", DEPTH,        \
                      NT == 0 ? "    " : NT == 1 ? "ntL " : NT == 2 ? "ntS " : NT == 3 ? "ntLS" : "ntA ", SIDE ? "+idx+li" : "       ", PER_CU);                                                       \
        time(nm, SIDE ? b_side : b_plain, [&] {                                                                                      \
            hipLaunchKernelGGL((traffic<ORDER, LINEAR, DEPTH, NT, SIDE>), dim3(256 * PER_CU), dim3(256), 0, 0, x, q, idx, li, n_groups); \
        });                                                                                                                          \
    }
    for (int round = 0; round < 2; ++round) {
        std::printf("---- round %d\n", round);
        RUN(1, false, 1, 0, 0, 1) RUN(1, false, 1, 0, 0, 2) RUN(1, false, 1, 0, 0, 3) RUN(1, false, 2, 0, 0, 1) RUN(1, false, 2, 0, 0, 2)
        RUN(1, false, 1, 3, 0, 1) RUN(1, false, 1, 3, 0, 2) RUN(1, false, 1, 3, 0, 3) RUN(1, false, 2, 3, 0, 1) RUN(1, false, 2, 3, 0, 2)
        RUN(1, false, 1, 3, 1, 1) RUN(1, false, 1, 3, 1, 2) RUN(1, false, 1, 3, 1, 3) RUN(1, false, 2, 3, 1, 1) RUN(1, false, 2, 3, 1, 2) RUN(1, false, 2, 3, 1, 3)
        RUN(1, false, 1, 1, 1, 3) RUN(1, false, 1, 2, 1, 3) RUN(1, false, 1, 7, 1, 3) RUN(1, false, 1, 7, 1, 2) RUN(1, false, 2, 7, 1, 2)
        RUN(3, false, 1, 3, 1, 2) RUN(3, false, 1, 3, 1, 3) RUN(3, false, 1, 3, 1, 4) RUN(3, false, 1, 0, 1, 3)
        RUN(0, false, 1, 3, 1, 3) RUN(0, false, 1, 3, 1, 8)
    }
    return 0;
}
