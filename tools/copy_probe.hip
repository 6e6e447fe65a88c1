// Stand-alone probe (not part of the library): what HBM rate do plain copy kernels reach on THIS box, by access pattern?
// The ceiling the quantiser kernel (fsq_kernel, 512 B in + 540 B out per token) can be priced against.
//   hipcc -O3 --offload-arch=gfx950 tools/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
// Patterns (all 16 B per lane per access, read N bytes + write N bytes):
//   linear      lane l of a wave: base + 16 l            (1 KiB contiguous per wave instruction), grid-stride
//   linear x4   the same, 4 loads in flight per lane before the 4 stores
//   rows8       fsq_kernel's pattern: 8 lanes per 512-B row, quads round-robin (8 rows x 128 B per wave instruction), 4 per lane
//   nt          linear x4 with non-temporal loads and stores
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));           \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_linear(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

template <bool NT>
__global__ __launch_bounds__(256) void copy_linear4(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    // each block walks a contiguous range; per iteration a wave moves 4 consecutive KiB
    const size_t per_block = (n / 1024 + gridDim.x - 1) / gridDim.x * 1024;
    const size_t begin = (size_t)blockIdx.x * per_block, end = begin + per_block < n ? begin + per_block : n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t i = begin + (size_t)wave * 256 + lane; i + 192 < end; i += 1024) {
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = NT ? __builtin_nontemporal_load(a + i + 64 * k) : a[i + 64 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NT)
                __builtin_nontemporal_store(v[k], b + i + 64 * k);
            else
                b[i + 64 * k] = v[k];
        }
    }
}

__global__ __launch_bounds__(256) void copy_rows8(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    // rows of 32 quads (512 B); 8 lanes per row, lane `sub` takes quads sub, sub + 8, sub + 16, sub + 24
    const size_t rows = n / 32;
    const size_t per_block = (rows / 32 + gridDim.x - 1) / gridDim.x * 32;
    const size_t begin = (size_t)blockIdx.x * per_block, end = begin + per_block < rows ? begin + per_block : rows;
    const int sub = threadIdx.x & 7;
    for (size_t r = begin + (threadIdx.x >> 3); r < end; r += 32) {
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[r * 32 + sub + 8 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) b[r * 32 + sub + 8 * k] = v[k];
    }
}

// rows8 plus the quantiser's side outputs: 4 B per row (lane sub == 0) and 24 B per row (lanes sub < 6, contiguous per row)
template <int MODE>  // 0: both side outputs as fsq_kernel writes them, 1: only the 4-B one, 2: only the 24-B one, 3: both staged through LDS and written as whole lines
__global__ __launch_bounds__(256) void copy_rows8_side(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n, int* __restrict__ idx,
                                                       float* __restrict__ li) {
    __shared__ float stage[32 * 7];
    const size_t rows = n / 32;
    const size_t per_block = (rows / 32 + gridDim.x - 1) / gridDim.x * 32;
    const size_t begin = (size_t)blockIdx.x * per_block, end = begin + per_block < rows ? begin + per_block : rows;
    const int sub = threadIdx.x & 7;
    for (size_t r0 = begin; r0 < end; r0 += 32) {
        const size_t r = r0 + (threadIdx.x >> 3);
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[r * 32 + sub + 8 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) b[r * 32 + sub + 8 * k] = v[k];
        if (MODE == 0 || MODE == 1) {
            if (sub == 0) idx[r] = __float_as_int(v[0].x);
        }
        if (MODE == 0 || MODE == 2) {
            if (sub < 6) li[r * 6 + sub] = v[0].y;
        }
        if (MODE == 3) {  // the block's 32 rows: 128 B of indices + 768 B of level indices, contiguous in memory
            if (sub == 0) stage[threadIdx.x >> 3] = v[0].x;
            if (sub < 6) stage[32 + (threadIdx.x >> 3) * 6 + sub] = v[0].y;
            __syncthreads();
            if (threadIdx.x < 32) idx[r0 + threadIdx.x] = __float_as_int(stage[threadIdx.x]);
            if (threadIdx.x < 192) li[r0 * 6 + threadIdx.x] = stage[32 + threadIdx.x];
            __syncthreads();
        }
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB each way
    const size_t n = bytes / 16;
    v4f *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        std::printf("%-34s %8.1f GB/s (read + write; %.1f %% of 8 TB/s)\n", name, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 80.0);
    };
    for (int round = 0; round < 2; ++round) {
        for (int per_cu : {4, 8, 16, 32}) {
            const int grid = 256 * per_cu;
            char nm[64];
            std::snprintf(nm, sizeof nm, "linear, %d blocks/CU", per_cu);
            time(nm, [&] { hipLaunchKernelGGL(copy_linear, dim3(grid), dim3(256), 0, 0, a, b, n); });
        }
        for (int per_cu : {4, 8}) {
            const int grid = 256 * per_cu;
            char nm[64];
            std::snprintf(nm, sizeof nm, "linear x4, %d blocks/CU", per_cu);
            time(nm, [&] { hipLaunchKernelGGL(copy_linear4<false>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            std::snprintf(nm, sizeof nm, "linear x4 nt, %d blocks/CU", per_cu);
            time(nm, [&] { hipLaunchKernelGGL(copy_linear4<true>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            std::snprintf(nm, sizeof nm, "rows8 (fsq pattern), %d blocks/CU", per_cu);
            time(nm, [&] { hipLaunchKernelGGL(copy_rows8, dim3(grid), dim3(256), 0, 0, a, b, n); });
        }
        {
            int* idx;
            float* li;
            CHECK(hipMalloc(&idx, n / 32 * 4));
            CHECK(hipMalloc(&li, n / 32 * 24));
            for (int per_cu : {4, 8}) {
                const int grid = 256 * per_cu;
                char nm[64];
                std::snprintf(nm, sizeof nm, "rows8 + idx + li, %d blocks/CU", per_cu);
                time(nm, [&] { hipLaunchKernelGGL(copy_rows8_side<0>, dim3(grid), dim3(256), 0, 0, a, b, n, idx, li); });
                std::snprintf(nm, sizeof nm, "rows8 + idx only, %d blocks/CU", per_cu);
                time(nm, [&] { hipLaunchKernelGGL(copy_rows8_side<1>, dim3(grid), dim3(256), 0, 0, a, b, n, idx, li); });
                std::snprintf(nm, sizeof nm, "rows8 + li only, %d blocks/CU", per_cu);
                time(nm, [&] { hipLaunchKernelGGL(copy_rows8_side<2>, dim3(grid), dim3(256), 0, 0, a, b, n, idx, li); });
                std::snprintf(nm, sizeof nm, "rows8 + staged idx + li, %d blocks/CU", per_cu);
                time(nm, [&] { hipLaunchKernelGGL(copy_rows8_side<3>, dim3(grid), dim3(256), 0, 0, a, b, n, idx, li); });
            }
            CHECK(hipFree(idx));
            CHECK(hipFree(li));
        }
        time("hipMemcpyDtoD", [&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    }
    return 0;
}
