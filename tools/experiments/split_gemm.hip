// Experiment: fp32-grade GEMM on the bf16 matrix cores by operand splitting.
//
//   a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)   (3 x 8 significant bits = the
//   24-bit fp32 significand, exactly), same for w; a.w ~= sum over the six plane pairs with i + j <= 2 of a_i.w_j:
//   every kept product is exact in fp32 (8 x 8 bits), the dropped ones are <= 2^-26 |a.w| — below fp32's own product
//   rounding (2^-24).  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles
//   each) per 32x32x16 block: 2.67x fewer matrix-core cycles.
//
// Measures (a) error against an fp64 reference next to the k-ordered fp32 fmaf chain (= v_mfma_f32_32x32x2_f32
// numerics), (b) throughput on the dominant L3AC GEMM shapes, for several kernel structures (results: DESIGN.md §3.1):
//   mode 0/1  both operands through LDS planes (256 x 128 tile, 8 waves); 1 = low-order products in their own accumulator
//             (3x smaller error, not needed: mode 0 is already below the fp32 chain)
//   mode 2    "v2" = the shipped kernel: A fragments straight from global memory, split in registers; tile-ordered W image
//             through LDS, register-staged, double-buffered; compile-time DIAG bits switch single components off
//   mode 3    "v3" = v2 with W staged by LDS-DMA (global_load_lds_dwordx4), no staging VGPRs
//   mode 4    "v5" = k tile of 16, three DMA-fed LDS stages, 117 VGPRs -> four blocks per CU
//   mode 5    "v6" = v5 with 64 rows per wave (half the LDS reads / W staging per MFMA), two waves per SIMD
//   mode 6    "v7" = 64 rows per wave on the shipped k-32 image (three 24-KB DMA stages, A fetched in k-16 halves)
//   mode 7    "v8" = v2 made persistent: the next tile's first fetches are issued before the current tile's epilogue
// All of 2..7 land within a few % of each other (140-157 TFLOP/s fp32-equivalent); MFMA-only (DIAG 31) reaches 296.
//
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/split_gemm.hip -o tools/experiments/_build/split_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 struct arrays are not promoted to registers)

constexpr int BM = 256, BN = 128, BK = 32, THREADS = 512;
constexpr int A_PLANE = BM * 64;  // bytes of one bf16 plane of the A tile (64-B rows)
constexpr int W_PLANE = BN * 64;
constexpr int STAGE = 3 * A_PLANE + 3 * W_PLANE;

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// two fp32 -> three packed bf16 pairs (round to nearest even at every level)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    f32x2 v = {x0, x1};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __builtin_bit_cast(float, p0 << 16), x1 - __builtin_bit_cast(float, p0 & 0xffff0000u)};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 q = {r.x - __builtin_bit_cast(float, p1 << 16), r.y - __builtin_bit_cast(float, p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

// w [n][k] fp32 -> planes [3][n][k] bf16
__global__ void split_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ planes, int64_t numel) {
    const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= numel) return;
    unsigned p0, p1, p2;
    split2(w[i], w[i + 1], p0, p1, p2);
    *reinterpret_cast<unsigned*>(planes + i) = p0;
    *reinterpret_cast<unsigned*>(planes + numel + i) = p1;
    *reinterpret_cast<unsigned*>(planes + 2 * numel + i) = p2;
}

// w [n][k] fp32 -> tile-ordered LDS images: [n / 128][k / 32][plane 3][128 rows x 64 B, chunks XOR-swizzled] so that
// one k tile of one column block is 24 KB contiguous and is copied to LDS verbatim
__global__ void split_tiles_kernel(const float* __restrict__ w, unsigned char* __restrict__ img, int n, int k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-B chunk (8 k values of one row)
    const int chunks_per_row = k / 8;
    if (i >= (int64_t)n * chunks_per_row) return;
    const int row = (int)(i / chunks_per_row), kc = (int)(i % chunks_per_row);
    const float* src = w + (int64_t)row * k + 8 * kc;
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(src[2 * j], src[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    const int nb = row / 128, r = row % 128, kt = kc / 4, ch = kc % 4;
    unsigned char* tile = img + ((int64_t)nb * (k / 32) + kt) * (3 * 128 * 64);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<uint4*>(tile + pl * 128 * 64 + tile_off(r, ch)) = make_uint4(p[pl][0], p[pl][1], p[pl][2], p[pl][3]);
}

// c[m][n] = sum_k a[m][k] w[n][k]; a fp32 (split while staging), w pre-split planes; k % 32 == 0, m % 256 == 0, n % 128 == 0
template <int MODE>  // 0: six products, one accumulator; 1: six products, low-order terms in their own accumulator
__global__ __launch_bounds__(THREADS) void split_gemm_kernel(const float* __restrict__ a, const uint16_t* __restrict__ wp,
                                                             float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n_blocks = n / BN;
    const int64_t m_panels = m / BM;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * BM;
    const int n0 = (int)(in_group / panels_here) * BN;

    // staging roles
    const int ac4 = tid & 7, ar0 = tid >> 3;     // A: float4 column, rows ar0 + 64 i
    const int wch = tid & 3, wr = tid >> 2;      // W: 16-B chunk, row
    const float* a_src = a + (m0 + ar0) * k + 4 * ac4;
    const int64_t plane_elems = (int64_t)n * k;
    const uint16_t* w_src = wp + (int64_t)(n0 + wr) * k + 8 * wch;

    float4 a_reg[4];
    u32x4 w_reg[3];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_reg[i] = *reinterpret_cast<const float4*>(a_src + (int64_t)64 * i * k + kt * BK);
#pragma unroll
        for (int p = 0; p < 3; ++p) w_reg[p] = *reinterpret_cast<const u32x4*>(w_src + p * plane_elems + kt * BK);
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ar0 + 64 * i;
            unsigned lo0, lo1, lo2, hi0, hi1, hi2;
            split2(a_reg[i].x, a_reg[i].y, lo0, lo1, lo2);
            split2(a_reg[i].z, a_reg[i].w, hi0, hi1, hi2);
            const int off = tile_off(row, ac4 >> 1) + 8 * (ac4 & 1);
            *reinterpret_cast<uint2*>(base + off) = make_uint2(lo0, hi0);
            *reinterpret_cast<uint2*>(base + A_PLANE + off) = make_uint2(lo1, hi1);
            *reinterpret_cast<uint2*>(base + 2 * A_PLANE + off) = make_uint2(lo2, hi2);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<u32x4*>(base + 3 * A_PLANE + p * W_PLANE + tile_off(wr, wch)) = w_reg[p];
    };

    f32x16 acc[2][2], lo[MODE == 1 ? 2 : 1][MODE == 1 ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (MODE == 1) lo[i][j][r] = 0.f;
            }

    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = k / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < n_tiles;
        if (more) load_tile(kt + 1);
        const unsigned char* as = smem + buf * STAGE;
        const unsigned char* ws = as + 3 * A_PLANE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[i][p] = *reinterpret_cast<const bf16x8*>(as + p * A_PLANE + tile_off(64 * wm + 32 * i + li, 2 * s + lh));
                    bf[i][p] = *reinterpret_cast<const bf16x8*>(ws + p * W_PLANE + tile_off(64 * wn + 32 * i + li, 2 * s + lh));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16& l = MODE == 1 ? lo[i][j] : acc[i][j];
                    l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], l, 0, 0, 0);
                    l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], l, 0, 0, 0);
                    l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], l, 0, 0, 0);
                    l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], l, 0, 0, 0);
                    l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], l, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
            if (s == 0 && more) store_tile(buf ^ 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                c[row * n + col] = MODE == 1 ? acc[i][j][r] + lo[i][j][r] : acc[i][j][r];
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// v2: 4 waves, tile 128 x 128, wave w owns rows [32w, 32w+32) x 128 columns.  A never touches LDS: every lane loads the
// 16 fp32 of ITS fragment rows straight from global memory (one 128-B line per two lanes), splits them in registers;
// only the pre-split W planes are staged through LDS (double-buffered).  A is prefetched two k tiles ahead.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int V2_W_PLANE = 128 * 64;
constexpr int V2_STAGE = 3 * V2_W_PLANE;
#ifndef STORE_AT
#define STORE_AT 3
#endif
#ifndef V2_WAVES
#define V2_WAVES 4
#endif
#ifndef V2_OCC
#define V2_OCC 3
#endif
constexpr int V2_T = 64 * V2_WAVES, V2_BM = 32 * V2_WAVES, V2_WLD = 1536 / V2_T;  // W chunks (16 B) per thread per tile

// DIAG bit 32: s_memtime stamps per wave, summed into g_stamps: [0] A wait + split + load issue, [1] MFMA loop (incl. B reads
// and the W store), [2] barrier, [3] epilogue, [4] whole wave, [5] waves
__device__ unsigned long long* g_stamp_buf;  // [waves][5], plain stores (atomics on one word would perturb the run)

template <int DIAG>  // diagnostic bits: 1 no A loads, 2 no split math, 4 no B fragment reads, 8 no W staging, 16 no C store
__global__ __launch_bounds__(V2_T, V2_OCC) void split_gemm_v2_kernel(const float* __restrict__ a, const uint16_t* __restrict__ wp,
                                                               float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = n / 128;
    const int64_t m_panels = m / V2_BM;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * V2_BM;
    const int n0 = (int)(in_group / panels_here) * 128;
    const int li = lane & 31, lh = lane >> 5;

    const float* a_src = a + (m0 + 32 * wave + li) * k + 8 * lh;
    const unsigned char* w_src = reinterpret_cast<const unsigned char*>(wp) + (int64_t)(n0 / 128) * (k / 32) * V2_STAGE + 16 * tid;

    float4 a_pre[2][4];
    u32x4 w_reg[V2_WLD];
    auto load_a = [&](int kt, float4 (&dst)[4]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 32;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
        dst[2] = *reinterpret_cast<const float4*>(src + 16);
        dst[3] = *reinterpret_cast<const float4*>(src + 20);
    };
    auto load_w = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < V2_WLD; ++i) w_reg[i] = *reinterpret_cast<const u32x4*>(w_src + (int64_t)kt * V2_STAGE + 16 * V2_T * i);
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * V2_STAGE;
#pragma unroll
        for (int i = 0; i < V2_WLD; ++i) *reinterpret_cast<u32x4*>(base + 16 * tid + 16 * V2_T * i) = w_reg[i];
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    unsigned long long t_split = 0, t_mfma = 0, t_bar = 0, t_begin = 0;
    if (DIAG & 32) t_begin = __builtin_amdgcn_s_memtime();
    const int n_tiles = k / 32;
    load_a(0, a_pre[0]);
    if (n_tiles > 1) load_a(1, a_pre[1]);
    load_w(0);
    store_w(0);
    __syncthreads();
    auto read_b = [&](const unsigned char* ws, int s, int j, bf16x8 (&b)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(ws + p * V2_W_PLANE + tile_off(32 * j + li, 2 * s + lh));
    };
    // straight-line body (tile indices are clamped instead of branching: hipcc's s_waitcnt insertion stays exact)
    auto step = [&](int kt, float4 (&cur)[4]) __attribute__((always_inline)) {
        const int buf = kt & 1;
        const int last = n_tiles - 1;
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (DIAG & 32) s0 = __builtin_amdgcn_s_memtime();
        if (!(DIAG & 8)) load_w(kt + 1 < last ? kt + 1 : last);
        u32x4 af[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (DIAG & 2) {
                af[s][0] = __builtin_bit_cast(u32x4, cur[2 * s]);
                af[s][1] = __builtin_bit_cast(u32x4, cur[2 * s + 1]);
                af[s][2] = __builtin_bit_cast(u32x4, cur[2 * s]);
                continue;
            }
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(cur[2 * s].x, cur[2 * s].y, x0, x1, x2);
            split2(cur[2 * s].z, cur[2 * s].w, y0, y1, y2);
            split2(cur[2 * s + 1].x, cur[2 * s + 1].y, z0, z1, z2);
            split2(cur[2 * s + 1].z, cur[2 * s + 1].w, u0, u1, u2);
            af[s][0] = u32x4{x0, y0, z0, u0};
            af[s][1] = u32x4{x1, y1, z1, u1};
            af[s][2] = u32x4{x2, y2, z2, u2};
        }
        if (!(DIAG & 1)) load_a(kt + 2 < last ? kt + 2 : last, cur);  // the registers are free again: two tiles ahead
        const unsigned char* ws = smem + buf * V2_STAGE;
        bf16x8 bq[2][3];
        if (DIAG & 32) {
            asm volatile("" : "+v"(af[0][0]), "+v"(af[1][2]));  // the split is done before the stamp
            s1 = __builtin_amdgcn_s_memtime();
        }
        read_b(ws, 0, 0, bq[0]);
#ifdef V2_PRIO
        __builtin_amdgcn_s_setprio(V2_PRIO);
#endif
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int s = it >> 2, j = it & 3;
            if (it + 1 < 8 && !(DIAG & 4)) read_b(ws, (it + 1) >> 2, (it + 1) & 3, bq[(it + 1) & 1]);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[s][0]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[s][1]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[s][2]);
            const int bi = (DIAG & 4) ? 0 : (it & 1);
            const bf16x8 b0 = bq[bi][0], b1 = bq[bi][1], b2 = bq[bi][2];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
            if (it == STORE_AT && !(DIAG & 8)) store_w(buf ^ 1);
        }
#ifdef V2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (DIAG & 32) {
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));  // MFMAs retired
            s2 = __builtin_amdgcn_s_memtime();
        }
        __syncthreads();
        if (DIAG & 32) {
            const unsigned long long s3 = __builtin_amdgcn_s_memtime();
            t_split += s1 - s0;
            t_mfma += s2 - s1;
            t_bar += s3 - s2;
        }
    };
    for (int kt = 0; kt < n_tiles; kt += 2) {
        step(kt, a_pre[0]);
        if (kt + 1 < n_tiles) step(kt + 1, a_pre[1]);
    }
    unsigned long long t_epi0 = 0;
    if (DIAG & 32) t_epi0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (!(DIAG & 16) || acc[j][r] == 1234.5f) c[row * n + col] = acc[j][r];
        }
    }
    if (DIAG & 32) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* o = g_stamp_buf + ((int64_t)blockIdx.x * V2_WAVES + wave) * 5;
            o[0] = t_split;
            o[1] = t_mfma;
            o[2] = t_bar;
            o[3] = t_end - t_epi0;
            o[4] = t_end - t_begin;
        }
    }
}

template <int DIAG>  // diagnostic bits: 1 no A loads, 2 no split math, 4 no B fragment reads, 8 no W staging, 16 no C store
__global__ __launch_bounds__(V2_T, V2_OCC) void split_gemm_v3_kernel(const float* __restrict__ a, const uint16_t* __restrict__ wp,
                                                               float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = n / 128;
    const int64_t m_panels = m / V2_BM;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * V2_BM;
    const int n0 = (int)(in_group / panels_here) * 128;
    const int li = lane & 31, lh = lane >> 5;

    const float* a_src = a + (m0 + 32 * wave + li) * k + 8 * lh;
    const unsigned char* w_src = reinterpret_cast<const unsigned char*>(wp) + (int64_t)(n0 / 128) * (k / 32) * V2_STAGE + 16 * tid;

    float4 a_pre[2][4];
    // W tiles go global -> LDS by DMA (no staging VGPRs, no ds_write): wave-uniform LDS base in M0, lane * 16 B added by HW
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma16 = [&](const unsigned char* src, unsigned lds_byte) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_byte)
                     : "memory");
    };
    auto load_a = [&](int kt, float4 (&dst)[4]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 32;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
        dst[2] = *reinterpret_cast<const float4*>(src + 16);
        dst[3] = *reinterpret_cast<const float4*>(src + 20);
    };
    auto dma_w = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < V2_WLD; ++i)
            dma16(w_src + (int64_t)kt * V2_STAGE + 16 * V2_T * i,
                  __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * V2_STAGE + 16 * V2_T * i + 1024 * wave)));
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int n_tiles = k / 32;
    load_a(0, a_pre[0]);
    if (n_tiles > 1) load_a(1, a_pre[1]);
    dma_w(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto read_b = [&](const unsigned char* ws, int s, int j, bf16x8 (&b)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(ws + p * V2_W_PLANE + tile_off(32 * j + li, 2 * s + lh));
    };
    // straight-line body (tile indices are clamped instead of branching: hipcc's s_waitcnt insertion stays exact)
    auto step = [&](int kt, float4 (&cur)[4]) __attribute__((always_inline)) {
        const int buf = kt & 1;
        const int last = n_tiles - 1;
        if (!(DIAG & 8)) dma_w(kt + 1 < last ? kt + 1 : last, buf ^ 1);
        u32x4 af[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (DIAG & 2) {
                af[s][0] = __builtin_bit_cast(u32x4, cur[2 * s]);
                af[s][1] = __builtin_bit_cast(u32x4, cur[2 * s + 1]);
                af[s][2] = __builtin_bit_cast(u32x4, cur[2 * s]);
                continue;
            }
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(cur[2 * s].x, cur[2 * s].y, x0, x1, x2);
            split2(cur[2 * s].z, cur[2 * s].w, y0, y1, y2);
            split2(cur[2 * s + 1].x, cur[2 * s + 1].y, z0, z1, z2);
            split2(cur[2 * s + 1].z, cur[2 * s + 1].w, u0, u1, u2);
            af[s][0] = u32x4{x0, y0, z0, u0};
            af[s][1] = u32x4{x1, y1, z1, u1};
            af[s][2] = u32x4{x2, y2, z2, u2};
        }
        if (!(DIAG & 1)) load_a(kt + 2 < last ? kt + 2 : last, cur);  // the registers are free again: two tiles ahead
        const unsigned char* ws = smem + buf * V2_STAGE;
        bf16x8 bq[2][3];
        read_b(ws, 0, 0, bq[0]);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int s = it >> 2, j = it & 3;
            if (it + 1 < 8 && !(DIAG & 4)) read_b(ws, (it + 1) >> 2, (it + 1) & 3, bq[(it + 1) & 1]);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[s][0]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[s][1]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[s][2]);
            const int bi = (DIAG & 4) ? 0 : (it & 1);
            const bf16x8 b0 = bq[bi][0], b1 = bq[bi][1], b2 = bq[bi][2];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
        }
        // own DMA pieces of the next tile have landed (only the 4 younger A loads may still be in flight), then everybody's
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
    };
    for (int kt = 0; kt < n_tiles; kt += 2) {
        step(kt, a_pre[0]);
        if (kt + 1 < n_tiles) step(kt + 1, a_pre[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (!(DIAG & 16) || acc[j][r] == 1234.5f) c[row * n + col] = acc[j][r];
        }
    }
}

template <int DIAG>
float run_v2(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    const unsigned grid = (unsigned)((m / V2_BM) * (n / 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v2_kernel<DIAG>, dim3(grid), dim3(V2_T), 2 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v2_kernel<DIAG>, dim3(grid), dim3(V2_T), 2 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int MODE>
float run(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * STAGE));
    const unsigned grid = (unsigned)((m / BM) * (n / BN));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_kernel<MODE>, dim3(grid), dim3(THREADS), 2 * STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(split_gemm_kernel<MODE>, dim3(grid), dim3(THREADS), 2 * STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int DIAG>
float run_v3(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    const unsigned grid = (unsigned)((m / V2_BM) * (n / 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v3_kernel<DIAG>, dim3(grid), dim3(V2_T), 2 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v3_kernel<DIAG>, dim3(grid), dim3(V2_T), 2 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int MODE>
float run_unused(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * STAGE));
    const unsigned grid = (unsigned)((m / BM) * (n / BN));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_kernel<MODE>, dim3(grid), dim3(THREADS), 2 * STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(split_gemm_kernel<MODE>, dim3(grid), dim3(THREADS), 2 * STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}


// ---------------------------------------------------------------------------------------------------------------------
// v5: k tile of 16 (one MFMA k step per barrier), W by LDS-DMA through THREE 12-KB stages two tiles ahead, 128 VGPRs ->
// four blocks (16 waves) per CU.  Image: [n/128][k/16][plane][128 rows x 32 B], 16-B chunk c of row r stored at
// c ^ ((r >> 3) & 1) (conflict-free b128 reads of 16 consecutive rows).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int V5_PLANE = 128 * 32;
constexpr int V5_STAGE = 3 * V5_PLANE;  // 12 KB
__device__ __host__ __forceinline__ int v5_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

__global__ void split_tiles16_kernel(const float* __restrict__ w, unsigned char* __restrict__ img, int n, int k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-B chunk (8 k values of one row)
    const int chunks_per_row = k / 8;
    if (i >= (int64_t)n * chunks_per_row) return;
    const int row = (int)(i / chunks_per_row), kc = (int)(i % chunks_per_row);
    const float* src = w + (int64_t)row * k + 8 * kc;
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(src[2 * j], src[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    const int nb = row / 128, r = row % 128, kt = kc / 2, ch = kc % 2;
    unsigned char* tile = img + ((int64_t)nb * (k / 16) + kt) * V5_STAGE;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(tile + pl * V5_PLANE + v5_off(r, ch)) = u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
}

#ifndef V5_OCC
#define V5_OCC 4
#endif
__global__ __launch_bounds__(256, V5_OCC) void split_gemm_v5_kernel(const float* __restrict__ a, const unsigned char* __restrict__ wimg,
                                                                  float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = n / 128;
    const int64_t m_panels = m / 128;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * 128;
    const int n0 = (int)(in_group / panels_here) * 128;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = k / 16;
    const int last = n_tiles - 1;

    const float* a_src = a + (m0 + 32 * wave + li) * k + 8 * lh;
    // this wave's 3 KB slice of every 12-KB tile image
    const unsigned char* w_src = wimg + (int64_t)(n0 / 128) * n_tiles * V5_STAGE + 3072 * wave + 16 * lane;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma16 = [&](const unsigned char* src, unsigned lds_byte) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_byte)
                     : "memory");
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            dma16(w_src + (int64_t)kt * V5_STAGE + 1024 * i,
                  __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * V5_STAGE + 3072 * wave + 1024 * i)));
    };
    float4 a_pre[2][2];
    auto load_a = [&](int kt, float4 (&dst)[2]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 16;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // prologue: W(0), W(1) in flight to stages 0, 1; A(0), A(1) in registers
    dma_w(0, 0);
    load_a(0, a_pre[0]);
    dma_w(last < 1 ? last : 1, 1);
    load_a(last < 1 ? last : 1, a_pre[1]);
    int stage = 0;  // kt % 3
    auto step = [&](int kt, float4 (&cur)[2]) __attribute__((always_inline)) {
        // own pieces of W(kt) have landed: younger in the queue are A(kt) [2], W(kt+1) [3], A(kt+1) [2]
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __syncthreads();  // everybody's pieces of W(kt) landed; everybody is done with stage (kt + 2) % 3 (read at step kt - 1)
        const int nstage = stage == 0 ? 2 : stage - 1;  // (kt + 2) % 3
        dma_w(kt + 2 < last ? kt + 2 : last, nstage);
        unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
        split2(cur[0].x, cur[0].y, x0, x1, x2);
        split2(cur[0].z, cur[0].w, y0, y1, y2);
        split2(cur[1].x, cur[1].y, z0, z1, z2);
        split2(cur[1].z, cur[1].w, u0, u1, u2);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, u32x4{x0, y0, z0, u0});
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, u32x4{x1, y1, z1, u1});
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, u32x4{x2, y2, z2, u2});
        load_a(kt + 2 < last ? kt + 2 : last, cur);
        const unsigned char* ws = smem + stage * V5_STAGE;
        bf16x8 bq[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][p] = *reinterpret_cast<const bf16x8*>(ws + p * V5_PLANE + v5_off(li, lh));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bq[(j + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(ws + p * V5_PLANE + v5_off(32 * (j + 1) + li, lh));
            }
            const bf16x8 b0 = bq[j & 1][0], b1 = bq[j & 1][1], b2 = bq[j & 1][2];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
    };
    for (int kt = 0; kt < n_tiles; kt += 2) {
        step(kt, a_pre[0]);
        if (kt + 1 < n_tiles) step(kt + 1, a_pre[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
            c[row * n + col] = acc[j][r];
        }
    }
}

float run_v5(const float* a, const unsigned char* wimg, float* c, int64_t m, int n, int k, int iters) {
    const unsigned grid = (unsigned)((m / 128) * (n / 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v5_kernel, dim3(grid), dim3(256), 3 * V5_STAGE, 0, a, wimg, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v5_kernel, dim3(grid), dim3(256), 3 * V5_STAGE, 0, a, wimg, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

#ifndef V6_OCC
#define V6_OCC 2
#endif
__global__ __launch_bounds__(256, V6_OCC) void split_gemm_v6_kernel(const float* __restrict__ a, const unsigned char* __restrict__ wimg,
                                                                  float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = n / 128;
    const int64_t m_panels = m / 256;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * 256;
    const int n0 = (int)(in_group / panels_here) * 128;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = k / 16;
    const int last = n_tiles - 1;

    const float* a_src = a + (m0 + 64 * wave + li) * k + 8 * lh;  // row tile 1: + 32 rows
    // this wave's 3 KB slice of every 12-KB tile image
    const unsigned char* w_src = wimg + (int64_t)(n0 / 128) * n_tiles * V5_STAGE + 3072 * wave + 16 * lane;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma16 = [&](const unsigned char* src, unsigned lds_byte) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_byte)
                     : "memory");
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            dma16(w_src + (int64_t)kt * V5_STAGE + 1024 * i,
                  __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * V5_STAGE + 3072 * wave + 1024 * i)));
    };
    float4 a_pre[2][4];
    auto load_a = [&](int kt, float4 (&dst)[4]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 16;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
        dst[2] = *reinterpret_cast<const float4*>(src + (int64_t)32 * k);
        dst[3] = *reinterpret_cast<const float4*>(src + (int64_t)32 * k + 4);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    // prologue: W(0), W(1) in flight to stages 0, 1; A(0), A(1) in registers
    dma_w(0, 0);
    load_a(0, a_pre[0]);
    dma_w(last < 1 ? last : 1, 1);
    load_a(last < 1 ? last : 1, a_pre[1]);
    int stage = 0;  // kt % 3
    auto step = [&](int kt, float4 (&cur)[4]) __attribute__((always_inline)) {
        // own pieces of W(kt) have landed: younger in the queue are A(kt) [4], W(kt+1) [3], A(kt+1) [4]
        asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        __syncthreads();  // everybody's pieces of W(kt) landed; everybody is done with stage (kt + 2) % 3 (read at step kt - 1)
        const int nstage = stage == 0 ? 2 : stage - 1;  // (kt + 2) % 3
        dma_w(kt + 2 < last ? kt + 2 : last, nstage);
        bf16x8 af[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(cur[2 * t].x, cur[2 * t].y, x0, x1, x2);
            split2(cur[2 * t].z, cur[2 * t].w, y0, y1, y2);
            split2(cur[2 * t + 1].x, cur[2 * t + 1].y, z0, z1, z2);
            split2(cur[2 * t + 1].z, cur[2 * t + 1].w, u0, u1, u2);
            af[t][0] = __builtin_bit_cast(bf16x8, u32x4{x0, y0, z0, u0});
            af[t][1] = __builtin_bit_cast(bf16x8, u32x4{x1, y1, z1, u1});
            af[t][2] = __builtin_bit_cast(bf16x8, u32x4{x2, y2, z2, u2});
        }
        load_a(kt + 2 < last ? kt + 2 : last, cur);
        const unsigned char* ws = smem + stage * V5_STAGE;
        bf16x8 bq[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][p] = *reinterpret_cast<const bf16x8*>(ws + p * V5_PLANE + v5_off(li, lh));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bq[(j + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(ws + p * V5_PLANE + v5_off(32 * (j + 1) + li, lh));
            }
            const bf16x8 b0 = bq[j & 1][0], b1 = bq[j & 1][1], b2 = bq[j & 1][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][2], b0, acc[t][j], 0, 0, 0);
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b1, acc[t][j], 0, 0, 0);
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b2, acc[t][j], 0, 0, 0);
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b0, acc[t][j], 0, 0, 0);
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b1, acc[t][j], 0, 0, 0);
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b0, acc[t][j], 0, 0, 0);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    };
    for (int kt = 0; kt < n_tiles; kt += 2) {
        step(kt, a_pre[0]);
        if (kt + 1 < n_tiles) step(kt + 1, a_pre[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + 64 * wave + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                c[row * n + col] = acc[t][j][r];
            }
        }
}

float run_v6(const float* a, const unsigned char* wimg, float* c, int64_t m, int n, int k, int iters) {
    const unsigned grid = (unsigned)((m / 256) * (n / 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v6_kernel, dim3(grid), dim3(256), 3 * V5_STAGE, 0, a, wimg, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v6_kernel, dim3(grid), dim3(256), 3 * V5_STAGE, 0, a, wimg, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}


// ---------------------------------------------------------------------------------------------------------------------
// v7: 64 rows per wave (256 x 128 tile, 4 waves) on the SHIPPED k-32 W image: W by LDS-DMA through three 24-KB stages two
// k steps ahead (one barrier per k-32 step); A fragments are fetched and split in k-16 halves one k-32 step ahead, so
// only 32 VGPRs of raw A are live.  Same per-element operation order as v2 -> bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void split_gemm_v7_kernel(const float* __restrict__ a, const uint16_t* __restrict__ wp,
                                                              float* __restrict__ c, int64_t m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = n / 128;
    const int64_t m_panels = m / 256;
    const int64_t group = blockIdx.x / (8 * n_blocks);
    const int64_t in_group = blockIdx.x % (8 * n_blocks);
    const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
    const int64_t m0 = (group * 8 + in_group % panels_here) * 256;
    const int n0 = (int)(in_group / panels_here) * 128;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = k / 32;
    const int last = n_tiles - 1;

    const float* a_src = a + (m0 + 64 * wave + li) * k + 8 * lh;  // row tile 1: + 32 rows
    const unsigned char* w_src = reinterpret_cast<const unsigned char*>(wp) + (int64_t)(n0 / 128) * n_tiles * V2_STAGE + 6144 * wave + 16 * lane;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma16 = [&](const unsigned char* src, unsigned lds_byte) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_byte)
                     : "memory");
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {  // this wave's 6 KB of the 24-KB tile
#pragma unroll
        for (int i = 0; i < 6; ++i)
            dma16(w_src + (int64_t)kt * V2_STAGE + 1024 * i,
                  __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * V2_STAGE + 6144 * wave + 1024 * i)));
    };
    // raw A of one k-16 half for both row tiles: [half][row tile * 2 + quad]
    float4 a_pre[2][4];
    auto load_a = [&](int kt, int half, float4 (&dst)[4]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 32 + 16 * half;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
        dst[2] = *reinterpret_cast<const float4*>(src + (int64_t)32 * k);
        dst[3] = *reinterpret_cast<const float4*>(src + (int64_t)32 * k + 4);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    // prologue.  Queue order per k step is always [W DMA x6][A half 0 x4][A half 1 x4].
    dma_w(0, 0);
    load_a(0, 0, a_pre[0]);
    load_a(0, 1, a_pre[1]);
    dma_w(last < 1 ? last : 1, 1);
    int stage = 0;
    for (int kt = 0; kt < n_tiles; ++kt) {
        // own pieces of W(kt) have landed: younger than them are A(kt) [8] and W(kt+1) [6]
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        __syncthreads();  // everybody's pieces landed; everybody is done with stage (kt + 2) % 3 (read during step kt - 1)
        const int nstage = stage == 0 ? 2 : stage - 1;
        const unsigned char* ws = smem + stage * V2_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float4(&cur)[4] = a_pre[s];
            bf16x8 af[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
                split2(cur[2 * t].x, cur[2 * t].y, x0, x1, x2);
                split2(cur[2 * t].z, cur[2 * t].w, y0, y1, y2);
                split2(cur[2 * t + 1].x, cur[2 * t + 1].y, z0, z1, z2);
                split2(cur[2 * t + 1].z, cur[2 * t + 1].w, u0, u1, u2);
                af[t][0] = __builtin_bit_cast(bf16x8, u32x4{x0, y0, z0, u0});
                af[t][1] = __builtin_bit_cast(bf16x8, u32x4{x1, y1, z1, u1});
                af[t][2] = __builtin_bit_cast(bf16x8, u32x4{x2, y2, z2, u2});
            }
            if (s == 0) dma_w(kt + 2 < last ? kt + 2 : last, nstage);  // (after the barrier; keeps the queue order)
            load_a(kt + 1 < last ? kt + 1 : last, s, cur);              // one k-32 step ahead
            bf16x8 bq[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[0][p] = *reinterpret_cast<const bf16x8*>(ws + p * V2_W_PLANE + tile_off(li, 2 * s + lh));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j + 1 < 4) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        bq[(j + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(ws + p * V2_W_PLANE + tile_off(32 * (j + 1) + li, 2 * s + lh));
                }
                const bf16x8 b0 = bq[j & 1][0], b1 = bq[j & 1][1], b2 = bq[j & 1][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][2], b0, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b1, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b2, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b0, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b1, acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b0, acc[t][j], 0, 0, 0);
                }
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + 64 * wave + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                c[row * n + col] = acc[t][j][r];
            }
        }
}

float run_v7(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_v7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * V2_STAGE));
    const unsigned grid = (unsigned)((m / 256) * (n / 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v7_kernel, dim3(grid), dim3(256), 3 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v7_kernel, dim3(grid), dim3(256), 3 * V2_STAGE, 0, a, wp, c, m, n, k);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}


// ---------------------------------------------------------------------------------------------------------------------
// v8: v2 made persistent — a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the next tile's first A / W
// fetches are issued BEFORE the current tile's epilogue stores, so the prologue latency hides behind the epilogue.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void split_gemm_v8_kernel(const float* __restrict__ a, const uint16_t* __restrict__ wp,
                                                              float* __restrict__ c, int64_t m, int n, int k, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int n_blocks = n / 128;
    const int64_t m_panels = m / 128;
    const int n_tiles = k / 32;
    const int last = n_tiles - 1;

    const float* a_src = nullptr;
    const unsigned char* w_src = nullptr;
    int64_t m0 = 0;
    int n0 = 0;
    auto locate = [&](int tile) __attribute__((always_inline)) {  // XCD-aware tile order, as v2
        const int64_t group = tile / (8 * n_blocks);
        const int64_t in_group = tile % (8 * n_blocks);
        const int64_t panels_here = (group * 8 + 8 <= m_panels) ? 8 : m_panels - group * 8;
        m0 = (group * 8 + in_group % panels_here) * 128;
        n0 = (int)(in_group / panels_here) * 128;
        a_src = a + (m0 + 32 * wave + li) * k + 8 * lh;
        w_src = reinterpret_cast<const unsigned char*>(wp) + (int64_t)(n0 / 128) * n_tiles * V2_STAGE + 16 * tid;
    };
    float4 a_pre[2][4];
    u32x4 w_reg[6];
    auto load_a = [&](int kt, float4 (&dst)[4]) __attribute__((always_inline)) {
        const float* src = a_src + kt * 32;
        dst[0] = *reinterpret_cast<const float4*>(src);
        dst[1] = *reinterpret_cast<const float4*>(src + 4);
        dst[2] = *reinterpret_cast<const float4*>(src + 16);
        dst[3] = *reinterpret_cast<const float4*>(src + 20);
    };
    auto load_w = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w_reg[i] = *reinterpret_cast<const u32x4*>(w_src + (int64_t)kt * V2_STAGE + 4096 * i);
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * V2_STAGE;
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(base + 16 * tid + 4096 * i) = w_reg[i];
    };
    auto read_b = [&](const unsigned char* ws, int s, int j, bf16x8 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(ws + p * V2_W_PLANE + tile_off(32 * j + li, 2 * s + lh));
    };
    f32x16 acc[4];
    auto step = [&](int kt, float4 (&cur)[4]) __attribute__((always_inline)) {
        const int buf = kt & 1;
        load_w(kt + 1 < last ? kt + 1 : last);
        u32x4 af[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(cur[2 * s].x, cur[2 * s].y, x0, x1, x2);
            split2(cur[2 * s].z, cur[2 * s].w, y0, y1, y2);
            split2(cur[2 * s + 1].x, cur[2 * s + 1].y, z0, z1, z2);
            split2(cur[2 * s + 1].z, cur[2 * s + 1].w, u0, u1, u2);
            af[s][0] = u32x4{x0, y0, z0, u0};
            af[s][1] = u32x4{x1, y1, z1, u1};
            af[s][2] = u32x4{x2, y2, z2, u2};
        }
        load_a(kt + 2 < last ? kt + 2 : last, cur);
        const unsigned char* ws = smem + buf * V2_STAGE;
        bf16x8 bq[2][3];
        read_b(ws, 0, 0, bq[0]);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int s = it >> 2, j = it & 3;
            if (it + 1 < 8) read_b(ws, (it + 1) >> 2, (it + 1) & 3, bq[(it + 1) & 1]);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[s][0]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[s][1]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[s][2]);
            const bf16x8 b0 = bq[it & 1][0], b1 = bq[it & 1][1], b2 = bq[it & 1][2];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
            if (it == 3) store_w(buf ^ 1);
        }
        __syncthreads();
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    locate(tile);
    load_a(0, a_pre[0]);
    load_a(last < 1 ? last : 1, a_pre[1]);
    load_w(0);
    while (true) {
        store_w(0);  // (every wave passed the last barrier of the previous tile: both buffers are free)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int kt = 0; kt < n_tiles; kt += 2) {
            step(kt, a_pre[0]);
            if (kt + 1 < n_tiles) step(kt + 1, a_pre[1]);
        }
        // the tail steps re-fetched the last k tile into a_pre / w_reg (clamped indices): those registers are free now
        const int64_t cm0 = m0;
        const int cn0 = n0;
        const int next = tile + gridDim.x;
        const bool more = next < total_tiles;
        if (more) {  // block-uniform
            locate(next);
            load_a(0, a_pre[0]);
            load_a(last < 1 ? last : 1, a_pre[1]);
            load_w(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cn0 + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = cm0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
                c[row * n + col] = acc[j][r];
            }
        }
        if (!more) break;
        tile = next;
    }
}

float run_v8(const float* a, const uint16_t* wp, float* c, int64_t m, int n, int k, int iters) {
    const int total = (int)((m / 128) * (n / 128));
    const unsigned grid = (unsigned)(total < 768 ? total : 768);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(split_gemm_v8_kernel, dim3(grid), dim3(256), 2 * V2_STAGE, 0, a, wp, c, m, n, k, total);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(split_gemm_v8_kernel, dim3(grid), dim3(256), 2 * V2_STAGE, 0, a, wp, c, m, n, k, total);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    struct Shape { int64_t m; int n, k; };
    const Shape shapes[] = {{230400, 1024, 256}, {230400, 256, 1024}, {46080, 2048, 512}, {46080, 512, 2048}, {46080, 768, 192}};
    const int dist_count = 2;
    for (const Shape& sh : shapes) {
        const int64_t m = sh.m;
        const int n = sh.n, k = sh.k;
        for (int dist = 0; dist < dist_count; ++dist) {
            std::mt19937 rng(1234 + dist);
            std::uniform_real_distribution<float> uni(-1.f, 1.f);
            std::vector<float> ha((size_t)m * k), hw((size_t)n * k);
            for (auto& v : ha) {
                const float u = uni(rng);
                v = dist == 0 ? u : 8.f * u * u * u * u * u;  // dist 1: heavy-tailed magnitudes
            }
            for (auto& v : hw) v = uni(rng) / std::sqrt((float)k) * (dist == 0 ? 1.f : 3.f);
            float *da, *dw, *dc;
            uint16_t* dp;
            CHECK(hipMalloc(&da, ha.size() * 4));
            CHECK(hipMalloc(&dw, hw.size() * 4));
            CHECK(hipMalloc(&dp, hw.size() * 2 * 3));
            CHECK(hipMalloc(&dc, (size_t)m * n * 4));
            CHECK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((hw.size() / 2 + 255) / 256)), dim3(256), 0, 0, dw, dp,
                               (int64_t)hw.size());
            unsigned char* dimg;
            CHECK(hipMalloc(&dimg, hw.size() * 2 * 3));
            hipLaunchKernelGGL(split_tiles_kernel, dim3((unsigned)((hw.size() / 8 + 255) / 256)), dim3(256), 0, 0, dw, dimg, n, k);
            CHECK(hipDeviceSynchronize());
            const uint16_t* dimg16 = reinterpret_cast<const uint16_t*>(dimg);
            unsigned char* dimg5;
            CHECK(hipMalloc(&dimg5, hw.size() * 2 * 3));
            hipLaunchKernelGGL(split_tiles16_kernel, dim3((unsigned)((hw.size() / 8 + 255) / 256)), dim3(256), 0, 0, dw, dimg5, n, k);
            CHECK(hipDeviceSynchronize());
            for (int mode = 2; mode < 8; ++mode) { if (mode == 3 || mode == 4 || mode == 5) continue;
                const int iters = 10;
                const float ms = mode == 0 ? run<0>(da, dp, dc, m, n, k, iters) : mode == 1 ? run<1>(da, dp, dc, m, n, k, iters) : mode == 2 ? run_v2<0>(da, dimg16, dc, m, n, k, iters) : mode == 3 ? run_v3<0>(da, dimg16, dc, m, n, k, iters) : mode == 4 ? run_v5(da, dimg5, dc, m, n, k, iters) : mode == 5 ? run_v6(da, dimg5, dc, m, n, k, iters) : mode == 6 ? run_v7(da, dimg16, dc, m, n, k, iters) : run_v8(da, dimg16, dc, m, n, k, iters);
                // error on sampled rows
                const int rows = 48;
                std::vector<float> hc((size_t)n);
                double max_split = 0, max_chain = 0, rms_split = 0, rms_chain = 0;
                int64_t cnt = 0;
                for (int s = 0; s < rows; ++s) {
                    const int64_t row = (m / rows) * s + (s * 37) % 256;
                    CHECK(hipMemcpy(hc.data(), dc + row * n, (size_t)n * 4, hipMemcpyDeviceToHost));
                    for (int col = 0; col < n; ++col) {
                        double ref = 0, mag = 0;
                        float chain = 0.f;
                        for (int kk = 0; kk < k; ++kk) {
                            const float x = ha[(size_t)row * k + kk], y = hw[(size_t)col * k + kk];
                            ref += (double)x * y;
                            mag += std::fabs((double)x * y);
                            chain = std::fmaf(x, y, chain);
                        }
                        const double es = std::fabs(hc[col] - ref) / mag, ec = std::fabs(chain - ref) / mag;
                        max_split = std::fmax(max_split, es);
                        max_chain = std::fmax(max_chain, ec);
                        rms_split += es * es;
                        rms_chain += ec * ec;
                        ++cnt;
                    }
                }
                printf("m=%lld n=%d k=%d dist=%d mode=%d: %.3f ms  %.1f TFLOP/s(fp32-equivalent)  err/sum|ab|: split max %.3e rms %.3e"
                       " | fp32 fmaf chain max %.3e rms %.3e\n",
                       (long long)m, n, k, dist, mode, ms, 2.0 * m * n * k / ms * 1e-9, max_split, std::sqrt(rms_split / cnt), max_chain,
                       std::sqrt(rms_chain / cnt));
                fflush(stdout);
            }
            if (dist == 0) {
                const size_t nw = (size_t)(m / V2_BM) * (n / 128) * V2_WAVES;
                unsigned long long* dbuf;
                CHECK(hipMalloc(&dbuf, nw * 5 * sizeof(unsigned long long)));
                CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dbuf, sizeof(dbuf)));
                const float ms = run_v2<32>(da, dimg16, dc, m, n, k, 5);
                std::vector<unsigned long long> hb(nw * 5);
                CHECK(hipMemcpy(hb.data(), dbuf, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                CHECK(hipFree(dbuf));
                double st[5] = {0, 0, 0, 0, 0};
                for (size_t i = 0; i < nw; ++i)
                    for (int q = 0; q < 5; ++q) st[q] += (double)hb[i * 5 + q];
                const double w = (double)nw;
                printf("   v2 stamps (%.3f ms with stamps; cycles per wave, %d k steps): A wait+split %.0f  MFMA loop %.0f  barrier %.0f  "
                       "epilogue %.0f  whole wave %.0f\n", ms, k / 32, st[0] / w, st[1] / w, st[2] / w, st[3] / w, st[4] / w);
                fflush(stdout);
            }
            CHECK(hipFree(dimg));
            CHECK(hipFree(dimg5));
            CHECK(hipFree(da));
            CHECK(hipFree(dw));
            CHECK(hipFree(dp));
            CHECK(hipFree(dc));
            if (sh.m > 100000 && dist == 0) break;  // one distribution is enough on the largest operands
        }
    }
    return 0;
}
