// fp32-grade GEMM on the bf16 matrix cores by operand splitting ("bf16x3"), used for the large channel contractions
// (same call sites as gemm_f32.hip; reference: nn.Linear / nn.Conv1d in l3ac/modules.py:19-36,96-99,161 and the
// local_attention linears).
//
//   c[m][n] = epilogue( sum_k a[m][k] * w[n][k] )
//
// Numerics.  x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round-to-nearest-even:
// three 8-bit significands = the whole 24-bit fp32 significand, so the split loses nothing.  a.w is evaluated as the
// six plane products with i + j <= 2 (a2w0, a1w1, a0w2, a1w0, a0w1, a0w0 — smallest first), each product exact in
// fp32 (8 x 8 bits), accumulated in the MFMA's fp32 accumulator; the three dropped products are <= 2^-26 |a.w|, below
// fp32's own product rounding (2^-24).  Measured against fp64 on the L3AC shapes (tools/experiments/split_gemm.hip (retired: git show 1a7dadb:tools/experiments/split_gemm.hip),
// tests/test_gpu_blocks.py::test_gemm_split_accuracy) the error is <= that of the k-ordered fp32 fmaf chain which
// v_mfma_f32_32x32x2_f32 (gemm_f32.hip) computes: rms 2.0e-8 vs 2.4e-8 of sum|a.w|.  This is NOT a reduced-precision
// path: every bit of both fp32 operands takes part.
//
// Why.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) do the work of eight v_mfma_f32_32x32x2_f32 (64 cycles each):
// 2.67x fewer matrix-core cycles per fp32 MAC; effective peak 2516.6 / 6 = 419 TFLOP/s(fp32-equivalent) vs 157.3.
//
// gfx950 design
//   * block = 4 waves, tile 128 (m) x 128 (n) x 32 (k); wave w owns rows [32w, 32w+32) across the column tiles; 3 blocks per CU.
//   * A never touches LDS: lane (row r, half h) loads the 16 fp32 of ITS MFMA fragments (k = 8h..8h+7 and
//     16+8h..16+8h+7 of the k tile: one 128-B line per two lanes) straight into registers, two k tiles ahead, and
//     splits them there (11 VALU ops per two values, v_cvt_pk_bf16_f32 based).
//   * W is split ONCE at model-build time into a tile-ordered image: [n/128][k/32][plane][128 rows x 64 B], the 16-B
//     chunks XOR-swizzled exactly as the LDS tile wants them, so a k tile of a column block is 24 KB contiguous in
//     HBM/L2 and is copied to LDS verbatim (coalesced 1 KB per wave instruction), double-buffered.
//   * XCD-aware block -> tile order as in gemm_f32.hip.
#include "gemm_split_common.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

// w [n][k] fp32 (row stride ldw) -> tile-ordered split image (device-side builder; network.hip builds the same image on
// the host).  One thread = one 16-B chunk (8 k values of one row) of each plane.
__global__ void split_image_kernel(const float* __restrict__ w, int64_t ldw, int n, int k, unsigned char* __restrict__ img) {
    const int k_tiles = (k + BK - 1) / BK;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t chunks = (int64_t)((n + BN - 1) / BN) * BN * k_tiles * 4;
    if (i >= chunks) return;
    const int row = (int)(i / (k_tiles * 4)), kc = (int)(i % (k_tiles * 4));
    unsigned p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = 8 * kc + 2 * j;
        const float x0 = (row < n && kk < k) ? w[(int64_t)row * ldw + kk] : 0.f;
        const float x1 = (row < n && kk + 1 < k) ? w[(int64_t)row * ldw + kk + 1] : 0.f;
        split2(x0, x1, p[0][j], p[1][j], p[2][j]);
    }
    unsigned char* tile = img + ((int64_t)(row / BN) * k_tiles + kc / 4) * W_TILE;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(tile + pl * W_PLANE + tile_off(row % BN, kc % 4)) = u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
}

// KTAIL (template parameter of the kernels below): k % 32 != 0 (k % 8 == 0): the last tile's out-of-range 8-value groups re-read the
// row's last valid group — finite numbers that meet the image's zero padding.

// ---- the GEMM on v_mfma_f32_16x16x32_bf16 ------------------------------------------------------------------------------
// Tile 128 x 128 x 32, wave w = rows [32w, 32w+32) across all 128 columns, six plane products per MAC on 16x16x32 instructions (16
// cycles each; rounds 1-2 used v_mfma_f32_32x32x16_bf16: the same matrix-core cycles per FLOP).  Why: on
// this power-limited chip the clock a dense MFMA loop holds depends on the instruction's shape — MI355X_MICROARCH.md (DVFS, item
// 7) measures the 16x16x32 loop at 1.12-1.15 x the FLOP/s of the 32x32x16 loop at equal cycles; a timing-only substitution in
// this kernel gave 4.24 -> 3.75 ms per step over its 29 launches.
//   A (activations): lane (m = lane & 15, kg = lane >> 4) owns k = 8 kg .. 8 kg + 7 of rows m and 16 + m of the wave's strip:
//                    2 x 32 B per row and k tile, four lanes per 128-B line.
//   B (weights):     lane (n = lane & 15, kg) reads chunk kg of row 16 t + n of the k tile's planes: the image and its XOR
//                    swizzle are the 32x32 kernel's (any 64 lanes of this pattern touch 64 distinct 16-B slots of one KB).
//   D:               acc[h][t][i] = c[row 16 h + 4 (lane >> 4) + i][column 16 t + (lane & 15)].
// DEEP: the W tiles are requested TWO k tiles ahead through two register sets (the same MFMAs in the same order: same bits).
// With one block per CU — a single clip's products: 8-32 blocks — nothing else covers the half k tile between a W request and
// its LDS store, and every k tile waited ~1.5 us for it; the 24 extra registers would cost the full grids their third block per CU.
// CONV: A is an implicit 1-D convolution over the frames of a clip (GemmArgs::taps / dil / cin / frames, cin % 32 == 0 so that a k
// tile lies inside one tap): k tile kt reads channels [c0, c0 + 32) of frame t + (tap - taps / 2) * dil, zeros outside the clip.
template <bool KTAIL, int RG, bool DEEP, bool CONV = false>
__device__ __forceinline__ void gemm_split_body16(const GemmArgs& p, const int gp) {
    static_assert(!(CONV && KTAIL), "the implicit-conv A operand has whole k tiles (cin % 32 == 0)");
    constexpr int WM = 16 * RG;   // rows per wave
    constexpr int BMW = 4 * WM;   // rows per block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    // XCD-aware tile order (gemm_f32.hip): all column tiles of one A row panel run on one XCD
    const int n_blocks = (p.n + BN - 1) / BN;
    const int64_t m_panels = (p.m + BMW - 1) / BMW;
    const int GP = gp;
    // (32-bit on purpose: the 64-bit forms of these four wave-uniform divisions are ~100 scalar instructions each, a visible part of
    // a short-K block's life; the launcher keeps the grid, hence every quotient, below 2^31)
    const unsigned group = blockIdx.x / (unsigned)(GP * n_blocks);
    const unsigned in_group = blockIdx.x % (unsigned)(GP * n_blocks);
    const unsigned panels_here = ((int64_t)group * GP + GP <= m_panels) ? (unsigned)GP : (unsigned)(m_panels - (int64_t)group * GP);
    const int64_t m0 = ((int64_t)group * GP + in_group % panels_here) * BMW;
    const int n0 = (int)(in_group / panels_here) * BN;
    const int n_tiles = (p.k + BK - 1) / BK;
    const int last = n_tiles - 1;

    // rows past the edge are clamped to row 0: they only feed accumulators that are never stored
    const float* a_row[RG];
    int a_t[RG];  // CONV: the row's frame inside its clip
#pragma unroll
    for (int h = 0; h < RG; ++h) {
        const int64_t row = m0 + WM * wave + 16 * h + ln;
        const int64_t rr = row < p.m ? row : 0;
        a_row[h] = p.a + rr * p.lda + 8 * lg;
        a_t[h] = CONV ? (int)((unsigned)rr % (unsigned)p.frames) : 0;  // (m < 2^31: checked by the launcher)
    }
    const int conv_tiles_per_tap = CONV ? p.cin / BK : 1;
    const int conv_half = CONV ? p.taps >> 1 : 0;
    const unsigned char* w_src = p.w_img + (int64_t)(n0 / BN) * n_tiles * W_TILE + 16 * tid;

    float4 a_pre[2][2 * RG];  // [k tile parity][row group x 2 float4]
    u32x4 w_reg[DEEP ? 2 : 1][W_LOADS];
    auto load_a = [&](int kt, float4 (&dst)[2 * RG]) __attribute__((always_inline)) {
        int o = kt * BK;
        if (KTAIL) {  // the last tile's groups beyond k re-read the row's last valid group: finite values on the image's zero padding
            const int kmax = p.k - 8 - 8 * lg;
            o = o < kmax ? o : kmax;
        }
        if constexpr (CONV) {
            const int tap = kt / conv_tiles_per_tap;  // wave-uniform
            const int shift = (tap - conv_half) * p.dil;
            const int64_t delta = (int64_t)shift * p.lda + (int64_t)(kt - tap * conv_tiles_per_tap) * BK;
#pragma unroll
            for (int h = 0; h < RG; ++h) {
                const int t = a_t[h] + shift;
                const bool ok = t >= 0 && t < (int)p.frames;
                const float* src = a_row[h] + (ok ? delta : 0);  // (outside the clip: any valid address, the values are dropped)
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                dst[2 * h] = ok ? v0 : make_float4(0.f, 0.f, 0.f, 0.f);
                dst[2 * h + 1] = ok ? v1 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < RG; ++h) {
            dst[2 * h] = *reinterpret_cast<const float4*>(a_row[h] + o);
            dst[2 * h + 1] = *reinterpret_cast<const float4*>(a_row[h] + o + 4);
        }
    };
    auto load_w = [&](int kt, auto set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i)
            w_reg[decltype(set)::value][i] = *reinterpret_cast<const u32x4*>(w_src + (int64_t)kt * W_TILE + 16 * THREADS * i);
    };
    auto store_w = [&](int buf, auto set) __attribute__((always_inline)) {
        unsigned char* base = smem_split + buf * W_TILE + 16 * tid;
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i) *reinterpret_cast<u32x4*>(base + 16 * THREADS * i) = w_reg[decltype(set)::value][i];
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;
    // planes read in the order 2, 1, 0: the first MFMA of a column tile takes plane 0 of the weights, the YOUNGEST read, so the one
    // s_waitcnt in front of it covers all three (LDS returns in order) instead of one wait per plane
    auto read_b = [&](const unsigned char* ws, int t, bf16x8 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) b[pl] = *reinterpret_cast<const bf16x8*>(ws + pl * W_PLANE + tile_off(16 * t + ln, lg));
    };

    f32x4a acc[RG][8];
#pragma unroll
    for (int h = 0; h < RG; ++h)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[h][t] = f32x4a{0.f, 0.f, 0.f, 0.f};

    load_a(0, a_pre[0]);
    load_a(last < 1 ? last : 1, a_pre[1]);
    load_w(0, Set0{});
    store_w(0, Set0{});
    if (DEEP) load_w(last < 1 ? last : 1, Set1{});  // tile 1 waits in the second set for step 0's store
    __syncthreads();
    // straight-line body: tile indices are clamped instead of branched on, which keeps hipcc's s_waitcnt counts exact
    // (odd: the step's parity — which register set receives this step's W request and which one is stored)
    auto step = [&](int kt, float4 (&cur)[2 * RG], auto odd) __attribute__((always_inline)) {
        using Mine = std::integral_constant<int, DEEP ? decltype(odd)::value : 0>;       // receives tile kt + 2 (DEEP) / kt + 1
        using Next = std::integral_constant<int, DEEP ? 1 - decltype(odd)::value : 0>;   // holds tile kt + 1: stored in this step
        const int buf = kt & 1;
        if (DEEP) {  // (tile kt + 1 was requested during step kt - 1 / in the prologue)
            load_w(kt + 2 < last ? kt + 2 : last, Mine{});
        } else {
            load_w(kt + 1 < last ? kt + 1 : last, Mine{});
        }
        u32x4 af[RG][3];
#pragma unroll
        for (int h = 0; h < RG; ++h) {
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(cur[2 * h].x, cur[2 * h].y, x0, x1, x2);
            split2(cur[2 * h].z, cur[2 * h].w, y0, y1, y2);
            split2(cur[2 * h + 1].x, cur[2 * h + 1].y, z0, z1, z2);
            split2(cur[2 * h + 1].z, cur[2 * h + 1].w, u0, u1, u2);
            af[h][0] = u32x4{x0, y0, z0, u0};
            af[h][1] = u32x4{x1, y1, z1, u1};
            af[h][2] = u32x4{x2, y2, z2, u2};
        }
        load_a(kt + 2 < last ? kt + 2 : last, cur);  // the registers are free again: two tiles ahead
        // Keep these loads HERE.  Nothing reads them before the next iteration, so hipcc's scheduler sinks them to the end of the
        // loop body — just in front of the s_waitcnt vmcnt(0) that opens the next step's split — and every k tile then waits a full
        // memory latency for its A operand: 3.0-3.5 k of a wave's 6.6 k cycles per k tile (round-3 s_memtime stamps; the tool was retired with the diagnostic build: git show 0304bc1:tools/split_stamps.py).
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ws = smem_split + buf * W_TILE;
        if constexpr (DEEP) {
            // Alone on its SIMD a wave issues the six products of one accumulator one MFMA latency apart (~36 instead of 16 cycles:
            // 3.5 k of a k tile's 3.7 k cycles).  Two column tiles x RG row groups are therefore kept in flight: four accumulators
            // take turns, each one's own order of products unchanged (same bits).
            constexpr int UT = 4 / RG;  // column tiles taken together
            bf16x8 bq[2 * UT][3];
#pragma unroll
            for (int u = 0; u < UT; ++u) read_b(ws, u, bq[u]);
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 8; t += UT) {
                const int cur = (t / UT) & 1;
                if (t + UT < 8) {
#pragma unroll
                    for (int u = 0; u < UT; ++u) read_b(ws, t + UT + u, bq[UT * (cur ^ 1) + u]);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int u = 0; u < UT; ++u)
#pragma unroll
                        for (int h = 0; h < RG; ++h)
                            acc[h][t + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[h][PA[q]]),
                                                                                   bq[UT * cur + u][PB[q]], acc[h][t + u], 0, 0, 0);
                if (t + UT == 4) store_w(buf ^ 1, Next{});
            }
        } else {
        bf16x8 bq[2][3];
        read_b(ws, 0, bq[0]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 1 < 8) read_b(ws, t + 1, bq[(t + 1) & 1]);
            const bf16x8 b0 = bq[t & 1][0], b1 = bq[t & 1][1], b2 = bq[t & 1][2];
#pragma unroll
            for (int h = 0; h < RG; ++h) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[h][0]);
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[h][1]);
                const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[h][2]);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[h][t], 0, 0, 0);
            }
            if (t == 3) store_w(buf ^ 1, Next{});
        }
        }
        __syncthreads();
    };
    for (int kt = 0; kt < n_tiles; kt += 2) {
        step(kt, a_pre[0], std::integral_constant<int, 0>{});
        if (kt + 1 < n_tiles) step(kt + 1, a_pre[1], std::integral_constant<int, 1>{});
    }
    // Round 6: the epilogue of gemm_split_kernel_w256 here too — every strip of 16 rows x 64 columns through 4 KB of this wave's LDS (the W
    // buffers are free behind the loop's last barrier: 12 KB per wave) and back row-major, 16-B stores instead of 4-B ones (a quarter of the
    // store instructions; per element the same operations in the same order: the same bits).  A timing-only build without any epilogue ran
    // the C = 512 ConvUnit 19 % faster: the store instructions were the largest single cost left in this kernel.  Where the 16-B accesses
    // are not possible (GEGLU's interleaved columns, an output or parameter row that is not 16-B aligned) the permlane form stays.
    {
        auto al16 = [](const void* q) __attribute__((always_inline)) { return ((uintptr_t)q & 15) == 0; };
        const bool rows_ok = p.epi != EPI_GEGLU && p.n % 4 == 0 && p.ldc % 4 == 0 && al16(p.c) && al16(p.bias) &&
                             (p.epi != EPI_BIAS_RES || (p.ldres % 4 == 0 && al16(p.res))) &&
                             ((p.epi != EPI_SNAKE && p.epi != EPI_SNAKE_GRN) || (al16(p.alpha) && al16(p.inv_alpha))) &&
                             (p.epi != EPI_SNAKE_GRN || (al16(p.gamma) && al16(p.beta)));
        if (rows_ok) {  // (wave-uniform: kernel arguments only)
            // (the lane number made opaque HERE: everything the epilogue derives from it — LDS offsets, row and column numbers — is then
            // computed behind the loop; hoisted above it those values cost the 168-register loop 29 spilled registers: 1.56 -> 2.27 ms)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            epilogue_rows<RG, 2>(p, [&](int sq, int h, int tt) __attribute__((always_inline)) -> f32x4a& { return acc[h][4 * sq + tt]; }, m0, n0, wave, lane_e,
                                 reinterpret_cast<float*>(smem_split + 12288 * wave));
            return;
        }
    }
    gemm_epilogue16<RG>(p, acc, m0, n0, wave, lane);
}

// ---- a single clip's products (a streaming chunk: 60-180 rows): column SLICES ------------------------------------------------------
// gemm_split_kernel_few_blocks gives 180 x 512 x 2048 (the second 1x1 conv of a C = 512 ConvUnit) twelve blocks, each streaming its
// 128-column block of W — 1.5 MB — through one CU at the ~30 GB/s a lone workgroup draws: 66 us, and every k tile of 12 MFMAs per wave
// waits on it.  Here a block owns 64 rows x 32 COLUMNS (column tiles 2 c, 2 c + 1 of the 128-column block): four times the blocks,
// each copying only its 32 rows of every plane of the W tile (3 x 2 KB, into the same LDS layout), with W and A requested FOUR k
// tiles ahead through four register sets — the loop is pure latency.  Same k order, same six plane products per accumulator in the
// same order as every other form of this GEMM: same bits (the batch-invariance tests compare them).
template <bool KTAIL>
__global__ __launch_bounds__(THREADS, 2) void gemm_split_kernel_slices(const GemmArgs p) {
    constexpr int AHEAD = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int n_slices = (p.n + 31) / 32;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)n_slices) * 64;
    const int slice = (int)(blockIdx.x % (unsigned)n_slices);
    const int n0 = 32 * slice, c = slice & 3;           // c: which 32 rows of the 128-row W tile
    const int n_tiles = (p.k + BK - 1) / BK;
    const int last = n_tiles - 1;
    const int64_t row = m0 + 16 * wave + ln;
    const float* const a_row = p.a + (row < p.m ? row : 0) * p.lda + 8 * lg;   // rows past the edge: row 0, never stored
    // this thread's 16-B chunks of the slice's rows: chunk q = tid + 256 j < 384 -> plane q / 128, byte (q % 128) * 16 of the 2 KB
    const unsigned char* const w_tile0 = p.w_img + (int64_t)(n0 / BN) * n_tiles * W_TILE + c * (32 * 64);
    const int q1 = tid + 256;
    const bool has1 = q1 < 384;
    const int off0 = (tid / 128) * W_PLANE + (tid % 128) * 16, off1 = (q1 / 128) * W_PLANE + (q1 % 128) * 16;
    float4 a_pre[AHEAD][2];
    u32x4 w_reg[AHEAD][2];
    auto load_a = [&](int kt, float4 (&dst)[2]) __attribute__((always_inline)) {
        int o = kt * BK;
        if (KTAIL) {
            const int kmax = p.k - 8 - 8 * lg;
            o = o < kmax ? o : kmax;
        }
        dst[0] = *reinterpret_cast<const float4*>(a_row + o);
        dst[1] = *reinterpret_cast<const float4*>(a_row + o + 4);
    };
    auto load_w = [&](int kt, u32x4 (&dst)[2]) __attribute__((always_inline)) {
        const unsigned char* t = w_tile0 + (int64_t)kt * W_TILE;
        dst[0] = *reinterpret_cast<const u32x4*>(t + off0);
        dst[1] = *reinterpret_cast<const u32x4*>(t + (has1 ? off1 : off0));
    };
    auto store_w = [&](int buf, const u32x4 (&src)[2]) __attribute__((always_inline)) {
        unsigned char* base = smem_split + buf * W_TILE + c * (32 * 64);
        *reinterpret_cast<u32x4*>(base + off0) = src[0];
        if (has1) *reinterpret_cast<u32x4*>(base + off1) = src[1];
    };
    auto read_b = [&](const unsigned char* ws, int t, bf16x8 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) b[pl] = *reinterpret_cast<const bf16x8*>(ws + pl * W_PLANE + tile_off(16 * t + ln, lg));
    };
    f32x4a acc[1][2] = {{f32x4a{0.f, 0.f, 0.f, 0.f}, f32x4a{0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        load_a(j < last ? j : last, a_pre[j]);
        load_w(j < last ? j : last, w_reg[j]);
    }
    store_w(0, w_reg[0]);
    load_w(AHEAD < last ? AHEAD : last, w_reg[0]);
    __syncthreads();
    // step kt: LDS buffer kt & 1 holds W tile kt; register set (kt + 1) % AHEAD holds tile kt + 1 (stored here into the other buffer
    // and refilled with tile kt + 1 + AHEAD); A set kt % AHEAD holds tile kt (split here, refilled with tile kt + AHEAD)
    auto step = [&](int kt, auto set_) __attribute__((always_inline)) {
        constexpr int set = decltype(set_)::value, nxt = (set + 1) % AHEAD;
        const int buf = kt & 1;
        u32x4 af[3];
        {
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(a_pre[set][0].x, a_pre[set][0].y, x0, x1, x2);
            split2(a_pre[set][0].z, a_pre[set][0].w, y0, y1, y2);
            split2(a_pre[set][1].x, a_pre[set][1].y, z0, z1, z2);
            split2(a_pre[set][1].z, a_pre[set][1].w, u0, u1, u2);
            af[0] = u32x4{x0, y0, z0, u0};
            af[1] = u32x4{x1, y1, z1, u1};
            af[2] = u32x4{x2, y2, z2, u2};
        }
        load_a(kt + AHEAD < last ? kt + AHEAD : last, a_pre[set]);
        store_w(buf ^ 1, w_reg[nxt]);                                        // tile kt + 1
        load_w(kt + 1 + AHEAD < last ? kt + 1 + AHEAD : last, w_reg[nxt]);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ws = smem_split + buf * W_TILE;
        bf16x8 bq[2][3];
        read_b(ws, 2 * c, bq[0]);
        read_b(ws, 2 * c + 1, bq[1]);
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[PA[q]]), bq[u][PB[q]], acc[0][u], 0, 0, 0);
        __syncthreads();
    };
    for (int kt = 0; kt < n_tiles; kt += AHEAD) {
        step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < n_tiles) step(kt + 1, std::integral_constant<int, 1>{});
        if (kt + 2 < n_tiles) step(kt + 2, std::integral_constant<int, 2>{});
        if (kt + 3 < n_tiles) step(kt + 3, std::integral_constant<int, 3>{});
    }
    gemm_epilogue16<1, 2>(p, acc, m0, n0, wave, lane);
}

// ---- a single clip's products, STREAMED (round 5): every global read of the loop is an LDS-DMA copy, 8 k tiles in flight ---------------
// The slices form keeps four k tiles in flight in registers and pays ~0.47 us per k tile whatever it multiplies (36 us for 180 x 512 x
// 2048): one memory latency per four tiles.  Here a workgroup owns 32 rows x 64 columns — wave (rs, cg) = row strip rs of 16, column
// group cg of 32: the slices form's accumulators — and a k tile is one 16-KB ring slot, copied by global_load_lds_dwordx4 with nothing in
// registers: wave 0 the A rows (fp32, in FRAGMENT order: lane (row m, k group g) of block (rs, j) fetches the 16 B it will read back,
// k = 8 g + 4 j .. + 3 of row 16 rs + m), waves 1-3 one plane each of the workgroup's 64 rows of the W tile (verbatim: the image's XOR
// swizzle only depends on row / 4 mod 4).  PF = 7 or 8 slots stay in flight behind counted waits (four copies per wave and slot), the
// fragments of tile kt + 1 are read while tile kt multiplies.  Same k order, same six plane products per accumulator in the same order
// as every other form of this GEMM: same bits.
__device__ __forceinline__ void dma_1k(const unsigned char* base, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(base), "s"(lds_dst)
        : "memory");
}
constexpr int ST_SLOT = 16384, ST_RING = 9;  // A 4 KB | W plane 0 | plane 1 | plane 2 (4 KB each: 64 rows x 64 B)
// CONV (three taps): the A rows of k tile kt are those of frame t + (tap - 1) dil, tap = kt / (cin / 32).  A copy cannot write zeros, so
// the copying lane fetches a CLAMPED row (any valid address) and the multiplying wave zeroes the fragment of a frame whose tap falls outside
// its clip before the split — the values gemm_split_conv_kernel's masked loads produce: same bits.
template <int PF, bool CONV = false>
__global__ __launch_bounds__(THREADS, 1) void gemm_split_kernel_stream(const GemmArgs p) {
    static_assert(PF + 1 <= ST_RING && 4 * (PF - 1) <= 63, "ring / counted wait");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int rs = wave & 1, cg = wave >> 1;
    const int n_blocks = (p.n + 63) / 64;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)n_blocks) * 32;
    const int nb = (int)(blockIdx.x % (unsigned)n_blocks);
    const int n_tiles = p.k / BK;  // (whole k tiles, at least PF of them: the launcher)
    // this wave's four copies of a slot: global base (wave-uniform), two lane offsets, LDS offset inside the slot
    const unsigned char* src;
    int64_t src_step;          // per k tile
    unsigned lo[2];            // lane offsets of blocks 0-1 and 2-3
    unsigned lt0[2] = {0u, 0u}, lt1[2] = {0u, 0u}, lt2[2] = {0u, 0u};  // CONV, wave 0: the same for the rows of tap 0 / 1 / 2
    int blk_step;              // global bytes from a block to the next of the same pair
    const int tiles_per_tap = CONV ? p.cin / BK : 1;
    if (wave == 0) {
        src = reinterpret_cast<const unsigned char*>(p.a);
        src_step = BK * 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t row = m0 + 16 * r + ln;
            const int64_t rr = row < p.m ? row : 0;  // rows past the edge: row 0, never stored
            lo[r] = (unsigned)((rr * p.lda + 8 * lg) * 4);
            if constexpr (CONV) {
                // (a row outside the clip is zeroed by the consumer; outside the tensor: clamped)
                auto off = [&](int64_t rt) __attribute__((always_inline)) {
                    rt = rt < 0 ? 0 : (rt >= p.m ? p.m - 1 : rt);
                    return (unsigned)((rt * p.lda + 8 * lg) * 4);
                };
                lt0[r] = off(rr - p.dil);
                lt1[r] = off(rr);
                lt2[r] = off(rr + p.dil);
            }
        }
        blk_step = 16;  // j = 0, 1: the two 16-B halves of the lane's 32 B
    } else {
        src = p.w_img + (int64_t)(nb / 2) * n_tiles * W_TILE + (wave - 1) * W_PLANE + (nb & 1) * 4096;
        src_step = W_TILE;
        lo[0] = 16u * (unsigned)lane;
        lo[1] = 16u * (unsigned)lane + 2048u;
        blk_step = 1024;
    }
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_split;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        const unsigned char* g = src + (int64_t)kt * src_step;
        unsigned l0 = lo[0], l1 = lo[1];
        if constexpr (CONV) {
            if (wave == 0) {  // (wave-uniform) channel block kt % tiles_per_tap of the rows of tap kt / tiles_per_tap
                const int tap = kt / tiles_per_tap;
                g = src + (int64_t)(kt - tap * tiles_per_tap) * src_step;
                // (sums of masked differences: as a chain of selects hipcc built a table of the three offsets in scratch memory and indexed it)
                l0 = lt1[0] + (tap == 0 ? lt0[0] - lt1[0] : 0u) + (tap == 2 ? lt2[0] - lt1[0] : 0u);
                l1 = lt1[1] + (tap == 0 ? lt0[1] - lt1[1] : 0u) + (tap == 2 ? lt2[1] - lt1[1] : 0u);
            }
        }
        // (wave-uniform by construction; said so, or a k tile index that also feeds vector arithmetic — CONV's tap test — lands these in VGPRs)
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + (unsigned)((kt % ST_RING) * ST_SLOT + 4096 * wave)));
        auto uni = [](const unsigned char* q) __attribute__((always_inline)) {
            const uint64_t a = (uint64_t)q;
            return reinterpret_cast<const unsigned char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                                          (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
        };
        const unsigned char* const g0 = uni(g);
        const unsigned char* const g1 = uni(g + blk_step);
        dma_1k(g0, l0, dst);
        dma_1k(g1, l0, dst + 1024);
        dma_1k(g0, l1, dst + 2048);
        dma_1k(g1, l1, dst + 3072);
    };
    // CONV: which taps of this lane's frame (the row strip this wave multiplies) fall outside its clip: bit tap
    unsigned oob_taps = 0u;
    if constexpr (CONV) {
        const int64_t row = m0 + 16 * rs + ln;
        const int t = (int)((unsigned)(row < p.m ? row : 0) % (unsigned)p.frames);
        oob_taps = (t - p.dil < 0 ? 1u : 0u) | (t + p.dil >= (int)p.frames ? 4u : 0u);
    }
    float4 fa[2][2];
    bf16x8 fw[2][2][3];
    auto fetch = [&](int kt, auto par_) __attribute__((always_inline)) {
        constexpr int P = decltype(par_)::value;
        const unsigned char* slot = smem_split + (kt % ST_RING) * ST_SLOT;
        fa[P][0] = *reinterpret_cast<const float4*>(slot + rs * 2048 + 16 * lane);
        fa[P][1] = *reinterpret_cast<const float4*>(slot + rs * 2048 + 1024 + 16 * lane);
        if constexpr (CONV) {  // the conv's zero padding: the rows of a tap outside the clip count as zeros
            const bool oob = ((oob_taps >> (unsigned)(kt / tiles_per_tap)) & 1u) != 0u;
            fa[P][0] = oob ? make_float4(0.f, 0.f, 0.f, 0.f) : fa[P][0];
            fa[P][1] = oob ? make_float4(0.f, 0.f, 0.f, 0.f) : fa[P][1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
                fw[P][u][pl] = *reinterpret_cast<const bf16x8*>(slot + 4096 + pl * 4096 + tile_off(32 * cg + 16 * u + ln, lg));
    };
    f32x4a acc[1][2] = {{f32x4a{0.f, 0.f, 0.f, 0.f}, f32x4a{0.f, 0.f, 0.f, 0.f}}};
    auto multiply = [&](auto par_) __attribute__((always_inline)) {
        constexpr int P = decltype(par_)::value;
        unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
        split2(fa[P][0].x, fa[P][0].y, x0, x1, x2);
        split2(fa[P][0].z, fa[P][0].w, y0, y1, y2);
        split2(fa[P][1].x, fa[P][1].y, z0, z1, z2);
        split2(fa[P][1].z, fa[P][1].w, u0, u1, u2);
        const u32x4 af[3] = {u32x4{x0, y0, z0, u0}, u32x4{x1, y1, z1, u1}, u32x4{x2, y2, z2, u2}};
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[PA[q]]), fw[P][u][PB[q]], acc[0][u], 0, 0, 0);
    };
    // this wave's copies of slot kt have landed when at most 4 x `after` newer ones are outstanding; then everybody's have, and
    // everybody's reads of the slot before it are complete (its ring position is the next one to be overwritten)
    auto landed = [&](auto after_) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * decltype(after_)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int j = 0; j < PF; ++j) issue(j);
    landed(std::integral_constant<int, PF - 1>{});
    fetch(0, P0{});
    const int main_steps = n_tiles - PF;  // even (the launcher picks PF)
    for (int kt = 0; kt < main_steps; kt += 2) {
        issue(kt + PF);
        landed(std::integral_constant<int, PF - 1>{});
        fetch(kt + 1, P1{});
        multiply(P0{});
        issue(kt + 1 + PF);
        landed(std::integral_constant<int, PF - 1>{});
        fetch(kt + 2, P0{});
        multiply(P1{});
    }
    // the last PF tiles: nothing left to request, the counted waits shrink
    tail_for<PF>([&](auto j_) __attribute__((always_inline)) {
        constexpr int J = decltype(j_)::value;
        if constexpr (J + 1 < PF) {
            landed(std::integral_constant<int, PF - 2 - J>{});
            fetch(main_steps + J + 1, std::integral_constant<int, (J + 1) & 1>{});
        }
        multiply(std::integral_constant<int, J & 1>{});
    });
    gemm_epilogue16<1, 2>(p, acc, m0 + 16 * rs, 64 * nb + 32 * cg, 0, lane);
}

// The NARROW streamed form: 32 rows x 16 columns per workgroup, for products whose 32 x 64 grid would leave most of the chip idle while
// each workgroup pulls a megabyte through its CU (180 x 512 x 2048: 48 workgroups, 23 us at the ~45 GB/s a CU's LDS-DMA copies land at;
// here 192 workgroups of 448 KB).  A ring slot is FOUR k tiles (28 KB: per k tile the A rows as above and the 16 rows of each W plane,
// 1 KB each); wave v copies k tile v of every slot (seven copies), waves 0 and 1 multiply (row strip = wave, one accumulator tile, the
// six plane products of a k tile in the usual order, k tiles in order: same bits), one barrier per four k tiles.  Bias / residual epilogues.
constexpr int SN_KT = 7168, SN_SLOT = 4 * SN_KT, SN_RING = 5, SN_PF = SN_RING - 1;
__global__ __launch_bounds__(THREADS, 1) void gemm_split_kernel_stream_narrow(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int n_blocks = (p.n + 15) / 16;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)n_blocks) * 32;
    const int nb = (int)(blockIdx.x % (unsigned)n_blocks);
    const int n_tiles = p.k / BK, n_slots = n_tiles / 4;  // (whole slots, at least SN_PF of them: the launcher)
    unsigned lo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t row = m0 + 16 * r + ln;
        lo[r] = (unsigned)(((row < p.m ? row : 0) * p.lda + 8 * lg) * 4);  // rows past the edge: row 0, never stored
    }
    const unsigned char* const a_src = reinterpret_cast<const unsigned char*>(p.a) + (int64_t)wave * (BK * 4);
    const unsigned char* const w_src = p.w_img + ((int64_t)(nb / 8) * n_tiles + wave) * W_TILE + (nb & 7) * 1024;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_split;
    auto issue = [&](int q) __attribute__((always_inline)) {  // this wave's k tile 4 q + wave of slot q
        const unsigned char* ga = a_src + (int64_t)q * (4 * BK * 4);
        const unsigned char* gw = w_src + (int64_t)q * (4 * W_TILE);
        const unsigned dst = ring_lds + (unsigned)((q % SN_RING) * SN_SLOT + SN_KT * wave);
        dma_1k(ga, lo[0], dst);
        dma_1k(ga + 16, lo[0], dst + 1024);
        dma_1k(ga, lo[1], dst + 2048);
        dma_1k(ga + 16, lo[1], dst + 3072);
        dma_1k(gw, 16u * (unsigned)lane, dst + 4096);
        dma_1k(gw + W_PLANE, 16u * (unsigned)lane, dst + 5120);
        dma_1k(gw + 2 * W_PLANE, 16u * (unsigned)lane, dst + 6144);
    };
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](int q) __attribute__((always_inline)) {
        if (wave >= 2) return;
        const unsigned char* slot = smem_split + (q % SN_RING) * SN_SLOT;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned char* t = slot + kk * SN_KT;
            const float4 f0 = *reinterpret_cast<const float4*>(t + wave * 2048 + 16 * lane);
            const float4 f1 = *reinterpret_cast<const float4*>(t + wave * 2048 + 1024 + 16 * lane);
            bf16x8 fw[3];
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) fw[pl] = *reinterpret_cast<const bf16x8*>(t + 4096 + pl * 1024 + tile_off(ln, lg));
            unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, u0, u1, u2;
            split2(f0.x, f0.y, x0, x1, x2);
            split2(f0.z, f0.w, y0, y1, y2);
            split2(f1.x, f1.y, z0, z1, z2);
            split2(f1.z, f1.w, u0, u1, u2);
            const u32x4 af[3] = {u32x4{x0, y0, z0, u0}, u32x4{x1, y1, z1, u1}, u32x4{x2, y2, z2, u2}};
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int m = 0; m < 6; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[PA[m]]), fw[PB[m]], acc, 0, 0, 0);
        }
    };
    // slot q has landed for this wave when at most 7 x `after` newer copies are outstanding; then for everybody — and everybody's
    // reads of slot q - 1, whose ring position the next request overwrites, are complete
    auto landed = [&](auto after_) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(7 * decltype(after_)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int j = 0; j < SN_PF; ++j) issue(j);
    const int main_steps = n_slots - SN_PF;
    for (int q = 0; q < main_steps; ++q) {
        landed(std::integral_constant<int, SN_PF - 1>{});
        issue(q + SN_PF);
        multiply(q);
    }
    tail_for<SN_PF>([&](auto j_) __attribute__((always_inline)) {
        constexpr int J = decltype(j_)::value;
        landed(std::integral_constant<int, SN_PF - 1 - J>{});
        multiply(main_steps + J);
    });
    if (wave >= 2) return;
    const int col = 16 * nb + ln;
    if (col >= p.n) return;
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + 16 * wave + 4 * lg + i;
        if (m >= p.m) continue;
        float v = acc[i] + bias;
        if (p.epi == EPI_BIAS_RES) v = p.res[m * p.ldres + col] + v;
        c_store(p.c + m * p.ldc + col, v);
    }
}

// (The body is written for RG row groups of 16 per wave; RG = 4, i.e. 256-row blocks at two per CU, halves the LDS reads
// and the W traffic per MFMA and was measured: +3 % on the K = 2048 shapes, -17 % on the K = 512 ones, whose epilogue it doubles.)
template <bool KTAIL>
__global__ __launch_bounds__(THREADS, 3) void gemm_split_kernel(const GemmArgs p, const int gp) {
    gemm_split_body16<KTAIL, 2, false>(p, gp);
}
// the implicit-convolution A operand (a kernel of its own in the profiles: its A path differs)
__global__ __launch_bounds__(THREADS, 3) void gemm_split_conv_kernel(const GemmArgs p, const int gp) {
    gemm_split_body16<false, 2, false, true>(p, gp);
}
__global__ __launch_bounds__(THREADS, 2) void gemm_split_conv_kernel_few_blocks(const GemmArgs p, const int gp) {
    gemm_split_body16<false, 1, true, true>(p, gp);
}
// the grid leaves CUs idle (a single clip, a streaming chunk): W requested two k tiles ahead
template <bool KTAIL>
__global__ __launch_bounds__(THREADS, 2) void gemm_split_kernel_few_blocks(const GemmArgs p, const int gp) {
    gemm_split_body16<KTAIL, 1, true>(p, gp);  // 64-row blocks: twice the blocks, half the MFMAs per k tile and block
}

}  // namespace

// the route a new context starts on (l3ac_ctx::gemm_split; l3ac_ctx_set_gemm_split changes it per context)
bool gemm_split_default() {
    const char* e = std::getenv("L3AC_GEMM_SPLIT");
    return e ? (std::atoi(e) != 0) : true;
}

// n < 192 would be a single 128-column block per row panel: too few workgroups at the transformer's row counts, where
// the exact kernel's narrower tiles win (measured: 128 x 344 and 128 x 192 weights, 26 vs 45 TFLOP/s at 15360 rows)
bool gemm_split_eligible(int n, int k) { return n >= 192 && k >= 32 && k % 8 == 0; }
// implicit-conv A operand (taps > 1): whole k tiles inside a tap, 32-bit row arithmetic
bool gemm_split_conv_ok(const GemmArgs& g) {
    return g.cin > 0 && g.cin % BK == 0 && g.k == g.taps * g.cin && g.frames > 0 && g.m % g.frames == 0 && g.m < ((int64_t)1 << 31) && g.lda == g.cin;
}

// L3AC_GEMM_W256 (A/B measurements — same bits either way): 0 the batch products of 256-column weights stay on gemm_split_kernel |
// 1 (default) the light-epilogue long-K products on gemm_split_kernel_w256 | 2 every eligible shape there
static int w256_enabled() {
    static const int on = [] {
        const char* e = std::getenv("L3AC_GEMM_W256");
        return e ? std::atoi(e) : 1;
    }();
    return on;
}

int64_t gemm_split_image_bytes(int n, int k) { return (int64_t)((n + BN - 1) / BN) * ((k + BK - 1) / BK) * W_TILE; }

void gemm_split_image_host(const float* w, int64_t ldw, int n, int k, unsigned char* img) {
    const int k_tiles = (k + BK - 1) / BK;
    const int n_pad = (n + BN - 1) / BN * BN;
    for (int row = 0; row < n_pad; ++row)
        for (int kc = 0; kc < 4 * k_tiles; ++kc) {
            uint16_t pl[3][8];
            for (int j = 0; j < 8; ++j) {
                const int kk = 8 * kc + j;
                const float x = (row < n && kk < k) ? w[(int64_t)row * ldw + kk] : 0.f;
                uint16_t h[3];
                split3_host(x, h);
                pl[0][j] = h[0];
                pl[1][j] = h[1];
                pl[2][j] = h[2];
            }
            unsigned char* tile = img + ((int64_t)(row / BN) * k_tiles + kc / 4) * W_TILE;
            for (int p = 0; p < 3; ++p) std::memcpy(tile + p * W_PLANE + tile_off(row % BN, kc % 4), pl[p], 16);
        }
}

int launch_gemm_split_image(hipStream_t s, const float* w, int64_t ldw, int n, int k, unsigned char* img) {
    L3AC_REQUIRE(w && img && n > 0 && k > 0, "split image: bad arguments");
    const int64_t chunks = gemm_split_image_bytes(n, k) / 48;
    hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)ceil_div64(chunks, 256)), dim3(256), 0, s, w, ldw, n, k, img);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_gemm_split(hipStream_t s, const GemmArgs& g) {
    L3AC_REQUIRE(g.a && g.w_img && g.c, "split gemm: null operand");
    const bool conv = g.taps > 1;
    L3AC_REQUIRE(gemm_split_eligible(g.n, g.k) && (!conv || gemm_split_conv_ok(g)), "split gemm: unsupported shape n=%d k=%d taps=%d cin=%d",
                 g.n, g.k, g.taps, g.cin);
    L3AC_REQUIRE(g.lda % 4 == 0 && ((uintptr_t)g.a & 15) == 0 && ((uintptr_t)g.w_img & 15) == 0, "split gemm: operands must be 16-byte aligned");
    if (g.epi == EPI_GEGLU) L3AC_REQUIRE(g.n % 64 == 0, "split gemm: GEGLU epilogue needs interleaved 64-column tiles");
    if (g.epi == EPI_BIAS_RES) L3AC_REQUIRE(g.res, "split gemm: residual epilogue without residual");
    if (g.epi == EPI_SNAKE || g.epi == EPI_SNAKE_GRN) L3AC_REQUIRE(g.alpha && g.inv_alpha, "split gemm: snake without alpha");
    if (g.epi == EPI_SNAKE_GRN) L3AC_REQUIRE(g.gamma && g.beta, "split gemm: GRN without gamma/beta");
    const int64_t blocks = ceil_div64(g.m, BM) * ceil_div64(g.n, BN);
    if (blocks <= 0) return L3AC_OK;
    L3AC_REQUIRE(blocks < (int64_t)1 << 31, "split gemm: grid too large (m=%lld n=%d)", (long long)g.m, g.n);
    const int cus = l3ac_device_cu_count();
    // a batch's rows, 256-column blocks, long K, a light epilogue (the C = 512 stage's second product): one wave per SIMD, 192 x 256 per
    // workgroup (gemm_split_w256.hip).  Measured inside the 256-clip step (profiles/r06/gemm_w256.md): 24480 x 512 x 2048 0.242 -> 0.221 ms;
    // the first product (K = 512, snake + GRN epilogue: four tiles per CU, each with an exposed 38 k-cycle epilogue) 0.246 -> 0.258 — and
    // 0.265 in a persistent form that runs a tile's epilogue inside the next tile's k loop (tools/patches/gemm_split_w256d.patch) —,
    // 46080 x 256 x 512 0.063 -> 0.066: those stay on gemm_split_kernel (L3AC_GEMM_W256=2 sends them here too: same bits)
    const int w256 = conv ? 0 : w256_enabled();  // 1: the light-epilogue long-K products | 2: every eligible shape
    const bool use_w256 = w256 && gemm_split_w256_ok(g) && blocks > cus && (w256 >= 2 || (g.k >= 1024 && (g.epi == EPI_BIAS || g.epi == EPI_BIAS_RES)));
    char name[64];
    std::snprintf(name, sizeof(name), "%s %lldx%dx%d e%d", conv ? "gemm_split_conv_kernel" : use_w256 ? "gemm_split_kernel_w256" : "gemm_split_kernel",
                  (long long)g.m, g.n, g.k, g.epi);
    const double c_cols = g.epi == EPI_GEGLU ? (double)g.ldc : (double)g.n;
    ProfScope prof(s, name, 2.0 * (double)g.m * g.n * g.k,
                   4.0 * ((double)g.m * g.k + (double)g.m * c_cols * (g.epi == EPI_BIAS_RES ? 2.0 : 1.0)) + 6.0 * (double)g.n * g.k);
    const int gp = 8;  // row panels per group of the XCD-aware tile order (a multiple of the 8 XCDs)
    const bool tail = g.k % BK != 0;
    if (conv && g.taps == 3 && g.k / BK >= 8 && g.lda * g.m < ((int64_t)1 << 29) && ceil_div64(g.m, BM / 2) * ceil_div64(g.n, BN) <= cus / 4) {
        // a single clip's k3 conv: the streamed form (32 x 64 blocks), its A rows by tap
        static PerDeviceOnce configured;
        if (configured.first()) {
            L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_stream<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_RING * ST_SLOT));
            L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_stream<7, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_RING * ST_SLOT));
            configured.done();
        }
        const unsigned grid = (unsigned)(ceil_div64(g.m, 32) * ceil_div64(g.n, 64));
        if ((g.k / BK) % 2 == 0)
            hipLaunchKernelGGL((gemm_split_kernel_stream<8, true>), dim3(grid), dim3(THREADS), ST_RING * ST_SLOT, s, g);
        else
            hipLaunchKernelGGL((gemm_split_kernel_stream<7, true>), dim3(grid), dim3(THREADS), ST_RING * ST_SLOT, s, g);
    } else if (conv) {  // (whole k tiles: no tail; the 16x16x32 form only)
        if (blocks <= cus)
            hipLaunchKernelGGL(gemm_split_conv_kernel_few_blocks, dim3((unsigned)(ceil_div64(g.m, BM / 2) * ceil_div64(g.n, BN))), dim3(THREADS),
                               2 * W_TILE, s, g, gp);
        else
            hipLaunchKernelGGL(gemm_split_conv_kernel, dim3((unsigned)blocks), dim3(THREADS), 2 * W_TILE, s, g, gp);
    } else if (g.epi != EPI_GEGLU && !tail && g.k / BK >= 8 && g.lda * g.m < ((int64_t)1 << 29) && ceil_div64(g.m, BM / 2) * ceil_div64(g.n, BN) <= cus / 4) {
        // a single clip, whole k tiles: the streamed form (32 rows x 64 columns per block); PF of the same parity as the k tiles
        static PerDeviceOnce configured;
        if (configured.first()) {
            L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_stream<8>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_RING * ST_SLOT));
            L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_stream<7>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_RING * ST_SLOT));
            L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_stream_narrow), hipFuncAttributeMaxDynamicSharedMemorySize, SN_RING * SN_SLOT));
            configured.done();
        }
        const unsigned grid = (unsigned)(ceil_div64(g.m, 32) * ceil_div64(g.n, 64));
        if ((g.epi == EPI_BIAS || g.epi == EPI_BIAS_RES) && g.k % (4 * BK) == 0 && g.k / (4 * BK) >= SN_PF && 2 * grid <= (unsigned)cus)
            hipLaunchKernelGGL(gemm_split_kernel_stream_narrow, dim3((unsigned)(ceil_div64(g.m, 32) * ceil_div64(g.n, 16))), dim3(THREADS), SN_RING * SN_SLOT, s, g);
        else if ((g.k / BK) % 2 == 0)
            hipLaunchKernelGGL((gemm_split_kernel_stream<8>), dim3(grid), dim3(THREADS), ST_RING * ST_SLOT, s, g);
        else
            hipLaunchKernelGGL((gemm_split_kernel_stream<7>), dim3(grid), dim3(THREADS), ST_RING * ST_SLOT, s, g);
    } else if (g.epi != EPI_GEGLU && ceil_div64(g.m, BM / 2) * ceil_div64(g.n, BN) <= cus / 4) {
        // a single clip: column slices (64 rows x 32 columns per block)
        const unsigned grid = (unsigned)(ceil_div64(g.m, 64) * ceil_div64(g.n, 32));
        if (tail)
            hipLaunchKernelGGL((gemm_split_kernel_slices<true>), dim3(grid), dim3(THREADS), 2 * W_TILE, s, g);
        else
            hipLaunchKernelGGL((gemm_split_kernel_slices<false>), dim3(grid), dim3(THREADS), 2 * W_TILE, s, g);
    } else if (use_w256) {
        L3AC_TRY(launch_gemm_split_w256(s, g));
    } else if (blocks <= cus) {
        const unsigned few = (unsigned)(ceil_div64(g.m, BM / 2) * ceil_div64(g.n, BN));
        if (tail)
            hipLaunchKernelGGL((gemm_split_kernel_few_blocks<true>), dim3(few), dim3(THREADS), 2 * W_TILE, s, g, gp);
        else
            hipLaunchKernelGGL((gemm_split_kernel_few_blocks<false>), dim3(few), dim3(THREADS), 2 * W_TILE, s, g, gp);
    } else {
        if (tail)
            hipLaunchKernelGGL((gemm_split_kernel<true>), dim3((unsigned)blocks), dim3(THREADS), 2 * W_TILE, s, g, gp);
        else
            hipLaunchKernelGGL((gemm_split_kernel<false>), dim3((unsigned)blocks), dim3(THREADS), 2 * W_TILE, s, g, gp);
    }
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
