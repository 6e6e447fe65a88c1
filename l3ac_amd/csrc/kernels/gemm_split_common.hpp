// Pieces shared by the bf16x3 GEMM kernels (gemm_split.hip, gemm_split_w256.hip): tile geometry of the pre-split weight image, the
// accumulator-layout epilogue of the 16x16x32 kernels, a compile-time loop.
#pragma once

#include <type_traits>

#include "../kernels.hpp"
#include "device_math.hpp"
#include "gemm_epilogue.hpp"
#include "split_bf16.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, THREADS = 256;
constexpr int W_PLANE = BN * 64;      // bytes of one bf16 plane of a W tile (64-B rows)
constexpr int W_TILE = 3 * W_PLANE;   // = L3AC_SPLIT_TILE_BYTES
constexpr int W_LOADS = W_TILE / (16 * THREADS);
static_assert(W_TILE == L3AC_SPLIT_TILE_BYTES, "image geometry");

// byte offset of 16-B chunk `chunk` (8 k values) of row `row` in a [128][32] bf16 plane: chunk ^ sigma(row / 4) with sigma(q) = -q mod 4.
// A ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 — and a group
// is conflict-free when its lanes hit 16 distinct 16-B slots of the 256-B bank row, i.e. distinct values of 4 (row & 3) + position.
//   16x16x32 fragments: lane (n = lane & 15, kg = lane >> 4) reads chunk kg of row 16 t + n, so a group holds the (row / 4, kg) pairs
//     {(0,0), (3,0), (1,1), (2,1)} (and the like): kg ^ {0, 3, 2, 1}[row / 4] separates them; sigma(q) = q (round 1) does not.
//   32x32x16 fragments: lane (row = lane & 31, h) reads chunk 2 s + h: a group holds one chunk of rows with four different row / 4:
//     any bijective sigma is conflict-free.
__host__ __device__ __forceinline__ int tile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }

typedef float f32x4a __attribute__((ext_vector_type(4)));
// The bf16x3 GEMMs' output stores are non-temporal (bias / residual / snake epilogues): the streamed-out C tile does not
// push the A panel and the W tiles, which the column blocks of the same XCD re-read, out of L2.  Measured (profiles/r04/wide_nt.md):
// the step's 14 launches 3.05-3.07 -> 2.99 ms in both of two interleaved rounds (the consumer of the hidden tensor included).
__device__ __forceinline__ void c_store(float* p, float v) { __builtin_nontemporal_store(v, p); }

// Epilogue of the 16x16 accumulator layout.  Stored as they are, the tiles give 64-B row segments (16 lanes x 4 B) — measured: 29-42 k
// cycles per block, a quarter to a half of a block's life, ten times the 32x32 layout's epilogue.  v_permlane16_swap_b32 exchanges the
// odd 16-lane rows of one register with the even rows of another: applied to element i of the column tiles 2 u and 2 u + 1 it leaves
//   first  result: lanes 0-31 = row 16 h + i,     columns 32 u .. 32 u + 31; lanes 32-63 = row 16 h + 8 + i,  same columns
//   second result: lanes 0-31 = row 16 h + 4 + i, columns 32 u .. 32 u + 31; lanes 32-63 = row 16 h + 12 + i, same columns
// i.e. whole 128-B lines per half wave, and the lane's column (hence its bias / snake / GRN parameters) is 32 u + (lane & 31) for both.
// NT column tiles of 16 (8: the whole 128-column block; 2: one 32-column slice, n0 = its first column — not with EPI_GEGLU)
template <int RG, int NT = 8>
__device__ __forceinline__ void gemm_epilogue16(const GemmArgs& p, f32x4a (&acc)[RG][NT], int64_t m0, int n0, int wave, int lane) {
    const int c32 = lane & 31, hi = lane >> 5;
    const int64_t mw = m0 + 16 * RG * wave + 8 * hi;  // + 16 h + i (first result), + 16 h + 4 + i (second)
    // (inline asm: with __builtin_amdgcn_permlane16_swap hipcc 7.2 folded the four swaps of a tile pair into one — every row of a
    // group of eight came out as its first; the s_nop covers the VALU-write -> permlane read hazard of operands written just before)
    auto swapped = [&](int h, int u, int i, float& lo, float& up) __attribute__((always_inline)) {
        float a = acc[h][2 * u][i], b = acc[h][2 * u + 1][i];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
        lo = a;
        up = b;
    };
    if constexpr (NT == 8) if (p.epi == EPI_GEGLU) {
        // 32-column tiles come in (value, gate) pairs u = 0, 2 with u + 1; output column j = n0 / 2 + 16 u + c32
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const int j = (n0 >> 1) + 16 * u + c32;
#pragma unroll
            for (int h = 0; h < RG; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v0, v1, g0, g1;
                    swapped(h, u, i, v0, v1);
                    swapped(h, u + 1, i, g0, g1);
                    const int64_t m = mw + 16 * h + i;
                    if (j < (int)p.ldc) {
                        if (m < p.m) p.c[m * p.ldc + j] = v0 * gelu_erf(g0);
                        if (m + 4 < p.m) p.c[(m + 4) * p.ldc + j] = v1 * gelu_erf(g1);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        const int n = n0 + 32 * u + c32;
        const bool n_ok = n < p.n;
        const int nc = n_ok ? n : 0;
        const float bias = p.bias ? p.bias[nc] : 0.f;
        float alpha = 0.f, inv_alpha = 0.f, gamma = 0.f, beta = 0.f;
        if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {
            alpha = p.alpha[nc];
            inv_alpha = p.inv_alpha[nc];
        }
        if (p.epi == EPI_SNAKE_GRN) {
            gamma = p.gamma[nc];
            beta = p.beta[nc];
        }
#pragma unroll
        for (int h = 0; h < RG; ++h) {
            float lo[4], up[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) swapped(h, u, i, lo[i], up[i]);  // (every lane takes part in the swap: no early exit above)
            if (!n_ok) continue;
            if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {  // packed fp32 math on register pairs (two rows, same column)
                const f32x2 al = (f32x2)(alpha), ia = (f32x2)(inv_alpha), ga = (f32x2)(gamma), be = (f32x2)(beta), bi = (f32x2)(bias);
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        f32x2 hv;
                        hv.x = half ? up[i] : lo[i];
                        hv.y = half ? up[i + 1] : lo[i + 1];
                        const f32x2 sv = snake_act2(hv + bi, al, ia);
                        const f32x2 o = p.epi == EPI_SNAKE_GRN ? __builtin_elementwise_fma(ga, sv, be) + sv : sv;  // layers.py:115, n_x == 1
                        const int64_t m = mw + 16 * h + 4 * half + i;
                        if (m < p.m) c_store(p.c + m * p.ldc + n, o.x);
                        if (m + 1 < p.m) c_store(p.c + (m + 1) * p.ldc + n, o.y);
                    }
                continue;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = mw + 16 * h + 4 * half + i;
                    if (m >= p.m) continue;
                    float v = (half ? up[i] : lo[i]) + bias;
                    if (p.epi == EPI_BIAS_RES) v = p.res[m * p.ldres + n] + v;
                    c_store(p.c + m * p.ldc + n, v);
                }
        }
    }
}

// ---- epilogue.  The accumulator layout gives a lane four ROWS of one column; stored as it is that is one 4-B store instruction per
// element — 192 per lane, 56 cycles each with every CU storing at once: 43 k cycles per tile against 88 k of k loop (stamps,
// profiles/r06/gemm_w256.md), and nothing covers them at one wave per SIMD.  Instead every strip of 16 rows x 64 columns goes through
// 4 KB of LDS of this wave (the second W buffer is free behind the last barrier of the k loop: 12 KB per wave, two strips) and comes
// back row-major: 16 B per lane, 4 rows x 256 B per instruction, 48 non-temporal 16-B stores per lane.  LDS executes a wave's accesses in
// order, so the write -> read -> rewrite of a strip buffer needs no wait of its own.  Written: ds_write_b32, rows 4 lg + i at a stride of
// 256 B (2-way on the store, free); read: ds_read_b128, lane (row = lane >> 4 (+ 4 j), quad = lane & 15): conflict-free.
// Per element the operations of gemm_epilogue16, in its order: same bits.
// NQ column quarters of 64 (4: a 256-column block, 2: a 128-column block); acc_at(sq, h, tt): the accumulator tile of column tile tt of quarter sq,
// row group h; tb: 8 KB of LDS of this wave.  Columns >= p.n (n % 4 == 0) are skipped.
template <int RG, int NQ, class AccAt>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& p, AccAt&& acc_at, int64_t m0, int n0, int wave, int lane, float* tb) {
    {
    const int er = lane >> 4, eq = lane & 15, ln = lane & 15, lg = lane >> 4;
    const int64_t mw = m0 + 16 * RG * wave;
#pragma unroll
    for (int sq = 0; sq < NQ; ++sq) {
        const int col = n0 + 64 * sq + 4 * eq;  // this lane's four columns of the quarter
        const bool col_ok = col < p.n;
        const int colc = col_ok ? col : 0;
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 bias = p.bias ? *reinterpret_cast<const f4*>(p.bias + colc) : f4{0.f, 0.f, 0.f, 0.f};
        f4 alpha = {0.f, 0.f, 0.f, 0.f}, inv_alpha = alpha, gamma = alpha, beta = alpha;
        if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {
            alpha = *reinterpret_cast<const f4*>(p.alpha + colc);
            inv_alpha = *reinterpret_cast<const f4*>(p.inv_alpha + colc);
        }
        if (p.epi == EPI_SNAKE_GRN) {
            gamma = *reinterpret_cast<const f4*>(p.gamma + colc);
            beta = *reinterpret_cast<const f4*>(p.beta + colc);
        }
#pragma unroll
        for (int h = 0; h < RG; ++h) {
            float* const sb = tb + 1024 * ((sq * RG + h) & 1);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < 4; ++i) sb[(4 * lg + i) * 64 + 16 * tt + ln] = acc_at(sq, h, tt)[i];
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                const int64_t m = mw + 16 * h + er + 4 * jr;
                f4 v = *reinterpret_cast<const f4*>(sb + (er + 4 * jr) * 64 + 4 * eq);
                if (m >= p.m || !col_ok) continue;
                v = v + bias;
                if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {
                    const f32x2 s0 = snake_act2(f32x2{v.x, v.y}, f32x2{alpha.x, alpha.y}, f32x2{inv_alpha.x, inv_alpha.y});
                    const f32x2 s1 = snake_act2(f32x2{v.z, v.w}, f32x2{alpha.z, alpha.w}, f32x2{inv_alpha.z, inv_alpha.w});
                    if (p.epi == EPI_SNAKE_GRN) {  // layers.py:115, n_x == 1
                        const f32x2 o0 = __builtin_elementwise_fma(f32x2{gamma.x, gamma.y}, s0, f32x2{beta.x, beta.y}) + s0;
                        const f32x2 o1 = __builtin_elementwise_fma(f32x2{gamma.z, gamma.w}, s1, f32x2{beta.z, beta.w}) + s1;
                        v = f4{o0.x, o0.y, o1.x, o1.y};
                    } else {
                        v = f4{s0.x, s0.y, s1.x, s1.y};
                    }
                } else if (p.epi == EPI_BIAS_RES) {
                    v = *reinterpret_cast<const f4*>(p.res + m * p.ldres + col) + v;
                }
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p.c + m * p.ldc + col));
            }
        }
    }
}
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void tail_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        tail_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

}  // namespace
