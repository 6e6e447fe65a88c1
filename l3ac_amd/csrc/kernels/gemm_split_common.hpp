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

template <int N, class F, int I = 0>
__device__ __forceinline__ void tail_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        tail_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

}  // namespace
