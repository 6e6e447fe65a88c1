// Shared by the kernels that run "weights (A) x activations (B)" on v_mfma_f32_16x16x32_bf16 with both operands as exact bf16x3
// splits and the weights streamed through an LDS ring as fragment-ordered pieces (trans_stack.hip, conv_unit_ring.hip).
//
//   piece      16 rows x 32 k of a weight matrix as an operand fragment: 3 planes x 1 KB, lane (m = lane & 15, g = lane >> 4) holds the
//              8 bf16 W[row0 + m][k0 + sigma(g, j)], sigma(g, j) = (j < 4 ? 4 g + j : 16 + 4 g + j - 4) — the order in which two
//              16-row ACCUMULATOR tiles (rows 16 t + 4 g + i of this lane's column) line up as the other operand's k step of 32
//              (guide: 'An accumulator tile as the next MFMA's operand').
#pragma once

#include <cstring>
#include <type_traits>
#include <vector>

#include "lane_sums.hpp"
#include "split_bf16.hpp"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// the six plane products of one fragment pair, smallest first (split_bf16.hpp, mfma_split)
__device__ __forceinline__ f32x4_t mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4_t acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// two accumulator tiles (rows 16 t + 4 g + i and 16 (t + 1) + 4 g + i of this lane's column) as the three bf16 planes of one
// k step of 32 in the order sigma(g, j)
__device__ __forceinline__ void planes_of(const f32x4_t& lo, const f32x4_t& hi, bf16x8 (&out)[3]) {
    unsigned p[3][4];
    split2(lo[0], lo[1], p[0][0], p[1][0], p[2][0]);
    split2(lo[2], lo[3], p[0][1], p[1][1], p[2][1]);
    split2(hi[0], hi[1], p[0][2], p[1][2], p[2][2]);
    split2(hi[2], hi[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
}

// this wave's quarter (3 KB) of one ring slot: three 1-KB LDS-DMA pieces, lane l copying 16 B (as conv_unit_wide.hip)
__device__ __forceinline__ void ring_dma_quarter(const unsigned char* base, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(base), "s"(lds_dst)
        : "memory");
}

// one 1-KB block of a ring slot (lane l copies 16 B from base + 16 l to lds_dst + 16 l)
__device__ __forceinline__ void ring_dma_1k(const unsigned char* base, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(base), "s"(lds_dst)
        : "memory");
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void ring_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ring_static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}


// one piece: 16 rows x 32 k of a row-major [n][ld] matrix (rows >= n_rows / columns >= n_cols: zeros) as the three bf16 planes of
// the operand fragment — lane (m = lane & 15, g = lane >> 4) holds the 8 values W[row0 + m][k0 + sigma(g, j)], j = 0 .. 7
inline void ring_put_piece(std::vector<unsigned char>& img, const float* w, int64_t ld, int n_rows, int n_cols, int row0, int k0) {
    const size_t base = img.size();
    img.resize(base + 3072, 0);
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 16; ++m)
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4);
                const int r = row0 + m;
                const float v = (r < n_rows && k < n_cols) ? w[(int64_t)r * ld + k] : 0.f;
                uint16_t pl[3];
                split3_host(v, pl);
                for (int p = 0; p < 3; ++p) std::memcpy(img.data() + base + (size_t)p * 1024 + (size_t)(16 * g + m) * 16 + 2 * j, &pl[p], 2);
            }
}

