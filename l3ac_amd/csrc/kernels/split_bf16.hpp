// bf16x3 operand splitting shared by gemm_split.hip and the fused kernels.
//
//   x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   (round to nearest even)
//
// Three 8-bit significands cover the 24-bit fp32 significand, so the split is exact; a product a.w is the six plane
// products a_i.w_j with i + j <= 2 (each exact in fp32, the dropped ones are <= 2^-26 |a.w|) summed smallest first in
// the MFMA's fp32 accumulator.
#pragma once

#include <cstdint>
#include <cstring>

#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (arrays of HIP's uint4 struct are not promoted to registers)
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// two fp32 -> three packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const f32x2_t v = {x0, x1};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = {x0 - __builtin_bit_cast(float, p0 << 16), x1 - __builtin_bit_cast(float, p0 & 0xffff0000u)};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t q = {r.x - __builtin_bit_cast(float, p1 << 16), r.y - __builtin_bit_cast(float, p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
}

// acc += a . b over one k step of 16, operands given as their three planes (index = plane)
__device__ __forceinline__ f32x16_t mfma_split(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16_t acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// A 32 x 32 fp32 accumulator tile X (column on the lane, rows (r&3) + 8(r>>2) + 4h in the 16 registers) as the B operand
// of the next 32x32x16 products, Y = W . X summed over X's ROWS: registers 8s..8s+7 become k step s, whose element j in
// lane half h is row 16s + 8(j>>2) + 4h + (j&3) of X — the other operand's fragments must use the same k order
// (SIGMA below; guide §3 "An accumulator tile as the next MFMA's operand").
__device__ __forceinline__ void split_acc_tile(const f32x16_t& x, bf16x8 (&out)[2][3]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned p[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) out[s][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
    }
}
__host__ __device__ inline int split_sigma(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// host side (weights are finite)
inline uint16_t bf16_rne_host(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32_host(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline void split3_host(float x, uint16_t (&out)[3]) {
    out[0] = bf16_rne_host(x);
    const float r1 = x - bf16_to_f32_host(out[0]);
    out[1] = bf16_rne_host(r1);
    out[2] = bf16_rne_host(r1 - bf16_to_f32_host(out[1]));
}
