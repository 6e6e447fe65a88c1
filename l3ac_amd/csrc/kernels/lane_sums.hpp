// Sums over the lanes of a wave without the LDS crossbar (shared by the row kernels and the fused units).
#pragma once

#include <hip/hip_runtime.h>

// Reductions over the four 16-lane rows of a wave (the four k groups that share a column of an accumulator tile) without the LDS
// crossbar: v_permlane16_swap exchanges the odd rows of one register with the even rows of another, v_permlane32_swap the upper
// half of one with the lower half of another; applied to two copies of x they leave (x[l & ~16], x[l | 16]) resp. (x[l & ~32],
// x[l | 32]) in every lane.  A ds_bpermute round trip (what __shfl_xor(x, 16 / 32) compiles to) is ~100+ cycles of latency on the
// critical path of a softmax step; these are two vector instructions.  Inline asm as in gemm_split.hip (hipcc 7.2 folds repeated
// builtin swaps); the s_nop covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void row_pair16(float x, float& lo, float& hi) {
    lo = x, hi = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ void row_pair32(float x, float& lo, float& hi) {
    lo = x, hi = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float rows_sum(float x) {  // sum over lanes l, l ^ 16, l ^ 32, l ^ 48: the same bits in all four
    float a, b;
    row_pair16(x, a, b);
    x = a + b;
    row_pair32(x, a, b);
    return a + b;
}
__device__ __forceinline__ float rows_max(float x) {
    float a, b;
    row_pair16(x, a, b);
    x = fmaxf(a, b);
    row_pair32(x, a, b);
    return fmaxf(a, b);
}

// Sum over aligned groups of `lanes` consecutive lanes (a power of two, 1 .. 64), the same bits in every lane of a group: quad swaps and
// row mirrors as DPP operands of the adds (lanes of one 16-lane row), then rows_sum's permlane swaps across rows.  __shfl_xor compiles to
// ds_bpermute_b32 — one dependent LDS-crossbar round trip of ~100 cycles per step (six per 64-lane sum): measured on
// dwconv_ln_split_kernel<256>, whose two LayerNorm sums per frame were twelve of them, 0.448-0.464 -> 0.394-0.400 ms for the step's three
// launches.  The order of the additions is fixed (batch invariance) — a different fixed order than the xor butterfly's.
__device__ __forceinline__ float lanes_sum(float v, const int lanes) {  // `lanes` wave-uniform
#define L3AC_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    if (lanes >= 2) L3AC_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    if (lanes >= 4) L3AC_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    if (lanes >= 8) L3AC_DPP_ADD(0x141);  // row_half_mirror: the other quad of the half row
    if (lanes >= 16) L3AC_DPP_ADD(0x140); // row_mirror: the other half row
#undef L3AC_DPP_ADD
    if (lanes >= 32) {
        float a, b;
        row_pair16(v, a, b);
        v = a + b;
    }
    if (lanes >= 64) {
        float a, b;
        row_pair32(v, a, b);
        v = a + b;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return lanes_sum(v, 64); }
// v[l] + v[l ^ 32] (the same bits as v + __shfl_xor(v, 32): the addition commutes)
__device__ __forceinline__ float halves_sum(float v) {
    float a, b;
    row_pair32(v, a, b);
    return a + b;
}
