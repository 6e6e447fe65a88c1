// Epilogue shared by the GEMM kernels (gemm_f32.hip: exact-fp32 MFMA; gemm_split.hip: bf16x3 split operands).
// Both keep a 32-row x (32 NT)-column strip per wave in MFMA accumulator layout.
#pragma once

#include "../kernels.hpp"
#include "device_math.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- epilogue: lane holds column n = n0 + 32 nt + li, rows m0 + 32 wave + (r&3) + 8 (r>>2) + 4 lh ----
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[NT], int64_t m0, int n0, int wave, int li, int lh) {
    const int64_t mw = m0 + 32 * wave + 4 * lh;
    if (p.epi == EPI_GEGLU) {
        // column tiles come in (value, gate) pairs; output column j = n0/2 + 32 (nt/2) + li
#pragma unroll
        for (int nt = 0; nt + 1 < NT; nt += 2) {
            const int j = (n0 >> 1) + 16 * nt + li;
            if (j < (int)p.ldc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = mw + (r & 3) + 8 * (r >> 2);
                    if (m < p.m) p.c[m * p.ldc + j] = acc[nt][r] * gelu_erf(acc[nt + 1][r]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 32 * nt + li;
        if (n >= p.n) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        float alpha = 0.f, inv_alpha = 0.f, gamma = 0.f, beta = 0.f;
        if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {
            alpha = p.alpha[n];
            inv_alpha = p.inv_alpha[n];
        }
        if (p.epi == EPI_SNAKE_GRN) {
            gamma = p.gamma[n];
            beta = p.beta[n];
        }
        if (p.epi == EPI_SNAKE || p.epi == EPI_SNAKE_GRN) {  // packed fp32 math on register pairs (two rows, same column)
            const f32x2 al = (f32x2)(alpha), ia = (f32x2)(inv_alpha), ga = (f32x2)(gamma), be = (f32x2)(beta), bi = (f32x2)(bias);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 hv;
                hv.x = acc[nt][r];
                hv.y = acc[nt][r + 1];
                const f32x2 sv = snake_act2(hv + bi, al, ia);
                const f32x2 o = p.epi == EPI_SNAKE_GRN ? __builtin_elementwise_fma(ga, sv, be) + sv : sv;  // layers.py:115, n_x == 1
                const int64_t m = mw + (r & 3) + 8 * (r >> 2);
                if (m < p.m) p.c[m * p.ldc + n] = o.x;
                if (m + 1 < p.m) p.c[(m + 1) * p.ldc + n] = o.y;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = mw + (r & 3) + 8 * (r >> 2);
            if (m >= p.m) continue;
            float v = acc[nt][r] + bias;
            if (p.epi == EPI_BIAS_RES) v = p.res[m * p.ldres + n] + v;
            p.c[m * p.ldc + n] = v;
        }
    }
}
