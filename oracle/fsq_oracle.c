/*
 * fsq_oracle.c — plain-C restatement of the FSQ quantiser arithmetic (TEST INFRASTRUCTURE ONLY; nothing under
 * l3ac_amd/ links or loads this file).
 *
 * Follows the reference's SuperFSQ in eval mode:
 *   act  = (tanh(z) + 1) / 2                         l3ac/vq/fsq_act.py:38-39
 *   li   = round_half_even(act * (L - 1))            l3ac/vq/fsq.py:59   (torch.round == rintf under FE_TONEAREST)
 *   idx  = (int32) sum_d li_d * basis_d   in fp32    l3ac/vq/fsq.py:67-68, basis = cumprod([1] + levels[:-1]) :15
 *   q    = li / (L - 1) * 2 - 1                      l3ac/vq/fsq.py:60, :21
 * and the decode direction  li = (idx / basis) % levels  (l3ac/vq/fsq.py:70-71).
 * Checked against the reference's own outputs (tests/golden/fsq_kat.npz) by tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>

void fsq_oracle_quantize(const float* z, int64_t n, int32_t d, const int32_t* levels, float* q, int32_t* idx,
                         float* level_indices) {
    for (int64_t i = 0; i < n; ++i) {
        float sum = 0.0f;
        float basis = 1.0f;
        for (int32_t k = 0; k < d; ++k) {
            const float lm1 = (float)(levels[k] - 1);
            const float act = (tanhf(z[i * d + k]) + 1.0f) / 2.0f;
            const float li = rintf(act * lm1);
            const float q_act = li / lm1;
            q[i * d + k] = q_act * 2.0f - 1.0f;
            level_indices[i * d + k] = li;
            sum += li * basis;
            basis *= (float)levels[k];
        }
        idx[i] = (int32_t)sum;
    }
}

/* the rounding half alone, from act = (tanh(z) + 1) / 2 (SuperFSQ.quantize_act_value, l3ac/vq/fsq.py:56-65, then :67-68, :21) */
void fsq_oracle_quantize_act(const float* act, int64_t n, int32_t d, const int32_t* levels, float* q, int32_t* idx,
                             float* level_indices) {
    for (int64_t i = 0; i < n; ++i) {
        float sum = 0.0f;
        float basis = 1.0f;
        for (int32_t k = 0; k < d; ++k) {
            const float lm1 = (float)(levels[k] - 1);
            const float li = rintf(act[i * d + k] * lm1);
            const float q_act = li / lm1;
            q[i * d + k] = q_act * 2.0f - 1.0f;
            level_indices[i * d + k] = li;
            sum += li * basis;
            basis *= (float)levels[k];
        }
        idx[i] = (int32_t)sum;
    }
}

void fsq_oracle_indices_to_codes(const int32_t* idx, int64_t n, int32_t d, const int32_t* levels, float* codes) {
    for (int64_t i = 0; i < n; ++i) {
        int32_t basis = 1;
        for (int32_t k = 0; k < d; ++k) {
            const int32_t li = (idx[i] / basis) % levels[k];
            codes[i * d + k] = (float)li / (float)(levels[k] - 1) * 2.0f - 1.0f;
            basis *= levels[k];
        }
    }
}

/* brute-force nearest neighbour over an explicit codebook, lowest index on ties (the search FSQ is the closed form of) */
void fsq_oracle_argmin(const float* queries, int64_t n, const float* codebook, int32_t k, int32_t d, int32_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        float best = INFINITY;
        int32_t best_j = 0;
        for (int32_t j = 0; j < k; ++j) {
            float dist = 0.0f;
            for (int32_t c = 0; c < d; ++c) {
                const float t = queries[i * d + c] - codebook[(int64_t)j * d + c];
                dist = fmaf(t, t, dist);
            }
            if (dist < best) {
                best = dist;
                best_j = j;
            }
        }
        out[i] = best_j;
    }
}
