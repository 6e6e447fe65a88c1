// FSQ quantiser kernels (reference l3ac/vq/__init__.py:20-30, l3ac/vq/fsq.py:30-81, l3ac/vq/fsq_act.py:38-39).
//
// fsq_kernel — the closed form, fused:   x [n][feat]
//   -> lat = x . W_in^T + b_in            (nn.Linear(feat, D))
//   -> act = (tanh(lat) + 1) / 2
//   -> li  = round_half_even(act * (L - 1))          (torch.round; v_rndne_f32)
//   -> idx = int32(sum_d li_d * basis_d)             (exact in fp32, < 2^24)
//   -> q   = li / (L - 1) * 2 - 1
//   -> out = q . W_out^T + b_out          (nn.Linear(D, feat))
// HBM-bound: 4*feat B in + 4*feat B out + 4 + 4*D B per token.  feat/8 lanes own one token (8 channels = two
// 16-B loads each); the D partial dot products are combined by a butterfly over those lanes, every lane then
// holds identical latents and finishes its own 8 output channels.  Everything after tanh is exact IEEE arithmetic,
// so indices are bit-identical to the reference for identical activations.
//
// vq_screen_kernel + vq_resolve_kernel / vq_scan_kernel / vq_wave_kernel — the explicit-codebook nearest-neighbour search FSQ
// is the closed form of: see "explicit codebook search" and "screened form" below; lowest index wins exact ties.
#include <atomic>

#include "../kernels.hpp"
#include "lane_sums.hpp"

namespace {

// minimumNumber is the hardware's v_min / v_min3 as it stands; fminf first quiets every operand with a v_max.  Older clang
// has no such builtin: fminf gives the same values (the operands are finite or +inf here), only slower.
__device__ __forceinline__ float vq_min(float a, float b) {
#if __has_builtin(__builtin_elementwise_minimumnum)
    return __builtin_elementwise_minimumnum(a, b);
#else
    return fminf(a, b);
#endif
}


constexpr int THREADS = 256;
constexpr int MAXD = L3AC_MAX_LEVELS;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FsqDev {
    const float* x;
    int64_t n;
    int feat, d;
    int levels[MAXD];
    int basis[MAXD];
    const float* w_in;
    const float* b_in;
    const float* w_out;
    const float* b_out;
    const int32_t* idx_in;
    float* q_feature;
    int32_t* indices;
    float* level_indices;
    float* latents;
    int act_in;  // the `latents` input already holds act = (tanh(lat) + 1) / 2 (SuperFSQ.quantize_act_value's argument)
    int k_total;      // codebook size: valid indices are 0 .. k_total - 1
    int* bad_count;   // optional device counter of out-of-range input indices (which are clamped into range)
};

// streamed once: non-temporal accesses keep the rows out of the way of L2 / MALL residents
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
    __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
}

// Every mode: forward, quantise-from-activations, decode-from-indices, latents in / out; any feat = 8 * 2^j.
// NV 16-byte chunks (4 channels each) per lane: lanes per token = feat / (4 NV).
// Round 3 measurements (profiles/r03/fsq_notes.md): hipcc keeps most of the 2 D NV projection-weight quads of a lane in registers
// across the token loop (158 registers, 3 blocks per CU).  A variant that re-read them from LDS at 64-96 registers and 4 / 6 / 8
// blocks per CU was SLOWER: removed.  What paid: one resident round, the level indices as ONE contiguous store per wave, and
// (round 4, profiles/r04/fsq_notes.md) the ORDER in which the chip walks the tokens + non-temporal accesses: 4.97 -> 5.46 TB/s.
// Access order and cache policy of the streamed rows (round 4, tools/probes/fsq_pattern_probe.hip + profiles/r04/fsq_notes.md: the
// quantiser's traffic with no arithmetic, by how tokens are dealt to blocks / waves / lanes): tokens are walked GRID-STRIDE (block b
// takes groups b, b + G, ...: the chip sweeps one compact window of memory at a time; one contiguous range per block was 3 % slower,
// grid-stride inside each XCD's own eighth no better), and every streamed access — x, q_feature, indices, level indices — is
// non-temporal (+ 5 %).
template <typename T>
__device__ __forceinline__ void side_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
// the it-th token group of block `b` of `G` (n_groups = none left)
__device__ __forceinline__ int64_t fsq_group_at(int64_t it, int64_t b, int64_t G, int64_t n_groups) {
    const int64_t g = it * G + b;
    return g < n_groups ? g : n_groups;
}

template <int D, int NV>
__global__ __launch_bounds__(THREADS) void fsq_kernel(const FsqDev p, const int lpt) {
    const int feat = p.feat;
    // HBM-bound by construction (1 052 B per token); what limits it in practice is bytes in flight and issue slots,
    // so: projection weights in LDS (not registers), the next token group's rows prefetched into registers, each
    // block walking a contiguous token range, and ONE tanh per lane (lane d of a token's group quantises latent d
    // and the level indices are exchanged by shuffles) instead of D redundant ones.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Win = smem;                  // [D][feat]
    float* Wout = smem + D * feat;    // [D][feat] (project_out transposed)
    float* Bout = Wout + D * feat;    // [feat]
    const int tid = threadIdx.x;
    for (int i = tid; i < D * feat; i += THREADS) {
        const int d = i / feat, c = i % feat;
        Win[i] = p.w_in ? p.w_in[i] : 0.f;
        Wout[i] = p.w_out[c * D + d];
    }
    for (int i = tid; i < feat; i += THREADS) Bout[i] = p.b_out[i];
    __syncthreads();

    const int sub = tid % lpt;          // lane within the token's group
    const int lane = tid & 63;
    // channel quads are dealt round-robin over the lanes of a token: quad v of lane `sub` = channels 4 (v lpt + sub) .. +3,
    // so every load / store instruction of a wave covers whole 128-B lines (8 lanes x 16 B contiguous per token)
    const int cq = 4 * sub, cstep = 4 * lpt;
    const int tok_per_block = THREADS / lpt;
    const int64_t n_groups = (p.n + tok_per_block - 1) / tok_per_block;
    const bool spread = lpt >= D;       // one latent per lane; otherwise every lane quantises all D

    float4 x_next[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) x_next[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t g) {
        const int64_t t = g * tok_per_block + tid / lpt;
        if (p.x && g < n_groups && t < p.n) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float* src = p.x + t * feat + cq + cstep * v;
                x_next[v] = nt_load4(src);
            }
        }
    };
    auto quantise = [&](float lat, int d) -> float {
        const float act = p.act_in ? lat : (tanhf(lat) + 1.0f) * 0.5f;  // fsq_act.py:39
        return rintf(__fmul_rn(act, (float)(p.levels[d] - 1)));       // vq/fsq.py:59 (half-to-even)
    };
    int64_t g_next = fsq_group_at(0, blockIdx.x, gridDim.x, n_groups);
    fetch(g_next);
    for (int64_t it = 0; g_next < n_groups; ++it) {
        const int64_t g = g_next;
        const int64_t tok = g * tok_per_block + tid / lpt;
        const bool ok = tok < p.n;
        const int64_t tk = ok ? tok : 0;
        float4 xv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[v] = x_next[v];
        g_next = fsq_group_at(it + 1, blockIdx.x, gridDim.x, n_groups);
        fetch(g_next);
        float li[D];
        float li_mine = 0.f;  // spread mode: level index `sub` of this token (lanes sub < D)
        if (p.idx_in) {  // decode: indices -> level indices (vq/fsq.py:70-71)
            int idx = p.idx_in[tk];
            if ((unsigned)idx >= (unsigned)p.k_total) {  // corrupted / truncated stream: counted, clamped (never a wrapped level)
                if (p.bad_count && ok && sub == 0) atomicAdd(p.bad_count, 1);
                idx = idx < 0 ? 0 : p.k_total - 1;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) li[d] = (float)((idx / p.basis[d]) % p.levels[d]);
        } else {
            float lat[D];
            if (p.x) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    float s = 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float4 wv = *reinterpret_cast<const float4*>(Win + d * feat + cq + cstep * v);
                        s = fmaf(xv[v].x, wv.x, s); s = fmaf(xv[v].y, wv.y, s);
                        s = fmaf(xv[v].z, wv.z, s); s = fmaf(xv[v].w, wv.w, s);
                    }
                    s = lanes_sum(s, lpt);  // (DPP adds instead of log2(lpt) ds_bpermute round trips per latent: lane_sums.hpp)
                    lat[d] = s + (p.b_in ? p.b_in[d] : 0.f);
                }
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) lat[d] = p.latents[tk * D + d];
            }
            if (p.x && p.latents && ok && sub == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) p.latents[tok * D + d] = lat[d];
            }
            if (spread) {
                float mine = 0.f;  // latent `sub` of this token (all lanes of the group hold identical latents)
#pragma unroll
                for (int d = 0; d < D; ++d) mine = sub == d ? lat[d] : mine;
                li_mine = quantise(mine, sub < D ? sub : 0);
                const int group_base = lane - sub;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (lpt == 8) {  // lane d of the token's eight: row_newbcast of lane d / 8 + d of the 16-lane row, chosen by the lane's half
#define L3AC_BC(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, li_mine), 0x150 + (n), 0xf, 0xf, false))
                        float lo, hi;
                        switch (d) {
                            case 0: lo = L3AC_BC(0), hi = L3AC_BC(8); break;
                            case 1: lo = L3AC_BC(1), hi = L3AC_BC(9); break;
                            case 2: lo = L3AC_BC(2), hi = L3AC_BC(10); break;
                            case 3: lo = L3AC_BC(3), hi = L3AC_BC(11); break;
                            case 4: lo = L3AC_BC(4), hi = L3AC_BC(12); break;
                            case 5: lo = L3AC_BC(5), hi = L3AC_BC(13); break;
                            case 6: lo = L3AC_BC(6), hi = L3AC_BC(14); break;
                            default: lo = L3AC_BC(7), hi = L3AC_BC(15); break;
                        }
#undef L3AC_BC
                        li[d] = (lane & 8) ? hi : lo;
                        continue;
                    }
                    li[d] = __shfl(li_mine, group_base + d, 64);
                }
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) li[d] = quantise(lat[d], d);
            }
        }
        float idx_f = 0.f;
        float4 o[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) o[v] = *reinterpret_cast<const float4*>(Bout + cq + cstep * v);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            idx_f += li[d] * (float)p.basis[d];                                 // exact (vq/fsq.py:67-68)
            const float q_act = __fdiv_rn(li[d], (float)(p.levels[d] - 1));     // vq/fsq.py:60
            const float q = __fsub_rn(__fmul_rn(q_act, 2.0f), 1.0f);            // vq/fsq.py:21
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 wv = *reinterpret_cast<const float4*>(Wout + d * feat + cq + cstep * v);
                o[v].x = fmaf(q, wv.x, o[v].x); o[v].y = fmaf(q, wv.y, o[v].y);
                o[v].z = fmaf(q, wv.z, o[v].z); o[v].w = fmaf(q, wv.w, o[v].w);
            }
        }
        if (ok) {
            float* dst = p.q_feature + tok * feat + cq;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                nt_store4(dst + cstep * v, o[v]);
            }
            if (sub == 0 && p.indices) side_store(p.indices + tok, (int32_t)idx_f);
            if (p.level_indices) {
                if (spread && !p.idx_in) {
                    // lane d of the token's group holds level index d: ONE store instruction per wave, the tokens' D values
                    // contiguous (a wave's tokens are consecutive: 8 x 24 B in a row) instead of D scattered 4-byte stores
                    if (sub < D) side_store(p.level_indices + tok * D + sub, li_mine);
                } else if (sub == 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) side_store(p.level_indices + tok * D + d, li[d]);
                }
            }
        }
    }
}

// ---- the hot form (round 5): forward pass at feat = 128 — 8 lanes per token, 4 channel quads per lane — with everything the generic
// kernel decides at run time fixed at compile time, and the per-token tail done ONCE per latent instead of once per lane:
//   * lane d of a token's eight owns latent d end to end: tanh, round, and — new — the dequantised value q_d = li / (L - 1) * 2 - 1
//     (one exact division per lane instead of six per lane) and its term li_d * basis_d of the index;
//   * the six q_d reach the token's lanes by DPP row broadcasts (as the level indices did), the index by a three-step DPP lane sum
//     (small integers: exact in any order); the level indices are stored by their owners and never exchanged.
// Per token the arithmetic that produces latents, level indices, index and q_feature is operation for operation the generic kernel's
// (same fmaf chains, same lane-sum order, the same exact IEEE tail): the same bits (tests: both kernels against the reference's
// known-answer vectors and against each other).  ~60 fewer vector instructions and ~100 fewer scalar branches per token group: on
// boxes that hold a lower shader clock the generic kernel ran at 0.89 of its copy ceiling (the driver's round-4 box, one of this round's).
// Registers: compiled for TWO workgroups per CU hipcc keeps the 2 D x 16 projection weights of a lane in registers across the token
// loop (200 at D = 6) and the loop reads nothing from LDS.  (Forcing it — explicit register arrays — spills 22; requesting two token
// groups ahead makes hipcc re-read the weights from LDS again.)  Measured on one box, two interleaved rounds (profiles/r05/fsq_hot_ab.txt, fractions of 8 TB/s): the
// generic kernel 0.691; this kernel at three workgroups per CU (168 registers, 24 spilled) 0.635; weights re-read from LDS every token
// group (92 registers, five workgroups per CU) 0.66-0.73; two workgroups per CU 0.767 = 1.01 x the copy kernel of the same residency.
template <int D>
__global__ __launch_bounds__(THREADS, 2) void fsq_forward128_kernel(const FsqDev p) {
    constexpr int FEAT = 128, NV = 4, LPT = 8, TPB = THREADS / LPT;
    static_assert(D <= LPT, "one latent per lane of a token's group");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Win = smem;                // [D][FEAT]
    float* Wout = smem + D * FEAT;    // [D][FEAT] (project_out transposed)
    float* Bout = Wout + D * FEAT;    // [FEAT]
    const int tid = threadIdx.x;
    for (int i = tid; i < D * FEAT; i += THREADS) {
        const int d = i / FEAT, c = i % FEAT;
        Win[i] = p.w_in[i];
        Wout[i] = p.w_out[c * D + d];
    }
    for (int i = tid; i < FEAT; i += THREADS) Bout[i] = p.b_out[i];
    __syncthreads();

    const int sub = tid & (LPT - 1), lane = tid & 63;
    const int cq = 4 * sub;
    constexpr int cstep = 4 * LPT;
    const int64_t n_groups = (p.n + TPB - 1) / TPB;
    // this lane's latent (lanes sub >= D shadow latent 0: their results are never used)
    const int dm = sub < D ? sub : 0;
    const float lm1 = (float)(p.levels[dm] - 1);
    const float basis_m = sub < D ? (float)p.basis[dm] : 0.f;
    const float bin_m = p.b_in ? p.b_in[dm] : 0.f;

    float4 x_next[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) x_next[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t g) __attribute__((always_inline)) {
        const int64_t t = g * TPB + tid / LPT;
        if (g < n_groups && t < p.n) {
#pragma unroll
            for (int v = 0; v < NV; ++v) x_next[v] = nt_load4(p.x + t * FEAT + cq + cstep * v);
        }
    };
    int64_t g_next = fsq_group_at(0, blockIdx.x, gridDim.x, n_groups);
    fetch(g_next);
    for (int64_t it = 0; g_next < n_groups; ++it) {
        const int64_t tok = g_next * TPB + tid / LPT;
        const bool ok = tok < p.n;
        float4 xv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[v] = x_next[v];
        g_next = fsq_group_at(it + 1, blockIdx.x, gridDim.x, n_groups);
        fetch(g_next);
        // latent d = x . Win[d] + b_in[d]: partial dot of this lane's 16 channels, summed over the token's 8 lanes (lanes_sum's order)
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 wv = *reinterpret_cast<const float4*>(Win + d * FEAT + cq + cstep * v);
                s = fmaf(xv[v].x, wv.x, s); s = fmaf(xv[v].y, wv.y, s);
                s = fmaf(xv[v].z, wv.z, s); s = fmaf(xv[v].w, wv.w, s);
            }
            s = lanes_sum(s, LPT);
            mine = sub == d ? s : mine;
        }
        mine += bin_m;
        const float act = (tanhf(mine) + 1.0f) * 0.5f;                           // fsq_act.py:39
        const float li_mine = rintf(__fmul_rn(act, lm1));                          // vq/fsq.py:59 (half-to-even)
        const float q_mine = __fsub_rn(__fmul_rn(__fdiv_rn(li_mine, lm1), 2.0f), 1.0f);  // vq/fsq.py:60, :21
        const float idx_f = lanes_sum(li_mine * basis_m, LPT);                  // exact (vq/fsq.py:67-68): integers below 2^24
        float4 o[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) o[v] = *reinterpret_cast<const float4*>(Bout + cq + cstep * v);
#define L3AC_BCQ(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q_mine), 0x150 + (n), 0xf, 0xf, false))
        const bool upper = (lane & 8) != 0;  // the second token of the 16-lane row
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float lo, hi;  // q_d of the row's first / second token: row_newbcast of row lane d / 8 + d
            switch (d) {
                case 0: lo = L3AC_BCQ(0), hi = L3AC_BCQ(8); break;
                case 1: lo = L3AC_BCQ(1), hi = L3AC_BCQ(9); break;
                case 2: lo = L3AC_BCQ(2), hi = L3AC_BCQ(10); break;
                case 3: lo = L3AC_BCQ(3), hi = L3AC_BCQ(11); break;
                case 4: lo = L3AC_BCQ(4), hi = L3AC_BCQ(12); break;
                case 5: lo = L3AC_BCQ(5), hi = L3AC_BCQ(13); break;
                case 6: lo = L3AC_BCQ(6), hi = L3AC_BCQ(14); break;
                default: lo = L3AC_BCQ(7), hi = L3AC_BCQ(15); break;
            }
            const float q = upper ? hi : lo;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 wv = *reinterpret_cast<const float4*>(Wout + d * FEAT + cq + cstep * v);
                o[v].x = fmaf(q, wv.x, o[v].x); o[v].y = fmaf(q, wv.y, o[v].y);
                o[v].z = fmaf(q, wv.z, o[v].z); o[v].w = fmaf(q, wv.w, o[v].w);
            }
        }
#undef L3AC_BCQ
        if (ok) {
            float* dst = p.q_feature + tok * FEAT + cq;
#pragma unroll
            for (int v = 0; v < NV; ++v) nt_store4(dst + cstep * v, o[v]);
            if (sub == 0 && p.indices) side_store(p.indices + tok, (int32_t)idx_f);
            if (sub < D && p.level_indices) side_store(p.level_indices + tok * D + sub, li_mine);
        }
    }
}

// The ceiling fsq_kernel is measured against: the same grid, the same per-lane access pattern (4 x 16 B loads per lane one
// token group ahead, 4 x 16 B stores, 4 B per token of indices, 24 B per token of level indices) and NO arithmetic.
__global__ __launch_bounds__(THREADS) void fsq_copy_ceiling_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ q,
                                                                 int32_t* __restrict__ idx, float* __restrict__ li) {
    constexpr int FEAT = 128, NV = 4, LPT = 8, D = 6, TPB = THREADS / LPT;
    const int tid = threadIdx.x, sub = tid % LPT;
    const int cq = 4 * sub, cstep = 4 * LPT;
    const int64_t n_groups = (n + TPB - 1) / TPB;
    float4 nxt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) nxt[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t g) {
        const int64_t t = g * TPB + tid / LPT;
        if (g < n_groups && t < n) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float* src = x + t * FEAT + cq + cstep * v;
                nxt[v] = nt_load4(src);
            }
        }
    };
    int64_t g_next = fsq_group_at(0, blockIdx.x, gridDim.x, n_groups);
    fetch(g_next);
    for (int64_t it = 0; g_next < n_groups; ++it) {
        const int64_t tok = g_next * TPB + tid / LPT;
        float4 cur[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = nxt[v];
        g_next = fsq_group_at(it + 1, blockIdx.x, gridDim.x, n_groups);
        fetch(g_next);
        if (tok < n) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                nt_store4(q + tok * FEAT + cq + cstep * v, cur[v]);
            }
            if (sub == 0) side_store(idx + tok, __float_as_int(cur[0].x));
            if (sub < D) side_store(li + tok * D + sub, cur[0].y);
        }
    }
}

// resident workgroups per CU of a quantiser kernel at its LDS size (the occupancy query is a host-side computation; cached per device and form)
static int fsq_resident(const void* fn, size_t lds, int form) {
    static std::atomic<uint64_t> per_cu[L3AC_MAX_DEVICES][3] = {};
    const int slot = l3ac_device_slot();
    const uint64_t cached = slot >= 0 ? per_cu[slot][form].load(std::memory_order_relaxed) : 0;
    int resident = (cached >> 8) == (uint64_t)lds ? (int)(cached & 0xff) : 0;
    if (resident <= 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, THREADS, lds) != hipSuccess || v <= 0) v = 3;
        resident = v > 8 ? 8 : v;
        if (slot >= 0) per_cu[slot][form].store(((uint64_t)lds << 8) | (uint64_t)resident, std::memory_order_relaxed);
    }
    return resident;
}
// is this call the hot form (fsq_forward128_kernel)?
static bool fsq_is_hot(const FsqDev& p) { return p.x && !p.idx_in && !p.act_in && !p.latents && p.feat == 128 && p.d <= 8; }

template <int D>
int launch_fsq_t(hipStream_t s, const FsqDev& p) {
    const int nv = p.feat >= 128 ? 4 : 2;        // 16 or 8 channels per lane
    const int lpt = p.feat / (4 * nv);
    const int tok_per_block = THREADS / lpt;
    int64_t blocks = ceil_div64(p.n, tok_per_block);
    if (blocks <= 0) return L3AC_OK;
    const size_t lds = (size_t)(2 * D + 1) * p.feat * sizeof(float);
    // One weight staging per block, tokens walked grid-stride, and ONE resident round: with more blocks than the chip
    // holds at once the last, partly filled round streams at a fraction of the HBM rate for a whole block lifetime
    // (2 048 blocks on 6 x 256 places: the last quarter of the work at a third of the bytes in flight).
    const bool hot = fsq_is_hot(p);
    const int form = hot ? 2 : (nv == 4 ? 1 : 0);
    const void* fns[3] = {reinterpret_cast<const void*>(fsq_kernel<D, 2>), reinterpret_cast<const void*>(fsq_kernel<D, 4>),
                          reinterpret_cast<const void*>(fsq_forward128_kernel<D>)};
    const int64_t places = (int64_t)l3ac_device_cu_count() * fsq_resident(fns[form], lds, form);
    if (blocks > places) blocks = places;
    const double in_b = p.x ? 4.0 * p.feat : (p.idx_in ? 4.0 : 4.0 * D);
    ProfScope prof(s, "fsq_kernel", 4.0 * D * p.feat * (double)p.n,
                   (double)p.n * (in_b + 4.0 * p.feat + (p.indices ? 4.0 : 0.0) + (p.level_indices ? 4.0 * D : 0.0)));
    if (hot)
        hipLaunchKernelGGL((fsq_forward128_kernel<D>), dim3((unsigned)blocks), dim3(THREADS), lds, s, p);
    else if (form == 1)
        hipLaunchKernelGGL((fsq_kernel<D, 4>), dim3((unsigned)blocks), dim3(THREADS), lds, s, p, lpt);
    else
        hipLaunchKernelGGL((fsq_kernel<D, 2>), dim3((unsigned)blocks), dim3(THREADS), lds, s, p, lpt);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

// ---- explicit codebook search ----------------------------------------------------------------------
// Brute-force L2 nearest neighbour, dist = sum_d (q_d - c_d)^2 accumulated with fmaf in dimension order, strict '<' while
// the codes are visited in increasing index (the lowest index wins an exact tie): 3 D N K algorithmic FLOP, fp32-VALU-bound.
// Three forms, all specialised on the true dimension (no padded FMAs), all splitting the codebook into `parts` index ranges
// whose partial results a small second kernel combines in index order, all returning the same bits:
//   vq_screen_kernel (many queries: the default from 5 120 on) see "screened form" below: scores on the fp32 matrix pipe pick
//                   16 candidates per query, the arithmetic above decides among them.
//   vq_scan_kernel  (many queries; l3ac_set_vq_form(1): the direct form at scale, the reference the screened form is tested
//                   against)  QPL queries per lane in registers; the code coordinates are wave-uniform, so they come
//                   through the SCALAR unit (s_load) and enter the VALU instructions as SGPR operands: no LDS staging, no
//                   barrier, nothing but 2 D + 3 vector instructions per (query, code) pair.
//   vq_wave_kernel  (few queries: a streaming chunk has 60)  the codebook slice is dealt over the 64 LANES, QW queries per
//                   wave held wave-uniform; every lane keeps a running minimum over its codes and the wave then reduces
//                   (distance, index) pairs by xor-butterfly, the lower index winning ties — so even one query fills a
//                   wave, and 60 of them the chip.
template <int D, int QPL>
__global__ __launch_bounds__(256) void vq_scan_kernel(const float* __restrict__ queries, int64_t n, const float* __restrict__ codebook,
                                                     int k, int k_per_part, float* __restrict__ part_dist, int32_t* __restrict__ part_idx) {
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * QPL) + (threadIdx.x & 63);
    float q[QPL][D];
    float best[QPL];
    int best_i[QPL];
#pragma unroll
    for (int j = 0; j < QPL; ++j) {
        const int64_t qi = q0 + 64 * j;
#pragma unroll
        for (int d = 0; d < D; ++d) q[j][d] = qi < n ? queries[qi * D + d] : 0.f;
        best[j] = INFINITY;
        best_i[j] = 0x7fffffff;
    }
    const int c_begin = blockIdx.y * k_per_part;
    const int c_end = min(k, c_begin + k_per_part);
    const float* __restrict__ cb = codebook + (int64_t)c_begin * D;  // wave-uniform: scalar loads
    // The running state per query is (best distance, first code of the BLOCK that brought it): a block of CB codes costs its
    // arithmetic, CB - 1 minimum instructions (v_min3 / v_min) and ONE compare + two selects, instead of a compare + two selects per
    // code.  Strict '<' between blocks keeps the first block that reaches the final minimum; which of its codes it was is found
    // afterwards by re-evaluating that one block (same instructions, hence the same bits) and taking the first code whose distance
    // equals the minimum — the lowest index wins exact ties exactly as in the code-by-code scan.  (An earlier variant redid the
    // ordered update inside the loop on a wave-uniform branch: 6.1 instead of 3.3 ms, some lane improves in most blocks.)
    auto dist_of = [&](const float (&cc)[D], int j) __attribute__((always_inline)) -> float {
        float t = q[j][0] - cc[0];
        float dist = t * t;
#pragma unroll
        for (int d = 1; d < D; ++d) {
            t = q[j][d] - cc[d];
            dist = fmaf(t, t, dist);
        }
        return dist;
    };
    auto visit = [&](const float (&cc)[D], int c) __attribute__((always_inline)) {  // a block of one code (the slice's remainder)
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const float dist = dist_of(cc, j);
            const bool lt = dist < best[j];
            best[j] = lt ? dist : best[j];
            best_i[j] = lt ? c : best_i[j];
        }
    };
    // Blocks of CB codes, two register sets in ping-pong: the scalar loads of the next block are issued BEFORE the current one
    // is evaluated (sched_barrier keeps them there), so the s_waitcnt in front of a block's first vector instruction finds its
    // coordinates already there.  One code per iteration left the wave waiting for its s_load in every iteration: 49 instead
    // of the instruction stream's 31 cycles per (query, code) pair.
    constexpr int CB = 4;
    float blk_a[CB][D], blk_b[CB][D];
    auto load_block = [&](float (&dst)[CB][D], const float* __restrict__ src) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int d = 0; d < D; ++d) dst[i][d] = src[i * D + d];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto visit_block = [&](const float (&src)[CB][D], int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            static_assert(CB == 4, "block minimum written for four codes");
            const float m = fminf(__builtin_fminf(__builtin_fminf(dist_of(src[0], j), dist_of(src[1], j)), dist_of(src[2], j)), dist_of(src[3], j));
            const bool lt = m < best[j];  // strict: the first block wins
            best[j] = lt ? m : best[j];
            best_i[j] = lt ? c0 : best_i[j];
        }
        // (the running minima end every dependency chain of the block: naming them pins the block's arithmetic in front of the
        // wait + loads that follow — sched_barrier alone orders only what instruction selection has already put there)
#pragma unroll
        for (int j = 0; j < QPL; ++j) asm volatile("" : "+v"(best[j]), "+v"(best_i[j])::"memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // scalar loads return out of order, so the only wait there is is lgkmcnt(0): it is placed BEFORE the next block's loads are
    // issued (the block it waits for was requested a whole block evaluation earlier), never between them and their use
    auto landed = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_blocks = (c_end - c_begin) / CB;
    const float* const cb_last = cb + (int64_t)(n_blocks > 0 ? n_blocks - 1 : 0) * (CB * D);  // prefetches never run past the slice
    int c = c_begin;
    if (n_blocks > 0) load_block(blk_a, cb);
#pragma unroll 1
    for (int b = 0; b + 1 < n_blocks; b += 2) {
        const float* nb = cb + CB * D;
        landed();
        load_block(blk_b, nb);
        visit_block(blk_a, c);
        nb = nb + CB * D <= cb_last ? nb + CB * D : cb_last;
        landed();
        load_block(blk_a, nb);
        visit_block(blk_b, c + CB);
        cb += 2 * CB * D;
        c += 2 * CB;
    }
    if (n_blocks & 1) {  // blk_a holds the last block
        visit_block(blk_a, c);
        cb += CB * D;
        c += CB;
    }
    for (; c < c_end; ++c, cb += D) {
        float cc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cc[d] = cb[d];
        visit(cc, c);
    }
    // which code of the winning block: the first whose distance equals the minimum (codes past the slice do not take part)
#pragma unroll
    for (int j = 0; j < QPL; ++j) {
        const int64_t qi = q0 + 64 * j;
        if (qi < n) {
            int win = best_i[j];
            if (win != 0x7fffffff) {
                const int base = win;
                bool found = false;
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const int ci = base + i < c_end ? base + i : c_end - 1;
                    float cc[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) cc[d] = codebook[(int64_t)ci * D + d];
                    const bool hit = !found && base + i < c_end && dist_of(cc, j) == best[j];
                    win = hit ? base + i : win;
                    found = found || hit;
                }
            }
            part_dist[(int64_t)blockIdx.y * n + qi] = best[j];
            part_idx[(int64_t)blockIdx.y * n + qi] = win;
        }
    }
}

// `list` != null: the queries are list[0 .. *list_n - 1] (numbers into `queries`), results are stored per list SLOT and
// `n` is only the slot stride of the partial arrays: the exact pass of the screened form, whose length the host does not know.
template <int D, int QW>
__global__ __launch_bounds__(256) void vq_wave_kernel(const float* __restrict__ queries, int64_t n, const float* __restrict__ codebook,
                                                     int k, int k_per_part, float* __restrict__ part_dist, int32_t* __restrict__ part_idx,
                                                     const int32_t* __restrict__ list, const int32_t* __restrict__ list_n) {
    const int lane = threadIdx.x & 63;
    const int64_t n_here = list ? (int64_t)*list_n : n;
    // (a listed run's length is not known to the host: its grid is a fixed number of wave columns that stride over the list)
    for (int64_t qbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QW; qbase < n_here; qbase += (int64_t)gridDim.x * 4 * QW) {
    float q[QW][D];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int64_t slot = qbase + j < n_here ? qbase + j : n_here - 1;
        const int64_t qi = list ? (int64_t)list[slot] : slot;
#pragma unroll
        for (int d = 0; d < D; ++d) q[j][d] = queries[qi * D + d];  // uniform address: every lane holds the query
    }
    float best[QW];
    int best_i[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        best[j] = INFINITY;
        best_i[j] = 0x7fffffff;
    }
    const int c_begin = blockIdx.y * k_per_part;
    const int c_end = min(k, c_begin + k_per_part);
    auto visit = [&](const float (&cc)[D], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            float t = q[j][0] - cc[0];
            float dist = t * t;
#pragma unroll
            for (int d = 1; d < D; ++d) {
                t = q[j][d] - cc[d];
                dist = fmaf(t, t, dist);
            }
            const bool lt = dist < best[j];
            best[j] = lt ? dist : best[j];
            best_i[j] = lt ? c : best_i[j];
        }
    };
    // a lane's codes come in increasing index (strict '<' keeps its lowest); four of them are requested at a time: with few
    // waves per SIMD (a streaming chunk, the listed queries of the screened form) the loop runs at the speed of its loads
    constexpr int UC = 4;
    int c = c_begin + lane;
    for (; c + 64 * (UC - 1) < c_end; c += 64 * UC) {
        float cc[UC][D];
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int d = 0; d < D; ++d) cc[u][d] = codebook[(int64_t)(c + 64 * u) * D + d];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UC; ++u) visit(cc[u], c + 64 * u);
    }
    for (; c < c_end; c += 64) {
        float cc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cc[d] = codebook[(int64_t)c * D + d];
        visit(cc, c);
    }
    // wavefront-level argmin: (distance, index) pairs combined over the 64 lanes, the lower index winning equal distances
#pragma unroll
    for (int j = 0; j < QW; ++j) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float od = __shfl_xor(best[j], m, 64);
            const int oi = __shfl_xor(best_i[j], m, 64);
            const bool take = od < best[j] || (od == best[j] && oi < best_i[j]);
            best[j] = take ? od : best[j];
            best_i[j] = take ? oi : best_i[j];
        }
        if (lane == 0 && qbase + j < n_here) {
            part_dist[(int64_t)blockIdx.y * n + qbase + j] = best[j];
            part_idx[(int64_t)blockIdx.y * n + qbase + j] = best_i[j];
        }
    }
    }
}

__global__ __launch_bounds__(THREADS) void vq_argmin_combine_kernel(const float* __restrict__ part_dist,
                                                                   const int32_t* __restrict__ part_idx, int64_t n,
                                                                   int parts, int32_t* __restrict__ out_idx,
                                                                   const int32_t* __restrict__ list, const int32_t* __restrict__ list_n) {
    // 64 queries (list slots when there is a list) per block and round; the four waves take every fourth slice each — a single
    // chain of `parts` dependent-looking loads per query ran at one memory latency per slice — and wave 0 merges the four.
    // Slices are in increasing index order: among equal distances the lowest slice holds the lowest index.
    __shared__ float s_d[3][64];
    __shared__ int s_p[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n_here = list ? (int64_t)*list_n : n;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n_here; base += (int64_t)gridDim.x * 64) {  // block-uniform
        const int64_t qi = base + lane;
        const int64_t qc = qi < n_here ? qi : n_here - 1;
        float best = INFINITY;
        int best_p = 0x7fffffff;
#pragma unroll 4
        for (int p = w; p < parts; p += 4) {
            const float d = part_dist[(int64_t)p * n + qc];
            const bool lt = d < best;
            best = lt ? d : best;
            best_p = lt ? p : best_p;
        }
        if (w > 0) {
            s_d[w - 1][lane] = best;
            s_p[w - 1][lane] = best_p;
        }
        __syncthreads();
        if (w == 0 && qi < n_here) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float d = s_d[i][lane];
                const int pp = s_p[i][lane];
                const bool take = d < best || (d == best && pp < best_p);
                best = take ? d : best;
                best_p = take ? pp : best_p;
            }
            // (no slice below +inf: the first slice's entry, as a chain over the slices would have kept)
            out_idx[list ? (int64_t)list[qi] : qi] = part_idx[(int64_t)(best_p == 0x7fffffff ? 0 : best_p) * n + qi];
        }
        __syncthreads();
    }
}

// ---- screened form (many queries) ------------------------------------------------------------------
// The direct-form search above is the DEFINITION of the result; what costs 2 D + 1 vector instructions per (query, code)
// there is found here on the fp32 matrix pipe: score(q, c) = |c|^2 - 2 q.c (= dist - |q|^2) for a tile of 32 codes x 32
// queries is D / 2 v_mfma_f32_32x32x2_f32 (exact fp32 FMAs, the code norms enter as the accumulator's initial value), and
// the vector unit only keeps, per query, the smallest and second smallest BLOCK minimum (a block = the 16 codes of a tile one
// lane holds) and which block the smallest came from: 12 instructions per 16 pairs.  The scores are not the oracle's
// arithmetic, so they decide nothing by themselves: with u = 2^-24 and Qd = sum_d (|q_d| + max|c|)^2, the direct-form
// distance is within 11 u Qd and the score within 16 u Qd of the exact value, hence the direct-form winner's score is at most
// 2 (11 + 16) = 54 u Qd above the smallest score; the threshold used is 128 u Qd (it would still hold if the matrix pipe's
// accumulation were twice as inexact as an fmaf chain).  vq_resolve_kernel therefore
//   * second block minimum - smallest > 128 u Qd: the direct-form winner lies in the winning block: its (at most 16) codes are
//     evaluated with the direct form in index order, strict '<': the same bits and the same tie rule as the scan;
//   * otherwise (near-ties across blocks, exact duplicates, non-finite data): the query goes on a list, and the direct-form
//     wave kernel searches the WHOLE codebook for the listed queries.
// Either way every returned index is the direct form's, bit for bit (tests/test_gpu_blocks.py compares the two forms on
// duplicated codebooks); on the FSQ grids ~0.6 % of the queries take the second route.
template <int D>
__device__ __forceinline__ float vq_dist(const float (&q)[D], const float (&cc)[D]) {
    float t = q[0] - cc[0];
    float dist = t * t;
#pragma unroll
    for (int d = 1; d < D; ++d) {
        t = q[d] - cc[d];
        dist = fmaf(t, t, dist);
    }
    return dist;
}

constexpr int VQ_HDR_FLOATS = 64;  // scratch header: [0] listed-query counter (int), [1] max |c_d| (float bits, >= 0)

__global__ void vq_header_reset_kernel(int32_t* __restrict__ header) { header[threadIdx.x] = 0; }

// norms[c] = |c|^2 for c < k, +inf for the padding codes of the last tile; header[1] = max |c_d|
template <int D>
__global__ __launch_bounds__(THREADS) void vq_norms_kernel(const float* __restrict__ codebook, int k, int k_padded,
                                                          float* __restrict__ norms, unsigned* __restrict__ header) {
    const int c = blockIdx.x * THREADS + threadIdx.x;
    float cmax = 0.f;
    if (c < k_padded) {
        float nn = INFINITY;
        if (c < k) {
            float cc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                cc[d] = codebook[(int64_t)c * D + d];
                cmax = fmaxf(cmax, fabsf(cc[d]));
            }
            nn = cc[0] * cc[0];
#pragma unroll
            for (int d = 1; d < D; ++d) nn = fmaf(cc[d], cc[d], nn);
        }
        norms[c] = nn;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, m, 64));
    // non-negative floats order as their bits; the plain read only skips atomics that could not raise the maximum
    if ((threadIdx.x & 63) == 0 && __float_as_uint(cmax) > __builtin_nontemporal_load(header + 1)) atomicMax(header + 1, __float_as_uint(cmax));
}

constexpr int VQ_SCREEN_NQ = 8;  // query tiles (of 32) per wave: the code operand and the norms are loaded once for all of them
                                 // (8: 1.24 ms, 4: 1.28 ms at K = 250 047, N = 42 752)

template <int D, int NQ>
__global__ __launch_bounds__(256) void vq_screen_kernel(const float* __restrict__ queries, int64_t n, const float* __restrict__ codebook,
                                                       const float* __restrict__ norms, int k, int tiles_per_part,
                                                       float* __restrict__ part_b, float* __restrict__ part_sb,
                                                       int32_t* __restrict__ part_blk) {
    constexpr int KS = (D + 1) / 2;  // k = 2 per MFMA: lane (i = lane & 31, kk = lane >> 5) supplies element (i, kk) of both operands
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    const int64_t qw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (32 * NQ);
    if (qw >= n) return;  // wave-uniform
    float bq[NQ][KS];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int64_t qi = qw + 32 * j + col;
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[j][s] = (qi < n && 2 * s + h < D) ? -2.0f * queries[qi * D + 2 * s + h] : 0.f;
    }
    float b[NQ], sb[NQ];
    int blk[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        b[j] = INFINITY;
        sb[j] = INFINITY;
        blk[j] = 0;
    }
    const int n_tiles = (k + 31) >> 5;
    const int t_begin = blockIdx.y * tiles_per_part;
    const int t_end = min(n_tiles, t_begin + tiles_per_part);
    if (t_begin >= t_end) return;
    // The operands of tile t + 1 are requested before tile t is evaluated.  Rows past the codebook's end read its last row:
    // their norm is +inf, so is their score.  (Odd D: the missing coordinate of the last k-step is a zero on both sides.)
    float a_nx[KS];
    float4 cn_nx[4];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const float* row = codebook + (int64_t)min(t * 32 + col, k - 1) * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (2 * s + 1 < D) {
                a_nx[s] = row[2 * s + h];
            } else {  // odd D, last k-step: coordinate D - 1 for h = 0, nothing for h = 1
                a_nx[s] = row[2 * s];
                a_nx[s] = h ? 0.f : a_nx[s];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) cn_nx[g] = *reinterpret_cast<const float4*>(norms + t * 32 + 8 * g + 4 * h);
    };
    fetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        float a[KS];
        f32x16 cn;  // accumulator register r of lane (col, h) is code row 8 (r / 4) + 4 h + r % 4 of the tile
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = a_nx[s];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cn[4 * g + 0] = cn_nx[g].x; cn[4 * g + 1] = cn_nx[g].y; cn[4 * g + 2] = cn_nx[g].z; cn[4 * g + 3] = cn_nx[g].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(t + 1 < t_end ? t + 1 : t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bq[j][0], cn, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[j][s], acc, 0, 0, 0);
            // (minimumNumber is the hardware's v_min / v_min3 as it stands: fminf would first quiet every operand with a v_max)
            float m = vq_min(vq_min(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) m = vq_min(vq_min(m, acc[r]), acc[r + 1]);
            m = vq_min(m, acc[15]);
            sb[j] = __builtin_amdgcn_fmed3f(b[j], m, sb[j]);  // b <= sb always: the median is the new second smallest
            blk[j] = m < b[j] ? t : blk[j];
            b[j] = vq_min(b[j], m);
        }
    }
    // the two half-waves hold the two halves of every tile for the same 32 queries
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float ob = __shfl_xor(b[j], 32, 64);
        const float osb = __shfl_xor(sb[j], 32, 64);
        const int oblk = __shfl_xor(blk[j], 32, 64);
        const bool mine = b[j] < ob || (b[j] == ob && h == 0);
        const float nb = __builtin_fminf(b[j], ob);
        const float nsb = __builtin_fminf(__builtin_fmaxf(b[j], ob), __builtin_fminf(sb[j], osb));
        const int nblk = mine ? 2 * blk[j] + h : 2 * oblk + (1 - h);
        const int64_t qi = qw + 32 * j + col;
        if (h == 0 && qi < n) {
            part_b[(int64_t)blockIdx.y * n + qi] = nb;
            part_sb[(int64_t)blockIdx.y * n + qi] = nsb;
            part_blk[(int64_t)blockIdx.y * n + qi] = nblk;
        }
    }
}

template <int D>
__global__ __launch_bounds__(THREADS) void vq_resolve_kernel(const float* __restrict__ queries, int64_t n, const float* __restrict__ codebook,
                                                            int k, int parts, const float* __restrict__ part_b,
                                                            const float* __restrict__ part_sb, const int32_t* __restrict__ part_blk,
                                                            int32_t* __restrict__ header, int32_t* __restrict__ list,
                                                            int32_t* __restrict__ out_idx) {
    // 64 queries per block; the four waves merge every fourth slice each, wave 0 merges the four results and decides
    __shared__ float s_b[3][64], s_sb[3][64];
    __shared__ int s_blk[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 64 + lane;
    const int64_t qc = qi < n ? qi : n - 1;
    float b = INFINITY, sb = INFINITY;
    int blk = 0;
    auto merge = [&](float pb, float psb, int pblk) __attribute__((always_inline)) {
        sb = fminf(fmaxf(b, pb), fminf(sb, psb));
        blk = pb < b ? pblk : blk;
        b = fminf(b, pb);
    };
    for (int p = w; p < parts; p += 4) merge(part_b[(int64_t)p * n + qc], part_sb[(int64_t)p * n + qc], part_blk[(int64_t)p * n + qc]);
    if (w > 0) {
        s_b[w - 1][lane] = b;
        s_sb[w - 1][lane] = sb;
        s_blk[w - 1][lane] = blk;
    }
    __syncthreads();
    if (w > 0 || qi >= n) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) merge(s_b[i][lane], s_sb[i][lane], s_blk[i][lane]);
    float q[D];
    const float cmax = __int_as_float(header[1]);
    float qd = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q[d] = queries[qi * D + d];
        const float e = fabsf(q[d]) + cmax;
        qd = fmaf(e, e, qd);
    }
    const float tol = fmaf(0x1p-17f, qd, 1e-35f);  // 128 u Qd
    if (!(sb - b > tol)) {  // also every non-finite case
        list[atomicAdd(header, 1)] = (int32_t)qi;
        return;
    }
    const int base = (blk >> 1) * 32 + 4 * (blk & 1);
    float best = INFINITY;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // increasing index
        const int c = base + 8 * (r >> 2) + (r & 3);
        const int cl = c < k ? c : k - 1;
        float cc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cc[d] = codebook[(int64_t)cl * D + d];
        const float dist = vq_dist<D>(q, cc);
        const bool lt = c < k && dist < best;
        best = lt ? dist : best;
        best_i = lt ? c : best_i;
    }
    out_idx[qi] = best_i;
}

constexpr int VQ_QPL = 4;       // queries per lane of the scan form
constexpr int VQ_QW = 4;        // queries per wave of the wavefront form
constexpr int64_t VQ_WAVE_MAX_N = 5120;  // below this many queries the wavefront form fills the chip better (measured crossover at K = 117 649: 4096-8192)

enum VqForm { VQ_WAVE, VQ_SCAN, VQ_SCREEN };

struct VqPlan {
    VqForm form;
    int parts;           // codebook index ranges of the first kernel
    int k_per_part;      // (screened form: a multiple of 32)
    int parts_exact;     // screened form: index ranges of the exact pass over the listed queries
    int k_padded;        // screened form: k rounded up to whole tiles
    size_t off_norms, off_list, off_a, off_b, off_c, off_xd, off_xi, bytes;  // scratch layout (bytes)
};

constexpr int VQ_SCREEN_ROUNDS = 8;
constexpr int VQ_SCREEN_WGS_PER_CU = 3;  // <= 168 registers per lane
constexpr int VQ_SCREEN_LDS = (160 * 1024 / VQ_SCREEN_WGS_PER_CU) & ~255;  // requested, not used: pins that many workgroups per CU


VqForm vq_form(int64_t n, int form) {  // form 1: the direct-form scan wherever the screened form would run
    if (n < VQ_WAVE_MAX_N) return VQ_WAVE;
    return form == 1 ? VQ_SCAN : VQ_SCREEN;
}

// How the codebook is cut: enough (query group, slice) work items for ~8 waves per SIMD at the scan form, ~4 at the wave form,
// ~16 rounds of waves at the screened form
VqPlan vq_plan(int64_t n, int k, int form) {
    VqPlan pl{};
    pl.form = vq_form(n, form);
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if (pl.form == VQ_SCREEN) {
        // Work items (4 waves = 4 * 32 NQ queries, one index range) are equal in cost, so their number is made to FILL a whole
        // number of rounds of the chip (VQ_SCREEN_WGS_PER_CU workgroups per CU run at a time: the launch pins that with its LDS
        // request): 4 200 items on 1 024 places take five rounds, 4 032 take four.
        const int n_tiles = (k + 31) / 32;
        const int64_t groups_q = ceil_div64(n, 4 * 32 * VQ_SCREEN_NQ);
        const int64_t places = (int64_t)l3ac_device_cu_count() * VQ_SCREEN_WGS_PER_CU;
        int64_t parts = (VQ_SCREEN_ROUNDS * places) / groups_q;  // rounded down: at most ROUNDS full rounds
        const int64_t max_parts = ceil_div64(n_tiles, 16);
        if (parts > max_parts) parts = max_parts;
        if (parts < 1) parts = 1;
        const int tiles_per_part = (int)ceil_div64(n_tiles, parts);
        pl.parts = (int)ceil_div64(n_tiles, tiles_per_part);
        pl.k_per_part = tiles_per_part * 32;
        pl.k_padded = n_tiles * 32;
        // the exact pass runs after the partial minima have been consumed and reuses their 12 n parts bytes (8 n per range)
        int64_t pe = ceil_div64(k, 512);
        const int64_t pe_max = (int64_t)pl.parts * 3 / 2;
        if (pe > pe_max) pe = pe_max;
        if (pe > 256) pe = 256;
        if (pe < 1) pe = 1;
        pl.parts_exact = (int)pe;
        size_t off = align(VQ_HDR_FLOATS * 4);
        pl.off_norms = off; off = align(off + (size_t)pl.k_padded * 4);
        pl.off_list = off;  off = align(off + (size_t)n * 4);
        const size_t third = align((size_t)pl.parts * n * 4);
        size_t exact = align((size_t)pl.parts_exact * n * 4);
        pl.off_a = off;
        pl.off_b = off + third;
        pl.off_c = off + 2 * third;
        pl.off_xd = off;
        pl.off_xi = off + exact;
        off += (3 * third > 2 * exact ? 3 * third : 2 * exact);
        pl.bytes = off;
        return pl;
    }
    const bool wave_form = pl.form == VQ_WAVE;
    const int64_t waves_q = wave_form ? ceil_div64(n, VQ_QW) : ceil_div64(n, 64 * VQ_QPL);
    int64_t parts = ceil_div64(wave_form ? 4096 : 8192, waves_q);
    const int64_t max_parts = ceil_div64(k, wave_form ? 1024 : 512);  // a slice should still amortise its prologue
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    pl.parts = (int)parts;
    pl.k_per_part = (int)ceil_div64(k, parts);
    pl.off_a = 0;
    pl.off_b = (size_t)pl.parts * n * 4;
    pl.bytes = (size_t)pl.parts * n * 8;
    return pl;
}

template <int D>
int launch_vq_t(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, const VqPlan& pl, char* scratch,
                int32_t* out_idx) {
    const double flop = 3.0 * D * (double)n * k, bytes = 4.0 * ((double)n * D + (double)k * D + n);
    if (pl.form != VQ_SCREEN) {
        float* part_dist = reinterpret_cast<float*>(scratch + pl.off_a);
        int32_t* part_idx = reinterpret_cast<int32_t*>(scratch + pl.off_b);
        {
            ProfScope prof(s, pl.form == VQ_WAVE ? "vq_wave_kernel" : "vq_scan_kernel", flop, bytes);
            if (pl.form == VQ_WAVE) {
                const dim3 grid((unsigned)ceil_div64(n, 4 * VQ_QW), (unsigned)pl.parts);
                hipLaunchKernelGGL((vq_wave_kernel<D, VQ_QW>), grid, dim3(256), 0, s, queries, n, codebook, k, pl.k_per_part, part_dist,
                                   part_idx, (const int32_t*)nullptr, (const int32_t*)nullptr);
            } else {
                const dim3 grid((unsigned)ceil_div64(n, 4 * 64 * VQ_QPL), (unsigned)pl.parts);
                hipLaunchKernelGGL((vq_scan_kernel<D, VQ_QPL>), grid, dim3(256), 0, s, queries, n, codebook, k, pl.k_per_part, part_dist,
                                   part_idx);
            }
            L3AC_LAUNCH_CHECK();
        }
        const int64_t cblocks = ceil_div64(n, 64);
        hipLaunchKernelGGL(vq_argmin_combine_kernel, dim3((unsigned)(cblocks < 4096 ? cblocks : 4096)), dim3(THREADS), 0, s, part_dist, part_idx,
                           n, pl.parts, out_idx, (const int32_t*)nullptr, (const int32_t*)nullptr);
        L3AC_LAUNCH_CHECK();
        return L3AC_OK;
    }
    int32_t* header = reinterpret_cast<int32_t*>(scratch);
    float* norms = reinterpret_cast<float*>(scratch + pl.off_norms);
    int32_t* list = reinterpret_cast<int32_t*>(scratch + pl.off_list);
    float* part_b = reinterpret_cast<float*>(scratch + pl.off_a);
    float* part_sb = reinterpret_cast<float*>(scratch + pl.off_b);
    int32_t* part_blk = reinterpret_cast<int32_t*>(scratch + pl.off_c);
    float* x_dist = reinterpret_cast<float*>(scratch + pl.off_xd);
    int32_t* x_idx = reinterpret_cast<int32_t*>(scratch + pl.off_xi);
    // (a kernel, not hipMemsetAsync: captured into a hipGraph the memset node ran on the first replay only — the counter kept
    // growing, the list overflowed: tests/test_gpu_blocks.py::test_vq_argmin_graph_capture)
    hipLaunchKernelGGL(vq_header_reset_kernel, dim3(1), dim3(VQ_HDR_FLOATS), 0, s, header);
    L3AC_LAUNCH_CHECK();
    {
        ProfScope prof(s, "vq_norms_kernel", 2.0 * D * k, 4.0 * (D + 1) * k);
        hipLaunchKernelGGL((vq_norms_kernel<D>), dim3((unsigned)ceil_div64(pl.k_padded, THREADS)), dim3(THREADS), 0, s, codebook, k,
                           pl.k_padded, norms, reinterpret_cast<unsigned*>(header));
        L3AC_LAUNCH_CHECK();
    }
    {
        ProfScope prof(s, "vq_screen_kernel", flop, bytes);
        const dim3 grid((unsigned)ceil_div64(n, 4 * 32 * VQ_SCREEN_NQ), (unsigned)pl.parts);
        hipLaunchKernelGGL((vq_screen_kernel<D, VQ_SCREEN_NQ>), grid, dim3(256), VQ_SCREEN_LDS, s, queries, n, codebook, norms, k,
                           pl.k_per_part / 32, part_b, part_sb, part_blk);
        L3AC_LAUNCH_CHECK();
    }
    {
        ProfScope prof(s, "vq_resolve_kernel", 3.0 * D * 16 * (double)n, 12.0 * pl.parts * (double)n);
        hipLaunchKernelGGL((vq_resolve_kernel<D>), dim3((unsigned)ceil_div64(n, 64)), dim3(THREADS), 0, s, queries, n, codebook, k,
                           pl.parts, part_b, part_sb, part_blk, header, list, out_idx);
        L3AC_LAUNCH_CHECK();
    }
    {   // the listed queries, direct form over the whole codebook (blocks past the list's end return at once)
        ProfScope prof(s, "vq_wave_kernel", 0.0, 0.0);
        const int kpp = (int)ceil_div64(k, pl.parts_exact);
        const int64_t cols = ceil_div64(n, 4 * VQ_QW);
        const dim3 grid((unsigned)(cols < 64 ? cols : 64), (unsigned)pl.parts_exact);
        hipLaunchKernelGGL((vq_wave_kernel<D, VQ_QW>), grid, dim3(256), 0, s, queries, n, codebook, k, kpp, x_dist, x_idx,
                           (const int32_t*)list, (const int32_t*)header);
        L3AC_LAUNCH_CHECK();
        const int64_t cblocks = ceil_div64(n, 64);
        hipLaunchKernelGGL(vq_argmin_combine_kernel, dim3((unsigned)(cblocks < 64 ? cblocks : 64)), dim3(THREADS), 0, s, x_dist, x_idx, n,
                           pl.parts_exact, out_idx, (const int32_t*)list, (const int32_t*)header);
        L3AC_LAUNCH_CHECK();
    }
    return L3AC_OK;
}

}  // namespace

int launch_fsq(hipStream_t s, const FsqArgs& a) {
    L3AC_REQUIRE(a.n >= 0 && a.feat >= 8 && a.feat % 8 == 0 && (a.feat / 8 & (a.feat / 8 - 1)) == 0 && a.feat <= 512,
                 "fsq: feature_dim=%d must be 8 * 2^j, at most 512", a.feat);
    L3AC_REQUIRE(a.n_levels >= 1 && a.n_levels <= MAXD, "fsq: n_levels=%d out of range", a.n_levels);
    L3AC_REQUIRE(a.w_out && a.b_out && a.q_feature, "fsq: null output projection / output");
    L3AC_REQUIRE(a.idx_in || a.x || a.latents, "fsq: no input");
    if (a.x) L3AC_REQUIRE(a.w_in != nullptr, "fsq: missing project_in");
    FsqDev p{};
    p.x = a.x; p.n = a.n; p.feat = a.feat; p.d = a.n_levels;
    int64_t basis = 1;
    for (int d = 0; d < a.n_levels; ++d) {
        L3AC_REQUIRE(a.levels[d] >= 2, "fsq: level %d < 2", a.levels[d]);
        p.levels[d] = a.levels[d];
        p.basis[d] = (int)basis;
        basis *= a.levels[d];
    }
    L3AC_REQUIRE(basis < (1 << 24), "fsq: codebook size %lld exceeds the exact fp32 index range", (long long)basis);
    p.w_in = a.w_in; p.b_in = a.b_in; p.w_out = a.w_out; p.b_out = a.b_out; p.idx_in = a.idx_in;
    p.q_feature = a.q_feature; p.indices = a.indices; p.level_indices = a.level_indices; p.latents = a.latents;
    p.act_in = a.act_in ? 1 : 0;
    p.k_total = (int)basis;
    p.bad_count = a.bad_count;
    if (a.act_in) L3AC_REQUIRE(!a.x && !a.idx_in && a.latents, "fsq: act_in needs the activation values in `latents` and no other input");
    switch (a.n_levels) {
        case 1: return launch_fsq_t<1>(s, p);
        case 2: return launch_fsq_t<2>(s, p);
        case 3: return launch_fsq_t<3>(s, p);
        case 4: return launch_fsq_t<4>(s, p);
        case 5: return launch_fsq_t<5>(s, p);
        case 6: return launch_fsq_t<6>(s, p);
        case 7: return launch_fsq_t<7>(s, p);
        default: return launch_fsq_t<8>(s, p);
    }
}

int launch_fsq_copy_ceiling(hipStream_t s, const float* x, int64_t n, float* q, int32_t* idx, float* li, int blocks_per_cu) {
    L3AC_REQUIRE(x && q && idx && li && n > 0 && blocks_per_cu >= 0 && blocks_per_cu <= 8, "fsq_copy_ceiling: bad arguments");
    int64_t blocks = ceil_div64(n, THREADS / 8);
    // blocks_per_cu = 0: the quantiser kernel's own residency (register-bound) — the ceiling of ITS launch shape; otherwise the given
    // residency (the copy kernel needs neither LDS nor many registers: its own best residency is higher, and so is its rate)
    const int per_cu = blocks_per_cu > 0 ? blocks_per_cu
                                         : fsq_resident(reinterpret_cast<const void*>(fsq_forward128_kernel<6>), (size_t)(2 * 6 + 1) * 128 * sizeof(float), 2);
    const int64_t places = (int64_t)l3ac_device_cu_count() * per_cu;
    if (blocks > places) blocks = places;
    ProfScope prof(s, "fsq_copy_ceiling_kernel", 0.0, (double)n * 1052.0);
    hipLaunchKernelGGL(fsq_copy_ceiling_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, s, x, n, q, idx, li);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

size_t vq_argmin_scratch_bytes(int64_t n, int k, int form) { return vq_plan(n, k, form).bytes; }

// scratch = vq_argmin_scratch_bytes(n, k) bytes
int launch_vq_argmin(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, int dim, void* scratch,
                     int32_t* out_idx, int form) {
    L3AC_REQUIRE(dim >= 1 && dim <= 8 && k > 0 && n >= 0, "vq_argmin: bad shape (dim=%d k=%d)", dim, k);
    L3AC_REQUIRE(n < ((int64_t)1 << 31), "vq_argmin: %lld queries exceed the 32-bit query numbers of one call", (long long)n);
    if (n == 0) return L3AC_OK;
    const VqPlan pl = vq_plan(n, k, form);
    char* sc = reinterpret_cast<char*>(scratch);
    switch (dim) {
        case 1: return launch_vq_t<1>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 2: return launch_vq_t<2>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 3: return launch_vq_t<3>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 4: return launch_vq_t<4>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 5: return launch_vq_t<5>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 6: return launch_vq_t<6>(s, queries, n, codebook, k, pl, sc, out_idx);
        case 7: return launch_vq_t<7>(s, queries, n, codebook, k, pl, sc, out_idx);
        default: return launch_vq_t<8>(s, queries, n, codebook, k, pl, sc, out_idx);
    }
}
