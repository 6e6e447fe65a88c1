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
// vq_scan_kernel / vq_wave_kernel — the explicit-codebook nearest-neighbour search FSQ is the closed form of:
// see "explicit codebook search" below; lowest index wins exact ties.
#include "../kernels.hpp"

namespace {

constexpr int THREADS = 256;
constexpr int MAXD = L3AC_MAX_LEVELS;

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

template <int D, int NV>  // NV 16-byte chunks (4 channels each) per lane: lanes per token = feat / (4 NV)
__global__ __launch_bounds__(THREADS) void fsq_kernel(const FsqDev p, const int lpt) {
    // HBM-bound by construction (1 052 B per token); what limits it in practice is bytes in flight and issue slots,
    // so: projection weights in LDS (not registers), the next token group's rows prefetched into registers, each
    // block walking a contiguous token range, and ONE tanh per lane (lane d of a token's group quantises latent d
    // and the level indices are exchanged by shuffles) instead of D redundant ones.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Win = smem;                  // [D][feat]
    float* Wout = smem + D * p.feat;    // [D][feat] (project_out transposed)
    float* Bout = Wout + D * p.feat;    // [feat]
    const int tid = threadIdx.x;
    for (int i = tid; i < D * p.feat; i += THREADS) {
        const int d = i / p.feat, c = i % p.feat;
        Win[i] = p.w_in ? p.w_in[i] : 0.f;
        Wout[i] = p.w_out[c * D + d];
    }
    for (int i = tid; i < p.feat; i += THREADS) Bout[i] = p.b_out[i];
    __syncthreads();

    const int sub = tid % lpt;          // lane within the token's group
    const int lane = tid & 63;
    // channel quads are dealt round-robin over the lanes of a token: quad v of lane `sub` = channels 4 (v lpt + sub) .. +3,
    // so every load / store instruction of a wave covers whole 128-B lines (8 lanes x 16 B contiguous per token)
    const int cq = 4 * sub, cstep = 4 * lpt;
    const int tok_per_block = THREADS / lpt;
    const int64_t n_groups = (p.n + tok_per_block - 1) / tok_per_block;
    const int64_t per_block = (n_groups + gridDim.x - 1) / gridDim.x;  // contiguous token range per block
    const int64_t g_begin = (int64_t)blockIdx.x * per_block;
    const int64_t g_end = g_begin + per_block < n_groups ? g_begin + per_block : n_groups;
    const bool spread = lpt >= D;       // one latent per lane; otherwise every lane quantises all D

    float4 x_next[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) x_next[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t g) {
        const int64_t t = g * tok_per_block + tid / lpt;
        if (p.x && g < g_end && t < p.n) {
#pragma unroll
            for (int v = 0; v < NV; ++v) x_next[v] = *reinterpret_cast<const float4*>(p.x + t * p.feat + cq + cstep * v);
        }
    };
    auto quantise = [&](float lat, int d) -> float {
        const float act = p.act_in ? lat : (tanhf(lat) + 1.0f) * 0.5f;  // fsq_act.py:39
        return rintf(__fmul_rn(act, (float)(p.levels[d] - 1)));       // vq/fsq.py:59 (half-to-even)
    };
    fetch(g_begin);
    for (int64_t g = g_begin; g < g_end; ++g) {
        const int64_t tok = g * tok_per_block + tid / lpt;
        const bool ok = tok < p.n;
        const int64_t tk = ok ? tok : 0;
        float4 xv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[v] = x_next[v];
        fetch(g + 1);
        float li[D];
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
                        const float4 wv = *reinterpret_cast<const float4*>(Win + d * p.feat + cq + cstep * v);
                        s = fmaf(xv[v].x, wv.x, s); s = fmaf(xv[v].y, wv.y, s);
                        s = fmaf(xv[v].z, wv.z, s); s = fmaf(xv[v].w, wv.w, s);
                    }
                    for (int mask = lpt >> 1; mask > 0; mask >>= 1) s += __shfl_xor(s, mask, 64);
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
                const float li_mine = quantise(mine, sub < D ? sub : 0);
                const int group_base = lane - sub;
#pragma unroll
                for (int d = 0; d < D; ++d) li[d] = __shfl(li_mine, group_base + d, 64);
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
                const float4 wv = *reinterpret_cast<const float4*>(Wout + d * p.feat + cq + cstep * v);
                o[v].x = fmaf(q, wv.x, o[v].x); o[v].y = fmaf(q, wv.y, o[v].y);
                o[v].z = fmaf(q, wv.z, o[v].z); o[v].w = fmaf(q, wv.w, o[v].w);
            }
        }
        if (ok) {
            float* dst = p.q_feature + tok * p.feat + cq;
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<float4*>(dst + cstep * v) = o[v];
            if (sub == 0) {
                if (p.indices) p.indices[tok] = (int32_t)idx_f;
                if (p.level_indices) {
#pragma unroll
                    for (int d = 0; d < D; ++d) p.level_indices[tok * D + d] = li[d];
                }
            }
        }
    }
}

template <int D>
int launch_fsq_t(hipStream_t s, const FsqDev& p) {
    const int nv = p.feat >= 128 ? 4 : 2;        // 16 or 8 channels per lane
    const int lpt = p.feat / (4 * nv);
    const int tok_per_block = THREADS / lpt;
    int64_t blocks = ceil_div64(p.n, tok_per_block);
    if (blocks <= 0) return L3AC_OK;
    if (blocks > 256 * 8) blocks = 256 * 8;  // one weight staging per block, contiguous token range each
    const size_t lds = (size_t)(2 * D + 1) * p.feat * sizeof(float);
    const double in_b = p.x ? 4.0 * p.feat : (p.idx_in ? 4.0 : 4.0 * D);
    ProfScope prof(s, "fsq_kernel", 4.0 * D * p.feat * (double)p.n,
                   (double)p.n * (in_b + 4.0 * p.feat + (p.indices ? 4.0 : 0.0) + (p.level_indices ? 4.0 * D : 0.0)));
    if (nv == 4)
        hipLaunchKernelGGL((fsq_kernel<D, 4>), dim3((unsigned)blocks), dim3(THREADS), lds, s, p, lpt);
    else
        hipLaunchKernelGGL((fsq_kernel<D, 2>), dim3((unsigned)blocks), dim3(THREADS), lds, s, p, lpt);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

// ---- explicit codebook search ----------------------------------------------------------------------
// Brute-force L2 nearest neighbour, dist = sum_d (q_d - c_d)^2 accumulated with fmaf in dimension order, strict '<' while
// the codes are visited in increasing index (the lowest index wins an exact tie): 3 D N K algorithmic FLOP, fp32-VALU-bound.
// Two forms, both specialised on the true dimension (no padded FMAs) and both splitting the codebook into `parts` index
// ranges whose (distance, index) minima a small second kernel combines in index order:
//   vq_scan_kernel  (many queries)  QPL queries per lane in registers; the code coordinates are wave-uniform, so they come
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

template <int D, int QW>
__global__ __launch_bounds__(256) void vq_wave_kernel(const float* __restrict__ queries, int64_t n, const float* __restrict__ codebook,
                                                     int k, int k_per_part, float* __restrict__ part_dist, int32_t* __restrict__ part_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t qbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QW;  // wave-uniform
    if (qbase >= n) return;
    float q[QW][D];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int64_t qi = qbase + j < n ? qbase + j : n - 1;
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
    for (int c = c_begin + lane; c < c_end; c += 64) {  // a lane's codes come in increasing index: strict '<' keeps its lowest
        float cc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cc[d] = codebook[(int64_t)c * D + d];
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
        if (lane == 0 && qbase + j < n) {
            part_dist[(int64_t)blockIdx.y * n + qbase + j] = best[j];
            part_idx[(int64_t)blockIdx.y * n + qbase + j] = best_i[j];
        }
    }
}

__global__ __launch_bounds__(THREADS) void vq_argmin_combine_kernel(const float* __restrict__ part_dist,
                                                                   const int32_t* __restrict__ part_idx, int64_t n,
                                                                   int parts, int32_t* __restrict__ out_idx) {
    const int64_t qi = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (qi >= n) return;
    float best = part_dist[qi];
    int best_i = part_idx[qi];
    for (int p = 1; p < parts; ++p) {  // parts are in increasing index order: strict '<' keeps the lowest index
        const float d = part_dist[(int64_t)p * n + qi];
        if (d < best) {
            best = d;
            best_i = part_idx[(int64_t)p * n + qi];
        }
    }
    out_idx[qi] = best_i;
}

constexpr int VQ_QPL = 4;       // queries per lane of the scan form
constexpr int VQ_QW = 4;        // queries per wave of the wavefront form
constexpr int64_t VQ_WAVE_MAX_N = 5120;  // below this many queries the wavefront form fills the chip better (measured crossover at K = 117 649: 4096-8192)

template <int D>
int launch_vq_t(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, int parts, int k_per_part,
                float* part_dist, int32_t* part_idx, bool wave_form) {
    if (wave_form) {
        const dim3 grid((unsigned)ceil_div64(n, 4 * VQ_QW), (unsigned)parts);
        hipLaunchKernelGGL((vq_wave_kernel<D, VQ_QW>), grid, dim3(256), 0, s, queries, n, codebook, k, k_per_part, part_dist, part_idx);
    } else {
        const dim3 grid((unsigned)ceil_div64(n, 4 * 64 * VQ_QPL), (unsigned)parts);
        hipLaunchKernelGGL((vq_scan_kernel<D, VQ_QPL>), grid, dim3(256), 0, s, queries, n, codebook, k, k_per_part, part_dist, part_idx);
    }
    L3AC_LAUNCH_CHECK();
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

// How the codebook is cut: enough (query group, slice) work items for ~8 waves per SIMD at the scan form, ~4 at the wave form
int vq_argmin_parts(int64_t n, int k) {
    const bool wave_form = n < VQ_WAVE_MAX_N;
    const int64_t waves_q = wave_form ? ceil_div64(n, VQ_QW) : ceil_div64(n, 64 * VQ_QPL);
    int64_t parts = ceil_div64(wave_form ? 4096 : 8192, waves_q);
    const int64_t max_parts = ceil_div64(k, wave_form ? 1024 : 512);  // a slice should still amortise its prologue
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    return (int)parts;
}
size_t vq_argmin_scratch_bytes(int64_t n, int k) { return (size_t)vq_argmin_parts(n, k) * (size_t)n * 8; }

// scratch = vq_argmin_scratch_bytes(n, k) bytes (partial distances, then partial indices)
int launch_vq_argmin(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, int dim, void* scratch,
                     int32_t* out_idx) {
    L3AC_REQUIRE(dim >= 1 && dim <= 8 && k > 0 && n >= 0, "vq_argmin: bad shape (dim=%d k=%d)", dim, k);
    if (n == 0) return L3AC_OK;
    const int parts = vq_argmin_parts(n, k);
    const int k_per_part = (int)ceil_div64(k, parts);
    float* part_dist = reinterpret_cast<float*>(scratch);
    int32_t* part_idx = reinterpret_cast<int32_t*>(part_dist + (size_t)parts * n);
    const bool wave_form = n < VQ_WAVE_MAX_N;
    {
        ProfScope prof(s, wave_form ? "vq_wave_kernel" : "vq_scan_kernel", 3.0 * dim * (double)n * k, 4.0 * ((double)n * dim + (double)k * dim + n));
        switch (dim) {
            case 1: L3AC_TRY(launch_vq_t<1>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 2: L3AC_TRY(launch_vq_t<2>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 3: L3AC_TRY(launch_vq_t<3>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 4: L3AC_TRY(launch_vq_t<4>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 5: L3AC_TRY(launch_vq_t<5>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 6: L3AC_TRY(launch_vq_t<6>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            case 7: L3AC_TRY(launch_vq_t<7>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
            default: L3AC_TRY(launch_vq_t<8>(s, queries, n, codebook, k, parts, k_per_part, part_dist, part_idx, wave_form)); break;
        }
    }
    hipLaunchKernelGGL(vq_argmin_combine_kernel, dim3((unsigned)ceil_div64(n, THREADS)), dim3(THREADS), 0, s, part_dist,
                       part_idx, n, parts, out_idx);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
