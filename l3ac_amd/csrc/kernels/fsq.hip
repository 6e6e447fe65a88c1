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
// vq_argmin_kernel — the explicit-codebook nearest-neighbour search FSQ is the closed form of: one
// query per lane in registers, codebook shards staged through LDS and read as wave-wide broadcasts, partial
// (distance, index) minima per shard combined by a second pass; lowest index wins exact ties.
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
};

template <int D>
__global__ __launch_bounds__(THREADS) void fsq_kernel(const FsqDev p, const int lpt) {
    const int tid = threadIdx.x;
    const int sub = tid % lpt;       // lane within the token's group
    const int c0 = sub * 8;          // this lane's 8 channels
    const int tok_per_block = THREADS / lpt;

    float w_in[D][8], w_out[8][D], b_out[8], b_in[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        b_in[d] = p.b_in ? p.b_in[d] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) w_in[d][e] = p.w_in ? p.w_in[d * p.feat + c0 + e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        b_out[e] = p.b_out[c0 + e];
#pragma unroll
        for (int d = 0; d < D; ++d) w_out[e][d] = p.w_out[(c0 + e) * D + d];
    }

    const int64_t n_groups = (p.n + tok_per_block - 1) / tok_per_block;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t tok = g * tok_per_block + tid / lpt;
        const bool ok = tok < p.n;
        const int64_t tk = ok ? tok : 0;
        float li[D];
        if (p.idx_in) {  // decode: indices -> level indices (vq/fsq.py:70-71)
            const int idx = p.idx_in[tk];
#pragma unroll
            for (int d = 0; d < D; ++d) li[d] = (float)((idx / p.basis[d]) % p.levels[d]);
        } else {
            float lat[D];
            if (p.x) {
                const float4 xa = *reinterpret_cast<const float4*>(p.x + tk * p.feat + c0);
                const float4 xb = *reinterpret_cast<const float4*>(p.x + tk * p.feat + c0 + 4);
                const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) s = fmaf(xv[e], w_in[d][e], s);
                    for (int mask = lpt >> 1; mask > 0; mask >>= 1) s += __shfl_xor(s, mask, 64);
                    lat[d] = s + b_in[d];
                }
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) lat[d] = p.latents[tk * D + d];
            }
            if (p.x && p.latents && ok && sub == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) p.latents[tok * D + d] = lat[d];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float act = (tanhf(lat[d]) + 1.0f) * 0.5f;                // fsq_act.py:39
                li[d] = rintf(__fmul_rn(act, (float)(p.levels[d] - 1)));        // vq/fsq.py:59
            }
        }
        float q[D];
        float idx_f = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            idx_f += li[d] * (float)p.basis[d];                                 // exact (vq/fsq.py:67-68)
            const float q_act = __fdiv_rn(li[d], (float)(p.levels[d] - 1));     // vq/fsq.py:60
            q[d] = __fsub_rn(__fmul_rn(q_act, 2.0f), 1.0f);                     // vq/fsq.py:21
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = b_out[e];
#pragma unroll
            for (int d = 0; d < D; ++d) s = fmaf(q[d], w_out[e][d], s);
            o[e] = s;
        }
        if (ok) {
            float* dst = p.q_feature + tok * p.feat + c0;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
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
    const int lpt = p.feat / 8;
    const int tok_per_block = THREADS / lpt;
    int64_t blocks = ceil_div64(p.n, tok_per_block);
    if (blocks <= 0) return L3AC_OK;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride: weights stay in registers across tokens
    const double in_b = p.x ? 4.0 * p.feat : (p.idx_in ? 4.0 : 4.0 * D);
    ProfScope prof(s, "fsq_kernel", 4.0 * D * p.feat * (double)p.n,
                   (double)p.n * (in_b + 4.0 * p.feat + (p.indices ? 4.0 : 0.0) + (p.level_indices ? 4.0 * D : 0.0)));
    hipLaunchKernelGGL((fsq_kernel<D>), dim3((unsigned)blocks), dim3(THREADS), 0, s, p, lpt);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

// ---- explicit codebook search ----------------------------------------------------------------------
constexpr int SHARD = 2048;  // codes per LDS shard (2048 * 8 floats = 64 KiB)

template <int D>  // D = dim padded to 4 or 8 (padding coordinates are 0 on both sides)
__global__ __launch_bounds__(THREADS) void vq_argmin_kernel(const float* __restrict__ queries, int64_t n, int dim,
                                                           const float* __restrict__ codebook, int k, int shards_per_block,
                                                           float* __restrict__ part_dist, int32_t* __restrict__ part_idx) {
    __shared__ __attribute__((aligned(16))) float cs[SHARD * 8];
    const int64_t qi = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const bool ok = qi < n;
    float q[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = (ok && d < dim) ? queries[qi * dim + d] : 0.f;
    float best = INFINITY;
    int best_i = 0x7fffffff;
    const int k_begin = blockIdx.y * shards_per_block * SHARD;
    const int k_end = min(k, k_begin + shards_per_block * SHARD);
    for (int s0 = k_begin; s0 < k_end; s0 += SHARD) {
        const int cnt = min(SHARD, k_end - s0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 8; i += THREADS) {
            const int code = i >> 3, d = i & 7;
            cs[i] = d < dim ? codebook[(int64_t)(s0 + code) * dim + d] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < cnt; ++c) {
            const float4 ca = *reinterpret_cast<const float4*>(cs + c * 8);
            const float4 cb = *reinterpret_cast<const float4*>(cs + c * 8 + 4);
            float dist = 0.f, t;
            t = q[0] - ca.x; dist = fmaf(t, t, dist);
            t = q[1] - ca.y; dist = fmaf(t, t, dist);
            t = q[2] - ca.z; dist = fmaf(t, t, dist);
            t = q[3] - ca.w; dist = fmaf(t, t, dist);
            if (D > 4) {
                t = q[4] - cb.x; dist = fmaf(t, t, dist);
                t = q[5] - cb.y; dist = fmaf(t, t, dist);
                t = q[6] - cb.z; dist = fmaf(t, t, dist);
                t = q[7] - cb.w; dist = fmaf(t, t, dist);
            }
            if (dist < best) {  // strict: the lowest index wins an exact tie
                best = dist;
                best_i = s0 + c;
            }
        }
    }
    if (ok) {
        part_dist[(int64_t)blockIdx.y * n + qi] = best;
        part_idx[(int64_t)blockIdx.y * n + qi] = best_i;
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

// scratch for the partial minima is owned by the caller-visible entry point in capi.cpp
int launch_vq_argmin_parts(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, int dim,
                           int parts, float* part_dist, int32_t* part_idx, int32_t* out_idx) {
    L3AC_REQUIRE(dim >= 1 && dim <= 8 && k > 0 && n >= 0, "vq_argmin: bad shape (dim=%d k=%d)", dim, k);
    if (n == 0) return L3AC_OK;
    const int shards = (int)ceil_div64(k, SHARD);
    const int shards_per_block = (int)ceil_div64(shards, parts);
    const dim3 grid((unsigned)ceil_div64(n, THREADS), (unsigned)parts);
    ProfScope prof(s, "vq_argmin_kernel", 3.0 * dim * (double)n * k, 4.0 * ((double)n * dim + (double)k * dim + n));
    if (dim <= 4)
        hipLaunchKernelGGL((vq_argmin_kernel<4>), grid, dim3(THREADS), 0, s, queries, n, dim, codebook, k, shards_per_block, part_dist, part_idx);
    else
        hipLaunchKernelGGL((vq_argmin_kernel<8>), grid, dim3(THREADS), 0, s, queries, n, dim, codebook, k, shards_per_block, part_dist, part_idx);
    L3AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_argmin_combine_kernel, dim3((unsigned)ceil_div64(n, THREADS)), dim3(THREADS), 0, s, part_dist,
                       part_idx, n, parts, out_idx);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
