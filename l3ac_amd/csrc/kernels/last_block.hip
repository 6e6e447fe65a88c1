// Decoder output stage (reference l3ac/modules.py:47-64, :174-179, :190-195):
//
//   LegacyUnit (x3, dilation 1 / 3 / 9):   y = x + Conv1x1( snake( Conv_k7_dilated( snake(x) ) ) )      C -> C
//   head:                                  audio = tanh( Conv_k7( snake(y) ) )                           C -> 1
//
// legacy_unit_kernel — persistent workgroups (2 per CU) walk tiles of 256 consecutive frames of one clip
// (8 waves x 32 frames); the weights are staged into LDS once per workgroup and the next tile's input rows are
// prefetched into registers while the current tile's products run:
//   * snake(x) for the 256 + 6*dil frames the block needs is evaluated ONCE into an LDS tile (zeros outside the
//     clip = the conv's zero padding), so the dilated k7 conv is an implicit product whose B operand is read from
//     that tile at row (frame + tap*dil);
//   * like the fused ConvUnit the products are transposed: X[n][m] = W1[n][(tap,c)] . S[m + (tap-3) dil][c]
//     leaves the hidden tile in accumulator registers (hidden channel on rows, frame on lanes), snake is applied
//     there, and the tile is fed straight back as the B operand of the 1x1 conv, Y[c][m] = W2[c][n] . X[n][m];
//   * C <= 32 channels are padded to one 32-row MFMA tile with zero weights.
// HBM traffic: 4C B in + 4C B out per frame (+ halo re-reads, L2-served); bound: fp32 MFMA issue
// ((7C/2 padded to 4 ceil(7C/8)) + 16 MFMAs of 32x32x2 per 32 frames).
//
// head_kernel — snake(y) staged once per 256-frame tile, one thread per output sample: 7C FMAs, HBM-bound.
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "split_bf16.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FRAMES = 256;  // frames per workgroup
constexpr int WAVES = 8;

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C>
struct LGeo {
    static_assert(C % 8 == 0, "a k-group of 8 must not straddle two taps");
    static constexpr int K = 7 * C;
    static constexpr int KQ = (K + 7) / 8;   // k groups of 8 (zero padded)
    static constexpr int SS = C + 4;         // S tile row stride: an odd number of 16-B slots (conflict-free b128)
    static constexpr int W1S = 8 * KQ + 4;
    static constexpr int W2S = 36;
    static constexpr int OFF_W1 = 0;                  // [32][W1S]
    static constexpr int OFF_W2 = OFF_W1 + 32 * W1S;  // [32][W2S]
    static constexpr int OFF_P = OFF_W2 + 32 * W2S;   // (alpha1, 1/alpha1) per hidden channel, b1, b2
    static constexpr int OFF_S = OFF_P + 32 * 4;
    static constexpr int lds_floats(int dil) { return OFF_S + (FRAMES + 6 * dil) * SS; }
};

template <int C>
__global__ __launch_bounds__(64 * WAVES, 4) void legacy_unit_kernel(const LegacyW w, const float* __restrict__ x,
                                                                float* __restrict__ y, int frames, int tiles_per_clip,
                                                                int total_tiles) {
    using G = LGeo<C>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1s = smem + G::OFF_W1;
    float* W2s = smem + G::OFF_W2;
    float* Ps = smem + G::OFF_P;
    float* Ss = smem + G::OFF_S;
    constexpr int NT = 64 * WAVES;
    // input staging: RPP rows per pass, each thread keeps ONE channel quad c0 (so snake's parameters are loaded
    // once) and PF rows of it; threads beyond RPP * C/4 idle (C = 24: 2 of 512)
    constexpr int RPP = NT / (C / 4);
    constexpr int PF = (FRAMES + 54 + RPP - 1) / RPP;  // dil <= 9
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dil = w.dil;
    const int rows = FRAMES + 6 * dil;

    const int srow = tid / (C / 4) + (tid < RPP * (C / 4) ? 0 : FRAMES + 54);  // idle threads: row beyond the tile
    const int sc0 = 4 * (tid % (C / 4));
    // input rows of one tile -> registers (zeros outside the clip = the conv's zero padding; snake(0) = 0)
    float4 pre[PF];
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * FRAMES;
        const float* clip = x + (int64_t)b * frames * C;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = srow + j * RPP;
            const int t = t0 - 3 * dil + row;
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows && t >= 0 && t < frames) pre[j] = *reinterpret_cast<const float4*>(clip + (int64_t)t * C + sc0);
        }
    };
    int tile = blockIdx.x;
    if (tile < total_tiles) prefetch(tile);

    // ---- weights (zero padded to 32 rows / 8*KQ columns), staged once per workgroup ---------------------------
    for (int i = tid; i < 32 * (G::W1S / 4); i += NT) {
        const int row = i / (G::W1S / 4), k = 4 * (i % (G::W1S / 4));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < C && k < G::K) v = *reinterpret_cast<const float4*>(w.w1 + row * G::K + k);
        *reinterpret_cast<float4*>(W1s + row * G::W1S + k) = v;
    }
    for (int i = tid; i < 32 * (G::W2S / 4); i += NT) {
        const int row = i / (G::W2S / 4), k = 4 * (i % (G::W2S / 4));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < C && k < C) v = *reinterpret_cast<const float4*>(w.w2 + row * C + k);
        *reinterpret_cast<float4*>(W2s + row * G::W2S + k) = v;
    }
    if (tid < 32) {
        const bool ok = tid < C;
        *reinterpret_cast<float4*>(Ps + 4 * tid) =
            make_float4(ok ? w.a1[tid] : 1.f, ok ? w.ia1[tid] : 0.f, ok ? w.b1[tid] : 0.f, ok ? w.b2[tid] : 0.f);
    }
    const int lj = lane & 31;
    const int lh = lane >> 5;
    const int m0 = 32 * wave;

    for (; tile < total_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * FRAMES;
        const float* clip = x + (int64_t)b * frames * C;
        // ---- S tile: snake(x) for frames [t0 - 3 dil, t0 + FRAMES + 3 dil) ---------------------------------
        const float4 al = *reinterpret_cast<const float4*>(w.a0 + sc0);
        const float4 ia = *reinterpret_cast<const float4*>(w.ia0 + sc0);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = srow + j * RPP;
            if (row < rows) {
                const float4 xv = pre[j];
                *reinterpret_cast<float4*>(Ss + row * G::SS + sc0) =
                    make_float4(snake_act(xv.x, al.x, ia.x), snake_act(xv.y, al.y, ia.y), snake_act(xv.z, al.z, ia.z),
                                snake_act(xv.w, al.w, ia.w));
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) prefetch(tile + gridDim.x);  // in flight during the products

        // An opaque zero offset per tile keeps the (tile-invariant) weight fragments in LDS: hoisted out of the
        // tile loop they would cost ~100 VGPRs and halve the occupancy.
        int woff = 0;
        asm volatile("" : "+s"(woff));
        const float* W1t = W1s + woff;
        const float* W2t = W2s + woff;
        const float* Pt = Ps + woff;
        if (t0 + m0 < frames) {  // wave-uniform: a wave wholly beyond the clip only takes part in the barriers
            // ---- X[n][m] = b1[n] + sum_{tap,c} W1[n][tap*C + c] S[m + (tap - 3) dil][c] ----------------------
            f32x16 xacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) xacc[r] = Pt[4 * rowmap(r, lh) + 2];
            // C % 8 == 0: both halves of a k-group of 8 lie in the same tap, so the S address is one per-lane base
            // + a wave-uniform tap offset + an immediate
            const float* sl = Ss + (m0 + lj) * G::SS + 4 * lh;
            const float* wl = W1t + lj * G::W1S + 4 * lh;
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const float* st = sl + tap * dil * G::SS;
#pragma unroll
                for (int g = 0; g < C / 8; ++g) {
                    const float4 sv = *reinterpret_cast<const float4*>(st + 8 * g);
                    const float4 wf = *reinterpret_cast<const float4*>(wl + tap * C + 8 * g);
                    xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, sv.x, xacc, 0, 0, 0);
                    xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, sv.y, xacc, 0, 0, 0);
                    xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.z, sv.z, xacc, 0, 0, 0);
                    xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.w, sv.w, xacc, 0, 0, 0);
                }
            }
            // ---- snake on the accumulator, then Y[c][m] = b2[c] + sum_n W2[c][n] X[n][m] -------------------
            f32x16 yacc;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {  // packed fp32 math on adjacent hidden channels
                const float4 p0 = *reinterpret_cast<const float4*>(Pt + 4 * rowmap(r, lh));
                const float4 p1 = *reinterpret_cast<const float4*>(Pt + 4 * rowmap(r + 1, lh));
                f32x2 hv, al, ia;
                hv.x = xacc[r]; hv.y = xacc[r + 1];
                al.x = p0.x; al.y = p1.x;
                ia.x = p0.y; ia.y = p1.y;
                const f32x2 s = snake_act2(hv, al, ia);  // padding rows: alpha 1, 1/alpha 0, bias 0 -> 0
                xacc[r] = s.x;
                xacc[r + 1] = s.y;
                yacc[r] = p0.w;
                yacc[r + 1] = p1.w;
            }
#pragma unroll
            for (int g = 0; g < (C + 7) / 8; ++g) {  // hidden rows >= C are zero padding: no contribution
                const float4 wf = *reinterpret_cast<const float4*>(W2t + lj * G::W2S + 8 * g + 4 * lh);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, xacc[4 * g], yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, xacc[4 * g + 1], yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.z, xacc[4 * g + 2], yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.w, xacc[4 * g + 3], yacc, 0, 0, 0);
            }
            // ---- residual + store: lane (frame lj, half lh) owns channels 8 g + 4 lh + {0..3} ---------------
            const int t = t0 + m0 + lj;
            if (t < frames) {
                const float* src = clip + (int64_t)t * C;
                float* dst = y + ((int64_t)b * frames + t) * C;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 8 * g + 4 * lh;
                    if (c0 < C) {
                        const float4 xr = *reinterpret_cast<const float4*>(src + c0);
                        *reinterpret_cast<float4*>(dst + c0) = make_float4(xr.x + yacc[4 * g], xr.y + yacc[4 * g + 1],
                                                                           xr.z + yacc[4 * g + 2], xr.w + yacc[4 * g + 3]);
                    }
                }
            }
        }
        __syncthreads();  // every wave is done with the S tile before the next one overwrites it
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 variant (split_bf16.hpp): the same tile walk, both products on the bf16 matrix cores at fp32 accuracy.
//   * snake(x) is split ONCE per element while it is staged: three bf16 planes of the S tile in LDS (row stride an odd
//     multiple of 16 B: conflict-free 16-B fragment reads);
//   * k order of the dilated conv: k step s = 8-channel groups g = 2s (lane half 0) and 2s+1 (lane half 1), g = tap * C/8 +
//     channel group, so a lane reads 8 consecutive channels of one tap's row; the W1 image is laid out to match;
//   * the hidden tile X^T (accumulator registers) is split in registers and is the B operand of the 1x1 conv; the W2
//     image uses the accumulator's row order (split_sigma);
//   * weight images hold only the C real rows: lanes of the padding rows read row C-1 (finite values whose products meet
//     zero weights or are never stored).
// MFMA shape: v_mfma_f32_16x16x32_bf16 (DESIGN.md 3.1 'MFMA shape'; a timing-only substitution ran this kernel 18 % faster): a wave's 32
// frames are two frame halves, the 32 (padded) hidden / output rows two 16-row tiles (the second holds C - 16 real rows); a k step
// of the dilated conv is 4 groups of 8 channels, lane (n = lane & 15, kg = lane >> 4) taking group 4 s + kg.
// MFMA cycles per 32 frames: (ceil(7C/32) * 2 NHH + NHH) * 2 * 6 * 16 vs (7C/2 + 16) * 64 for the fp32 kernel (C = 24: 2.7k vs 6.4k).
template <int C>
struct LSGeo {
    static constexpr int NG = C / 8;                // 8-channel groups per tap
    static constexpr int NGT = 7 * NG;              // groups in all
    static constexpr int NS1 = (NGT + 3) / 4;       // k steps (32) of the dilated conv
    static constexpr int NHH = (C + 15) / 16;       // 16-row tiles of hidden / output channels
    static constexpr int rows_of(int hh) { return C - 16 * hh < 16 ? C - 16 * hh : 16; }  // real rows of tile hh
    static constexpr int blk(int hh) { return 4 * rows_of(hh) * 16; }                       // bytes of one (tile, plane) fragment block
    static constexpr int STEP_BYTES = 3 * (blk(0) + (NHH > 1 ? blk(1) : 0));               // all tiles and planes of one k step
    static constexpr int off_of(int hh, int p) { return (hh == 0 ? 0 : 3 * blk(0)) + p * blk(hh); }
    static constexpr int PS = 2 * C + ((2 * C) % 32 == 16 ? 0 : 16);  // S plane row stride in bytes
    static constexpr int W1_BYTES = NS1 * STEP_BYTES;
    static constexpr int W2_BYTES = STEP_BYTES;
    static constexpr int OFF_W2 = W1_BYTES;
    static constexpr int OFF_P = OFF_W2 + W2_BYTES;  // 32 x (alpha1, 1/alpha1, b1, b2)
    static constexpr int OFF_S = OFF_P + 512;
    static constexpr int lds_bytes(int dil) { return OFF_S + 3 * (FRAMES + 6 * dil) * PS; }
    static_assert(C % 8 == 0 && C <= 32 && W1_BYTES % 16 == 0, "geometry");
};

typedef float f32x4_l __attribute__((ext_vector_type(4)));
// acc += a . b over one k step of 32, operands as their three planes, the six plane products in mfma_split's order
__device__ __forceinline__ f32x4_l mfma_split16(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4_l acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

template <int C>
__global__ __launch_bounds__(64 * WAVES, 4) void legacy_unit_split_kernel(const LegacyW w, const float* __restrict__ x,
                                                                        float* __restrict__ y, int frames, int tiles_per_clip,
                                                                        int total_tiles, int* __restrict__ counters) {
    using G = LSGeo<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* W1b = smem_b;
    unsigned char* W2b = smem_b + G::OFF_W2;
    float* Ps = reinterpret_cast<float*>(smem_b + G::OFF_P);
    unsigned char* Sb = smem_b + G::OFF_S;
    constexpr int NT = 64 * WAVES;
    constexpr int RPP = NT / (C / 4);
    constexpr int PF = (FRAMES + 54 + RPP - 1) / RPP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dil = w.dil;
    const int rows = FRAMES + 6 * dil;
    const int splane = rows * G::PS;  // bytes between the planes of the S tile
    const int srow_o = tid / (C / 4) + (tid < RPP * (C / 4) ? 0 : FRAMES + 54);
    const int sc0_o = 4 * (tid % (C / 4));

    float4 pre[PF];
    auto prefetch = [&](int tile, const int srow, const int sc0) __attribute__((always_inline)) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * FRAMES;
        const float* clip = x + (int64_t)b * frames * C;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = srow + j * RPP;
            const int t = t0 - 3 * dil + row;
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows && t >= 0 && t < frames) pre[j] = *reinterpret_cast<const float4*>(clip + (int64_t)t * C + sc0);
        }
    };
    // Tiles by counter (round 6, `counters` non-null; conv_unit_wide.hip, 'DYN'): the two workgroups of a CU are not served alike, so equal
    // static shares end at different times.  A workgroup's first two tiles are blockIdx and blockIdx + gridDim; thread 0 fetches the tile
    // after next at the top of an iteration and hands it over through LDS behind the iteration's last barrier.  The last workgroup to leave
    // zeroes the counters.  Which workgroup computes a tile does not enter its arithmetic: the same bits.
    __shared__ int next_tile_s;
    int tile = blockIdx.x;
    int next_tile = tile + (int)gridDim.x;
    if (tile < total_tiles) prefetch(tile, srow_o, sc0_o);

    // weight images are copied verbatim, once per workgroup
    for (int i = tid; i < (G::W1_BYTES + G::W2_BYTES) / 16; i += NT) {
        const int off = 16 * i;
        const u32x4 v = off < G::W1_BYTES ? *reinterpret_cast<const u32x4*>(w.w1_img + off)
                                         : *reinterpret_cast<const u32x4*>(w.w2_img + (off - G::W1_BYTES));
        *reinterpret_cast<u32x4*>(smem_b + off) = v;
    }
    if (tid < 32) {
        const bool ok = tid < C;
        *reinterpret_cast<float4*>(Ps + 4 * tid) =
            make_float4(ok ? w.a1[tid] : 1.f, ok ? w.ia1[tid] : 0.f, ok ? w.b1[tid] : 0.f, ok ? w.b2[tid] : 0.f);
    }
    const int m0 = 32 * wave;
    const int dps = dil * G::PS;

    for (; tile < total_tiles;) {
        int after_next = next_tile + (int)gridDim.x;
        if (counters && tid == 0) after_next = __hip_atomic_fetch_add(counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2 * (int)gridDim.x;
        // (the staging thread's row and channel quad, recomputed per tile from an opaque thread number like the lane values below)
        int tid_t = tid;
        asm volatile("" : "+v"(tid_t));
        const int srow = tid_t / (C / 4) + (tid_t < RPP * (C / 4) ? 0 : FRAMES + 54);
        const int sc0 = 4 * (tid_t % (C / 4));
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * FRAMES;
        const float* clip = x + (int64_t)b * frames * C;
        // ---- S tile: snake(x), split into three bf16 planes ----------------------------------------------------
        const float4 al = *reinterpret_cast<const float4*>(w.a0 + sc0);
        const float4 ia = *reinterpret_cast<const float4*>(w.ia0 + sc0);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int row = srow + j * RPP;
            if (row < rows) {
                const float4 xv = pre[j];
                unsigned a0, a1, a2, b0, b1, b2;
                split2(snake_act(xv.x, al.x, ia.x), snake_act(xv.y, al.y, ia.y), a0, a1, a2);
                split2(snake_act(xv.z, al.z, ia.z), snake_act(xv.w, al.w, ia.w), b0, b1, b2);
                unsigned char* dst = Sb + row * G::PS + 2 * sc0;
                *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(dst + splane) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(dst + 2 * splane) = make_uint2(a2, b2);
            }
        }
        __syncthreads();
        if (next_tile < total_tiles) prefetch(next_tile, srow, sc0);  // in flight during the products

        int woff = 0;  // opaque per tile: keeps the (tile-invariant) weight fragments in LDS instead of ~100 hoisted VGPRs
        asm volatile("" : "+s"(woff));
        // the lane number opaque per tile: everything derived from it — the per-step S offsets (selects on the k group), the per-lane weight
        // and parameter offsets, the residual / store addresses — is recomputed here, not hoisted out of the tile loop and spilled (round 6:
        // at 128 registers two to three such values came back from scratch behind an s_waitcnt vmcnt(0) right after the next tile's rows
        // had been requested: the prefetch was waited for on the spot)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int ln = lane_t & 15;  // frame within a 16-frame half (B operand / accumulator column) or weight row within a 16-row tile
        const int lg = lane_t >> 4;  // k group of a fragment; rows 4 lg .. 4 lg + 3 of an accumulator tile
        const int lgv = lg, lnv = ln;
        // per-lane offset inside a (tile hh, plane) weight block: lanes of the padding rows read the tile's last real row
        int wl[G::NHH];
#pragma unroll
        for (int hh = 0; hh < G::NHH; ++hh) wl[hh] = (lgv * G::rows_of(hh) + (lnv < G::rows_of(hh) ? lnv : G::rows_of(hh) - 1)) * 16;
        const unsigned char* W1t = W1b + woff;
        const unsigned char* W2t = W2b + woff;
        const float* Pt = Ps + woff;
        if (t0 + m0 < frames) {  // wave-uniform: a wave wholly beyond the clip only takes part in the barriers
            // ---- X^T[n][m] = b1[n] + sum_{tap,c} W1[n][tap][c] S[m + (tap - 3) dil][c] -------------------------
            // xt[hh][fh][i] = hidden row 16 hh + 4 lg + i at frame 16 fh + ln
            f32x4_l xt[G::NHH][2];
#pragma unroll
            for (int hh = 0; hh < G::NHH; ++hh) {
                f32x4_l bv;
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = Pt[4 * (16 * hh + 4 * lg + i) + 2];
                xt[hh][0] = bv;
                xt[hh][1] = bv;
            }
            const unsigned char* sl0 = Sb + (m0 + ln) * G::PS;
#pragma unroll
            for (int s = 0; s < G::NS1; ++s) {
                // group 4 s + kg of lane group kg: tap g / NG, channel group g % NG (groups past the end re-read group 0: zero weights)
                int off = 0;
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const int g = 4 * s + kg < G::NGT ? 4 * s + kg : 0;
                    const int o = (g / G::NG) * dps + (g % G::NG) * 16;
                    off = lgv == kg ? o : off;
                }
                bf16x8 sf[2][3], wf[G::NHH][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int fh = 0; fh < 2; ++fh) sf[fh][p] = *reinterpret_cast<const bf16x8*>(sl0 + fh * 16 * G::PS + off + p * splane);
#pragma unroll
                    for (int hh = 0; hh < G::NHH; ++hh)
                        wf[hh][p] = *reinterpret_cast<const bf16x8*>(W1t + s * G::STEP_BYTES + G::off_of(hh, p) + wl[hh]);
                }
#pragma unroll
                for (int hh = 0; hh < G::NHH; ++hh)
#pragma unroll
                    for (int fh = 0; fh < 2; ++fh) xt[hh][fh] = mfma_split16(wf[hh], sf[fh], xt[hh][fh]);
            }
            // ---- snake on the accumulators, split them, then Y^T[c][m] = b2[c] + sum_n W2[c][n] X^T[n][m] -----------
            // the B operand of frame half fh: word 2 hh + ip = hidden rows 16 hh + 4 lg + 2 ip, + 1 (k order sigma(lg, j) = j < 4 ?
            // 4 lg + j : 16 + 4 lg + j - 4, which the W2 image is built in); a missing second tile contributes zeros
            unsigned xw[2][3][4];
            f32x4_l yt[G::NHH][2];
#pragma unroll
            for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int k = 0; k < 4; ++k) xw[fh][p][k] = 0u;
#pragma unroll
            for (int hh = 0; hh < G::NHH; ++hh) {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const float4 p0 = *reinterpret_cast<const float4*>(Pt + 4 * (16 * hh + 4 * lg + 2 * ip));
                    const float4 p1 = *reinterpret_cast<const float4*>(Pt + 4 * (16 * hh + 4 * lg + 2 * ip + 1));
                    f32x2 al2, ia2;
                    al2.x = p0.x; al2.y = p1.x;
                    ia2.x = p0.y; ia2.y = p1.y;
#pragma unroll
                    for (int fh = 0; fh < 2; ++fh) {
                        f32x2 hv;
                        hv.x = xt[hh][fh][2 * ip]; hv.y = xt[hh][fh][2 * ip + 1];
                        const f32x2 sv = snake_act2(hv, al2, ia2);  // padding rows (copies of the last real row) stay finite
                        split2(sv.x, sv.y, xw[fh][0][2 * hh + ip], xw[fh][1][2 * hh + ip], xw[fh][2][2 * hh + ip]);
                        yt[hh][fh][2 * ip] = p0.w;
                        yt[hh][fh][2 * ip + 1] = p1.w;
                    }
                }
            }
            bf16x8 xb[2][3];
#pragma unroll
            for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    xb[fh][p] = __builtin_bit_cast(bf16x8, u32x4{xw[fh][p][0], xw[fh][p][1], xw[fh][p][2], xw[fh][p][3]});
#pragma unroll
            for (int rt = 0; rt < G::NHH; ++rt) {
                bf16x8 wf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[p] = *reinterpret_cast<const bf16x8*>(W2t + G::off_of(rt, p) + wl[rt]);
#pragma unroll
                for (int fh = 0; fh < 2; ++fh) yt[rt][fh] = mfma_split16(wf, xb[fh], yt[rt][fh]);
            }
            // ---- residual + store: lane (frame 16 fh + ln, row group lg) owns channels 16 rt + 4 lg + {0..3} ---------------
            // (round 6: the four residual pieces are requested together, from clamped addresses by every lane, and the stores are predicated —
            // inside `if (t < frames)` / `if (c0 < C)` each piece was load -> s_waitcnt vmcnt(0) -> add -> store: four serial round trips per tile)
            // (the statement below: not before the first product has been issued — requested earlier the pieces are held across it and spilled)
            asm volatile("" ::: "memory");
            float4 xr[2][G::NHH];
#pragma unroll
            for (int fh = 0; fh < 2; ++fh) {
                const int t = t0 + m0 + 16 * fh + ln;
                const float* src = clip + (int64_t)(t < frames ? t : frames - 1) * C;
#pragma unroll
                for (int rt = 0; rt < G::NHH; ++rt) {
                    const int c0 = 16 * rt + 4 * lg;
                    xr[fh][rt] = *reinterpret_cast<const float4*>(src + (c0 < C ? c0 : 0));
                }
            }
#pragma unroll
            for (int fh = 0; fh < 2; ++fh) {
                const int t = t0 + m0 + 16 * fh + ln;
                float* dst = y + ((int64_t)b * frames + t) * C;
#pragma unroll
                for (int rt = 0; rt < G::NHH; ++rt) {
                    const int c0 = 16 * rt + 4 * lg;
                    if (t < frames && c0 < C)
                        *reinterpret_cast<float4*>(dst + c0) = make_float4(xr[fh][rt].x + yt[rt][fh][0], xr[fh][rt].y + yt[rt][fh][1],
                                                                           xr[fh][rt].z + yt[rt][fh][2], xr[fh][rt].w + yt[rt][fh][3]);
                }
            }
        }
        if (counters && tid == 0) next_tile_s = after_next;
        __syncthreads();  // every wave is done with the S tile before the next one overwrites it
        tile = next_tile;
        next_tile = counters ? next_tile_s : after_next;  // (rewritten only behind the next iteration's first barrier)
    }
    if (counters && tid == 0) {
        const int left = __hip_atomic_fetch_add(counters + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (int)gridDim.x - 1) {
            __hip_atomic_store(counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(counters + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// head: audio[t] = tanh(b + sum_{tap,c} w[tap][c] snake(x[t + tap - 3][c]))
template <int C>
__global__ __launch_bounds__(FRAMES) void head_fused_kernel(const HeadW w, const float* __restrict__ x, int frames,
                                                           float* __restrict__ audio, const int pretanh) {
    constexpr int SS = C + 4;
    __shared__ __attribute__((aligned(16))) float Ss[(FRAMES + 6) * SS];
    __shared__ __attribute__((aligned(16))) float Ws[7 * C];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FRAMES;
    const float* clip = x + (int64_t)b * frames * C;
    for (int i = tid; i < 7 * C; i += FRAMES) Ws[i] = w.w[i];
    for (int i = tid; i < (FRAMES + 6) * (C / 4); i += FRAMES) {
        const int row = i / (C / 4), c0 = 4 * (i % (C / 4));
        const int t = t0 - 3 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < frames) {
            const float4 xv = *reinterpret_cast<const float4*>(clip + (int64_t)t * C + c0);
            const float4 al = *reinterpret_cast<const float4*>(w.alpha + c0);
            const float4 ia = *reinterpret_cast<const float4*>(w.inv_alpha + c0);
            v = make_float4(snake_act(xv.x, al.x, ia.x), snake_act(xv.y, al.y, ia.y), snake_act(xv.z, al.z, ia.z),
                            snake_act(xv.w, al.w, ia.w));
        }
        *reinterpret_cast<float4*>(Ss + row * SS + c0) = v;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= frames) return;
    float acc = w.b[0];
#pragma unroll
    for (int tap = 0; tap < 7; ++tap) {
#pragma unroll
        for (int c0 = 0; c0 < C; c0 += 4) {
            const float4 sv = *reinterpret_cast<const float4*>(Ss + (tid + tap) * SS + c0);
            const float4 wv = *reinterpret_cast<const float4*>(Ws + tap * C + c0);
            acc = fmaf(wv.x, sv.x, acc);
            acc = fmaf(wv.y, sv.y, acc);
            acc = fmaf(wv.z, sv.z, acc);
            acc = fmaf(wv.w, sv.w, acc);
        }
    }
    audio[(int64_t)b * frames + t] = pretanh ? acc : tanhf(acc);
}

template <int C>
int launch_legacy_t(hipStream_t s, const LegacyW& w, const float* x, float* y, int batch, int frames, bool split_route, int* counters) {
    L3AC_REQUIRE(w.dil >= 1 && w.dil <= 9, "legacy unit: dilation %d outside the LDS tile budget", w.dil);
    const bool split = split_route && w.w1_img && w.w2_img;
    using G = LGeo<C>;
    using GS = LSGeo<C>;
    const size_t lds = split ? (size_t)GS::lds_bytes(w.dil) : (size_t)G::lds_floats(w.dil) * sizeof(float);
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(legacy_unit_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::lds_floats(9) * sizeof(float))));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(legacy_unit_split_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GS::lds_bytes(9)));
        configured.done();
    }
    const double rows = (double)batch * frames;
    const int tiles_per_clip = (int)ceil_div64(frames, FRAMES);
    const int64_t total = (int64_t)tiles_per_clip * batch;
    L3AC_REQUIRE(total < (1ll << 31), "legacy unit: too many tiles");
    const unsigned grid = (unsigned)std::min<int64_t>(total, 2 * 256);  // persistent: 2 workgroups per CU
    ProfScope prof(s, split ? "legacy_unit_split_kernel" : "legacy_unit_kernel", rows * (2.0 * 7 * C * C + 2.0 * C * C + 40.0 * C),
                   rows * 8.0 * C);
    if (split)
        hipLaunchKernelGGL((legacy_unit_split_kernel<C>), dim3(grid), dim3(64 * WAVES), lds, s, w, x, y, frames, tiles_per_clip,
                           (int)total, total > 2 * (int64_t)grid ? counters : nullptr);
    else
        hipLaunchKernelGGL((legacy_unit_kernel<C>), dim3(grid), dim3(64 * WAVES), lds, s, w, x, y, frames, tiles_per_clip, (int)total);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

template <int C>
int launch_head_t(hipStream_t s, const HeadW& w, const float* x, int batch, int frames, float* audio, bool pretanh) {
    const double rows = (double)batch * frames;
    ProfScope prof(s, "head_fused_kernel", rows * (14.0 * C + 20.0 * C), rows * (4.0 * C + 4.0));
    hipLaunchKernelGGL((head_fused_kernel<C>), dim3((unsigned)ceil_div64(frames, FRAMES), (unsigned)batch), dim3(FRAMES), 0, s, w,
                       x, frames, audio, pretanh ? 1 : 0);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

bool last_block_fused_supported(int c, int max_dil) { return (c == 8 || c == 16 || c == 24 || c == 32) && max_dil <= 9; }

// x must not alias y (neighbouring blocks read each other's halo frames)
// counters: two zeroed ints of the context (the split kernel hands its tiles out by them and leaves them zeroed; null = static shares)
int launch_legacy_unit_fused(hipStream_t s, const LegacyW& w, const float* x, float* y, int batch, int frames, bool split, int* counters) {
    L3AC_REQUIRE(x != y && batch <= 65535, "legacy unit: bad arguments");
    switch (w.c) {
        case 8: return launch_legacy_t<8>(s, w, x, y, batch, frames, split, counters);
        case 16: return launch_legacy_t<16>(s, w, x, y, batch, frames, split, counters);
        case 24: return launch_legacy_t<24>(s, w, x, y, batch, frames, split, counters);
        case 32: return launch_legacy_t<32>(s, w, x, y, batch, frames, split, counters);
        default: l3ac_set_error("legacy unit: C=%d not supported by the fused kernel", w.c); return L3AC_EINVAL;
    }
}

int launch_head_fused(hipStream_t s, const HeadW& w, const float* x, int batch, int frames, float* audio, bool pretanh) {
    L3AC_REQUIRE(batch <= 65535, "head: bad arguments");
    switch (w.c) {
        case 8: return launch_head_t<8>(s, w, x, batch, frames, audio, pretanh);
        case 16: return launch_head_t<16>(s, w, x, batch, frames, audio, pretanh);
        case 24: return launch_head_t<24>(s, w, x, batch, frames, audio, pretanh);
        case 32: return launch_head_t<32>(s, w, x, batch, frames, audio, pretanh);
        default: l3ac_set_error("head: C=%d not supported by the fused kernel", w.c); return L3AC_EINVAL;
    }
}

// ---- host builders of the bf16x3 weight images (layouts: legacy_unit_split_kernel, LSGeo) -----------------------------
// Fragment blocks of v_mfma_f32_16x16x32_bf16's A operand: per k step, 16-row tile hh and plane p one block of 4 k groups x rows(hh)
// real rows x 16 B; lane (m, kg) reads (kg * rows + min(m, rows - 1)) * 16.
namespace {
struct LegacyImgGeo {
    int c, nhh, rows[2], blk[2], step_bytes;
    explicit LegacyImgGeo(int c_) : c(c_), nhh((c_ + 15) / 16) {
        rows[0] = c < 16 ? c : 16;
        rows[1] = c > 16 ? c - 16 : 0;
        blk[0] = 4 * rows[0] * 16;
        blk[1] = 4 * rows[1] * 16;
        step_bytes = 3 * (blk[0] + blk[1]);
    }
    size_t at(int step, int hh, int p, int kg, int m, int j) const {
        return (size_t)step * step_bytes + (hh == 0 ? 0 : 3 * blk[0]) + (size_t)p * blk[hh] + (size_t)(kg * rows[hh] + m) * 16 + 2 * j;
    }
};
}  // namespace

// W1 [C][7][C] (row r, tap, channel): k step s, k group kg = 8-channel group g = 4 s + kg = (tap g / NG, channels 8 (g % NG) ..)
std::vector<unsigned char> legacy_w1_image(const float* w1, int c) {
    const LegacyImgGeo geo(c);
    const int ng = c / 8, ngt = 7 * ng, ns1 = (ngt + 3) / 4;
    std::vector<unsigned char> img((size_t)ns1 * geo.step_bytes, 0);
    for (int s = 0; s < ns1; ++s)
        for (int kg = 0; kg < 4; ++kg) {
            const int g = 4 * s + kg;
            if (g >= ngt) continue;  // zero padding of the last step
            const int tap = g / ng, cg = g % ng;
            for (int r = 0; r < c; ++r)
                for (int j = 0; j < 8; ++j) {
                    uint16_t pl[3];
                    split3_host(w1[(size_t)r * 7 * c + (size_t)tap * c + 8 * cg + j], pl);
                    for (int p = 0; p < 3; ++p) std::memcpy(img.data() + geo.at(s, r / 16, p, kg, r % 16, j), &pl[p], 2);
                }
        }
    return img;
}

// W2 [C][C] (output row r, hidden n): one k step of 32 hidden rows in the order sigma(kg, j) = j < 4 ? 4 kg + j : 16 + 4 kg + j - 4
std::vector<unsigned char> legacy_w2_image(const float* w2, int c) {
    const LegacyImgGeo geo(c);
    std::vector<unsigned char> img((size_t)geo.step_bytes, 0);
    for (int kg = 0; kg < 4; ++kg)
        for (int j = 0; j < 8; ++j) {
            const int n = j < 4 ? 4 * kg + j : 16 + 4 * kg + j - 4;  // hidden channel this fragment element multiplies
            if (n >= c) continue;
            for (int r = 0; r < c; ++r) {
                uint16_t pl[3];
                split3_host(w2[(size_t)r * c + n], pl);
                for (int p = 0; p < 3; ++p) std::memcpy(img.data() + geo.at(0, r / 16, p, kg, r % 16, j), &pl[p], 2);
            }
        }
    return img;
}
