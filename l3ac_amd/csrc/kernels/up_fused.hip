// Decoder up layer in ONE kernel (reference l3ac/tconv/__init__.py:35-44 + l3ac/modules.py:160-164):
//
//     x' = x + merge(instnorm(branches(x))) * x          EnhanceBlock gate (the branch signals and their statistics come from
//                                                          enhance_branches / enhance_stats: they need the whole clip first)
//     c  = Conv1d(Cin -> Cout, k = 1)(x')                  weight-normed, folded at load
//     u  = Upsample(scale s, linear, align_corners=False)(c)
//     y  = ChannelNorm(u)                                  channels_first: (u - mean_C) / sqrt(var_C + eps) * w + b
//
// What it replaces, for the NARROW up layers (256 -> 96, 96 -> 48, 48 -> 24): the gated fp32-MFMA GEMM (a small-N product that ran at
// 2.1-3.4 TB/s of its own bytes: 0.15-0.18 ms each at 256 clips), the [frames][Cout] tensor it wrote, and row_kernel<LERP,CN> that
// read it back — 0.86 ms of the step for 2 GB of algorithmic traffic.
//
// One WAVE owns 16 consecutive input frames of a clip.  The conv is "weights (A) x activations (B)" on v_mfma_f32_16x16x32_bf16 with
// both operands as exact bf16x3 splits (ring_common.hpp: fp32 accuracy): lane (frame fl, k group g) loads the 8 channels
// sigma(g, .) of ITS frame for each k step of 32, applies the gate in registers, splits, and multiplies against the weight pieces
// resident in LDS; the result is in accumulator layout — channel 16 rt + 4 g + i of frame fl in register i of tile rt.  In that
// layout the linear upsample needs the neighbouring FRAME = the neighbouring lane of the 16-lane row (two DPP row shifts per
// register), and ChannelNorm's sums are RT x 4 registers plus the four k groups of a frame (permlane swaps): nothing goes through
// LDS or memory between the conv and the store.  Output frames s i .. s i + s - 1 of input frame i interpolate between frames i - 1,
// i, i + 1 only, so a wave stores the outputs of the 14 inner frames of its tile and tiles advance by 14 (12.5 % of the products and
// of the L2-served loads are redundant; HBM traffic is the algorithmic 4 Cin B in + 4 s Cout B out per frame).
// The arithmetic of gate, lerp and norm is row_kernel's, operation for operation.
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "ring_common.hpp"
#include "tickets.hpp"

#include <vector>

namespace {

constexpr int UF_WAVES = 16;  // one workgroup of 16 waves per CU (the 256 -> 96 weights are 144 KB of LDS)
constexpr int UF_CORE = 14;   // frames of a 16-frame tile whose outputs the wave stores

struct UpFusedArgs {
    const float* x;
    float* y;
    int batch, frames, scale;
    const unsigned char* img;  // RT x K1 pieces
    const float* bias;         // [cout]
    const float* nw;           // ChannelNorm affine [cout]
    const float* nb;
    float eps;
    const float* yi;           // raw branch signals [batch][frames][4]
    const float* stats;        // [batch][8] = mean[4], 1 / std[4]
    const float* in_w;         // InstanceNorm affine [4]
    const float* in_b;
    const float* gate_w;       // merge conv [cin][4]
    const float* gate_b;       // [cin]
};

template <int CIN, int COUT>
struct UfGeo {
    static constexpr int K1 = (CIN + 31) / 32, RT = (COUT + 15) / 16, CP = 16 * RT, CINP = 32 * K1;
    static constexpr int OFF_W = 0;
    static constexpr int OFF_PAR = RT * K1 * 3072;       // bias | nw | nb, CP floats each (zeros beyond cout)
    static constexpr int OFF_GW = OFF_PAR + 3 * CP * 4;  // gate_w [CINP][4] | gate_b [CINP] (zeros beyond cin)
    static constexpr int OFF_TICKET = OFF_GW + CINP * 5 * 4;  // the workgroup's next tile ticket (up_fused_kernel, 'tickets')
    static constexpr int LDS = OFF_TICKET + 16;
    static_assert(CIN % 16 == 0 && COUT % 8 == 0 && LDS <= 160 * 1024, "geometry");
};

// value of the lane one position down / up inside its 16-lane row (frame fl - 1 / fl + 1 of the same k group)
__device__ __forceinline__ float row_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));  // row_shr:1
}
__device__ __forceinline__ float row_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));  // row_shl:1
}

// DOWN (round 4): the same kernel as an encoder DOWN layer (reference l3ac/modules.py:96-99): Conv1d(k = stride) — a frame-major patch of
// `stride` frames is one contiguous row of CIN = stride x cin values — followed by ChannelNorm; no gate, no upsample, no halo (a wave stores
// all 16 frames of its tile).  It replaces a small-N fp32-MFMA GEMM, the tensor it wrote and row_kernel<PLAIN,CN>.
// WAVES: waves per workgroup — 16 (one workgroup per CU fills it), or 4 for grids of a few tiles (a single clip: 900 frames at 256 -> 96
// are 65 tiles = five 16-wave workgroups on five CUs, each filling 144 KB of LDS first: 31 us; as 17 workgroups of four waves ~ a third).
// A tile's arithmetic does not depend on it: the same bits.
template <int CIN, int COUT, bool DOWN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void up_fused_kernel(const UpFusedArgs p, const int tiles_per_clip, const int n_tiles, const int tickets) {
    using G = UfGeo<CIN, COUT>;
    constexpr int CORE = DOWN ? 16 : UF_CORE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_uf[];
    float* const par = reinterpret_cast<float*>(smem_uf + G::OFF_PAR);
    float* const gws = reinterpret_cast<float*>(smem_uf + G::OFF_GW);
    float* const gbs = gws + 4 * G::CINP;
    const int tid = threadIdx.x;
    {
        // the weight pieces by LDS-DMA, every 1-KB block of a wave in flight at once (through registers the fill was a chain of load ->
        // store round trips: most of the 24 us the 256 -> 96 layer took for a single clip, whose four workgroups do little else)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_uf;
        for (int blk = wv; blk < G::RT * G::K1 * 3; blk += WAVES) ring_dma_1k(p.img + 1024 * blk, 16u * (unsigned)(tid & 63), lds0 + 1024u * (unsigned)blk);
    }
    for (int i = tid; i < G::CP; i += 64 * WAVES) {
        par[i] = i < COUT ? p.bias[i] : 0.f;
        par[G::CP + i] = i < COUT ? p.nw[i] : 0.f;
        par[2 * G::CP + i] = i < COUT ? p.nb[i] : 0.f;
    }
    if constexpr (!DOWN) {
        for (int i = tid; i < G::CINP; i += 64 * WAVES) {
            const float4 w4 = i < CIN ? *reinterpret_cast<const float4*>(p.gate_w + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(gws + 4 * i) = w4;
            gbs[i] = i < CIN ? p.gate_b[i] : 0.f;
        }
    }
    int* const ticket_s = reinterpret_cast<int*>(smem_uf + G::OFF_TICKET);
    if (tid == 0) *ticket_s = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, lg = lane >> 4;
    const int T = p.frames, S = p.scale;
    const float rscale = (float)(1.0 / (double)S);
    float4 iw = make_float4(0.f, 0.f, 0.f, 0.f), ib = iw;
    if constexpr (!DOWN) {
        iw = *reinterpret_cast<const float4*>(p.in_w);
        ib = *reinterpret_cast<const float4*>(p.in_b);
    }
    const unsigned char* const wl = smem_uf + 16 * lane;
    // PRE (the narrow forms: at most three k steps = 24 registers): the NEXT tile's input rows and gate signals are requested before the
    // current tile's products — a wave's tile is otherwise load -> wait -> products -> s output frames with nothing of its own in flight
    // behind the loads (wait_any 0.6 of the wave cycles, 3.6-4.1 TB/s).  The same values reach the same operations: the same bits.
    // (Round 6, measured and not kept: the 256 -> 96 layer's batch form at EIGHT waves of 256 registers with this prefetch — 0.163 against
    // 0.167 ms at sixteen waves without spills, profiles/r06/up_fused_spills.txt — but hipcc contracts the gate's multiply-adds differently
    // in that instantiation than in the four-wave form a single clip takes: test_full_batch_properties_1kbps, clip alone != clip in a batch.)
    constexpr bool PRE = G::K1 <= 3;
    f32x4_t xpre[PRE ? G::K1 : 1][2];
    float4 ypre = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        if constexpr (PRE) {
            const int b = tile / tiles_per_clip;
            const int k = tile - b * tiles_per_clip;
            const int f = CORE * k - (DOWN ? 0 : 1) + fl;
            const bool valid = tile < n_tiles && f >= 0 && f < T;
            const int64_t r = valid ? (int64_t)b * T + f : 0;
            const float* const row = p.x + r * CIN;
#pragma unroll
            for (int s = 0; s < G::K1; ++s) {
                // (plain loads: the halo frames are re-read by the neighbouring tile — non-temporal ones measured 5-7 % slower)
                xpre[s][0] = *reinterpret_cast<const f32x4_t*>(row + 32 * s + 4 * lg);
                xpre[s][1] = 32 * s + 16 < CIN ? *reinterpret_cast<const f32x4_t*>(row + 32 * s + 16 + 4 * lg) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (!DOWN) ypre = *reinterpret_cast<const float4*>(p.yi + r * 4);
        }
    };
    const int tile_first = tickets ? take_tile<WAVES>(ticket_s, lane, n_tiles) : (int)blockIdx.x * WAVES + wave;
    prefetch(tile_first);

    const int lane_outer = lane;
    for (int tile = tile_first, tile_next = 0; tile < n_tiles; tile = tile_next) {
        tile_next = tickets ? take_tile<WAVES>(ticket_s, lane, n_tiles) : tile + (int)gridDim.x * WAVES;
        // (round 6: the lane number opaque per tile — what the tile derives from it (LDS offsets of the weight fragments, of the parameters and
        // of the gate table, row and store addresses) is computed here.  Hoisted out of the loop those values were spilled at 128 registers,
        // and a reload from scratch waits for vmcnt(0): for the NEXT tile's rows, requested a few instructions earlier)
        // (not at 48 -> 24, which has no spills and whose vector unit is the bound: recomputing the addresses cost it 8 %)
        int lane_t = lane_outer;
        if constexpr (G::K1 >= 3) asm volatile("" : "+v"(lane_t));
        const int fl = lane_t & 15, lg = lane_t >> 4;
        const unsigned char* const wl = smem_uf + 16 * lane_t;
        const int b = tile / tiles_per_clip;
        const int k = tile - b * tiles_per_clip;
        const int f = CORE * k - (DOWN ? 0 : 1) + fl;  // this lane's input frame
        const bool valid = f >= 0 && f < T;
        const int fv = valid ? f : 0;
        const float* const row = p.x + ((int64_t)b * T + fv) * CIN;
        f32x4_t xcur[PRE ? G::K1 : 1][2];
        float4 ycur = ypre;
        if constexpr (PRE) {
#pragma unroll
            for (int s = 0; s < G::K1; ++s) xcur[s][0] = xpre[s][0], xcur[s][1] = xpre[s][1];
            prefetch(tile_next);
        }
        // ---- gate input: z = InstanceNorm(branch signals) of this frame (rows.hip, SRC_GATE) ----------------------------------------
        float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
        if constexpr (!DOWN) {
            const float4 yraw = PRE ? ycur : *reinterpret_cast<const float4*>(p.yi + ((int64_t)b * T + fv) * 4);
            const float4 mean = *reinterpret_cast<const float4*>(p.stats + (int64_t)b * 8);
            const float4 istd = *reinterpret_cast<const float4*>(p.stats + (int64_t)b * 8 + 4);
            z0 = (yraw.x - mean.x) * istd.x * iw.x + ib.x;
            z1 = (yraw.y - mean.y) * istd.y * iw.y + ib.y;
            z2 = (yraw.z - mean.z) * istd.z * iw.z + ib.z;
            z3 = (yraw.w - mean.w) * istd.w * iw.w + ib.w;
        }
        // ---- c^T = W . x'^T + bias ---------------------------------------------------------------------------------------------------
        f32x4_t acc[G::RT];
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) acc[rt] = *reinterpret_cast<const f32x4_t*>(par + 16 * rt + 4 * lg);
        auto gated = [&](const f32x4_t xv, const int c0) __attribute__((always_inline)) -> f32x4_t {  // x + (gate_b + gate_w . z) * x
            if constexpr (DOWN) return xv;
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 gw = *reinterpret_cast<const float4*>(gws + 4 * (c0 + e));
                const float g = gbs[c0 + e] + gw.x * z0 + gw.y * z1 + gw.z * z2 + gw.w * z3;
                o[e] = xv[e] + g * xv[e];
            }
            return o;
        };
#pragma unroll
        for (int s = 0; s < G::K1; ++s) {
            const int c_lo = 32 * s + 4 * lg, c_hi = c_lo + 16;
            f32x4_t lo = f32x4_t{0.f, 0.f, 0.f, 0.f}, hi = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (PRE) {
                if (valid) lo = gated(xcur[s][0], c_lo);
                if (32 * s + 16 < CIN) {
                    if (valid) hi = gated(xcur[s][1], c_hi);
                }
            } else {
                // (round 6: loaded from the clamped row by EVERY lane and zeroed afterwards — behind `if (valid)` each k step's two loads sat in
                // a branch of their own, between reloads of spilled registers that wait for vmcnt(0): sixteen serial round trips per tile)
                const f32x4_t zero = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const f32x4_t xl = *reinterpret_cast<const f32x4_t*>(row + c_lo);   // (32 s + 15 < CIN for every k step: CIN % 16 == 0)
                lo = gated(xl, c_lo);
                lo = valid ? lo : zero;
                if (32 * s + 16 < CIN) {
                    const f32x4_t xh = *reinterpret_cast<const f32x4_t*>(row + c_hi);
                    hi = gated(xh, c_hi);
                    hi = valid ? hi : zero;
                }
            }
            bf16x8 bp[3];
            planes_of(lo, hi, bp);
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt) {
                bf16x8 wf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wl + (rt * G::K1 + s) * 3072 + 1024 * pl);
                acc[rt] = mfma6(wf, bp, acc[rt]);
            }
        }
        const bool core = DOWN ? f < T : fl >= 1 && fl <= UF_CORE && f < T;  // (f >= 0 for fl >= 1)
        float* const yclip = p.y + (int64_t)b * T * S * COUT;
        // ---- s output frames per input frame: lerp (ATen upsample_linear1d, rows.hip SRC_LERP), ChannelNorm, store -----------------
#pragma unroll 1
        for (int j = 0; j < S; ++j) {
            const int d = S * f + j;
            float src = __fsub_rn(__fmul_rn(rscale, (float)d + 0.5f), 0.5f);
            src = src < 0.f ? 0.f : src;
            const int i0 = (int)src;
            const int i1 = i0 + (i0 + 1 < T ? 1 : 0);
            float l1 = src - (float)i0;
            l1 = fminf(fmaxf(l1, 0.f), 1.f);
            const float l0 = 1.f - l1;
            const bool p0 = i0 < f, n1 = i1 > f;  // i0 in {f - 1, f}, i1 in {f, f + 1} (i1 == i0 only at the clip's last frame)
            f32x4_t v[G::RT];
            float s1 = 0.f;
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // (the neighbouring frame's value = the neighbouring lane's, fetched where it is used: kept for all s output frames,
                    // the two shifted copies of the accumulators cost the 256 -> 96 kernel 46 spilled registers)
                    // the shifts themselves run in EVERY lane, outside any lane-dependent control flow: a DPP read from a lane the EXEC mask has
                    // switched off returns 0, not that lane's register (the first build selected by branching and lost frame 0 at clip starts)
                    float u = acc[rt][i];
                    if constexpr (!DOWN) {
                        const float pv = row_prev(acc[rt][i]), nx = row_next(acc[rt][i]);
                        const float x0 = p0 ? pv : acc[rt][i];
                        const float x1 = n1 ? nx : acc[rt][i];
                        u = __fadd_rn(__fmul_rn(l0, x0), __fmul_rn(l1, x1));
                    }
                    v[rt][i] = (16 * rt + 4 * lg + i < COUT) ? u : 0.f;
                }
                s1 += (v[rt][0] + v[rt][1]) + (v[rt][2] + v[rt][3]);
            }
            const float mean = rows_sum(s1) / (float)COUT;
            float q = 0.f;
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dv = v[rt][i] - mean;
                    q += (16 * rt + 4 * lg + i < COUT) ? dv * dv : 0.f;
                }
            const float var = rows_sum(q) / (float)COUT;
            const float rstd = 1.0f / sqrtf(var + p.eps);
            if (core) {
                float* const dst = yclip + (int64_t)d * COUT + 4 * lg;
#pragma unroll
                for (int rt = 0; rt < G::RT; ++rt) {
                    if (16 * rt + 4 * lg < COUT) {
                        const f32x4_t w = *reinterpret_cast<const f32x4_t*>(par + G::CP + 16 * rt + 4 * lg);
                        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(par + 2 * G::CP + 16 * rt + 4 * lg);
                        f32x4_t o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = w[i] * ((v[rt][i] - mean) * rstd) + bb[i];
                        *reinterpret_cast<f32x4_t*>(dst + 16 * rt) = o;
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, bool DOWN, int WAVES>
int launch_uf_waves(hipStream_t s, const UpFusedArgs& a, int tiles_per_clip, int64_t tiles) {
    using G = UfGeo<CIN, COUT>;
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(up_fused_kernel<CIN, COUT, DOWN, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        configured.done();
    }
    int64_t blocks = ceil_div64(tiles, WAVES);
    const int64_t cus = l3ac_device_cu_count();
    if (blocks > cus) blocks = cus;
    // tiles by ticket (take_tile) where a wave has at least four of them (measured, profiles/r06/tickets_ab.txt: 48 -> 24 0.214 -> 0.193 ms,
    // 96 -> 48 and 256 -> 96 - 2 %; with two tiles per wave — the 48 -> 96 down layer — the tickets cost 5 %)
    const int tickets = tiles >= 4 * blocks * WAVES;
    hipLaunchKernelGGL((up_fused_kernel<CIN, COUT, DOWN, WAVES>), dim3((unsigned)blocks), dim3(64 * WAVES), G::LDS, s, a, tiles_per_clip, (int)tiles, tickets);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

template <int CIN, int COUT, bool DOWN = false>
int launch_uf(hipStream_t s, const UpFusedArgs& a) {
    constexpr int CORE = DOWN ? 16 : UF_CORE;
    const int tiles_per_clip = (a.frames + CORE - 1) / CORE;
    const int64_t tiles = (int64_t)a.batch * tiles_per_clip;
    L3AC_REQUIRE(tiles < ((int64_t)1 << 31) - 65536 && (int64_t)a.frames * a.scale < ((int64_t)1 << 30), "up_fused: too many tiles");
    const double rows = (double)a.batch * a.frames;
    char name[64];
    std::snprintf(name, sizeof(name), DOWN ? "down_fused_kernel<%d,%d>" : "up_fused_kernel<%d,%d>", CIN, COUT);
    ProfScope prof(s, name, rows * 2.0 * CIN * COUT, rows * 4.0 * (CIN + (DOWN ? 0 : 4) + (double)a.scale * COUT));
    // fewer 16-wave workgroups than half the CUs: four-wave workgroups spread the tiles over four times as many CUs (same bits)
    if (2 * ceil_div64(tiles, UF_WAVES) <= l3ac_device_cu_count()) return launch_uf_waves<CIN, COUT, DOWN, 4>(s, a, tiles_per_clip, tiles);
    return launch_uf_waves<CIN, COUT, DOWN, UF_WAVES>(s, a, tiles_per_clip, tiles);
}


// ---- encoder down layers 24 -> 48 and 48 -> 96, EXACT form (round 6): Conv1d(k = stride) + bias + ChannelNorm in one kernel with the
// arithmetic of the two kernels it replaces — gemm_f32_kernel (v_mfma_f32_32x32x2_f32, bias epilogue) and row_kernel<PLAIN,CN> — bit for
// bit, so that it can be the DEFAULT (the bf16x3 DOWN form above is an equally accurate but different rounding: one token of the stress
// weights changes sides with it).  What makes the bits equal:
//   * products: v_mfma_f32_16x16x4_f32 is, per output element, the k-ordered chain of fused multiply-adds in lane-group order
//     (tools/probes/mfma_f32_order_probe.hip), as 32x32x2 is in lane-half order.  gemm_f32_kernel feeds k in the order 0, 4, 1, 5, 2, 6, 3, 7
//     inside every group of 8 (its lanes read 16 B and instruction r takes element r of both halves); here lane group g of instruction j
//     of group q takes k = 8 q + 4 (g & 1) + (g >> 1) + 2 j: the same sequence (the scheme of gemm_f32_small_kernel), from zero;
//   * then acc + bias (gemm_epilogue, EPI_BIAS);
//   * ChannelNorm as row_kernel computes it for these widths: 12 lanes per row, one (C = 48) or two (C = 96) 16-B chunks per lane, a
//     lane's partial 0 + chunk_j (+ chunk_{j + 12}), summed by the segmented shift-down tree — for 12 lanes
//     ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7))) + ((v8 + v9) + (v10 + v11)).  In the accumulator layout chunk 4 rt + lg is lane
//     group lg of row tile rt, so rows_sum (x[lg 0] + x[lg 1]) + (x[lg 2] + x[lg 3]) of tile rt IS the tree's node of chunks 4 rt .. 4 rt + 3,
//     and the total is (node 0 + node 1) + node 2; the expressions of mean, variance and the affine are row_kernel's, verbatim.
// One wave owns 16 output frames; the weights sit in LDS as fp32 in fragment order: block (rt, q) = 64 lanes x 8 B = the two values a
// lane multiplies in group q.  The activation row of a frame is K contiguous floats: lane (frame, g) loads the 16 B it needs of every
// group (lanes g and g ^ 2 the same 16 B: L1 serves the second) — HBM traffic is the algorithmic 4 K in + 4 Cout out per frame.
template <int K, int COUT>
struct DxGeo {
    static constexpr int NQ = K / 8, RT = COUT / 16;
    static constexpr int OFF_PAR = RT * NQ * 512;        // bias | nw | nb, COUT floats each
    static constexpr int OFF_TICKET = OFF_PAR + 3 * COUT * 4;
    static constexpr int LDS = OFF_TICKET + 16;
    static_assert(K % 8 == 0 && COUT % 16 == 0 && (COUT == 48 || COUT == 96) && LDS <= 160 * 1024, "geometry (the norm's tree is written for 12 lanes per row)");
};

template <int K, int COUT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void down_exact_kernel(const UpFusedArgs p, const int tiles_per_clip, const int n_tiles, const int tickets) {
    using G = DxGeo<K, COUT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_uf[];
    float* const par = reinterpret_cast<float*>(smem_uf + G::OFF_PAR);
    const int tid = threadIdx.x;
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_uf;
        for (int blk = wv; blk < G::RT * G::NQ / 2; blk += WAVES) ring_dma_1k(p.img + 1024 * blk, 16u * (unsigned)(tid & 63), lds0 + 1024u * (unsigned)blk);
    }
    for (int i = tid; i < COUT; i += 64 * WAVES) {
        par[i] = p.bias[i];
        par[COUT + i] = p.nw[i];
        par[2 * COUT + i] = p.nb[i];
    }
    int* const ticket_s = reinterpret_cast<int*>(smem_uf + G::OFF_TICKET);
    if (tid == 0) *ticket_s = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, lg = lane >> 4;
    const int T = p.frames;
    const unsigned char* const wl = smem_uf + 8 * lane;
    const bool odd = (lg >> 1) != 0;  // elements 1, 3 of the 16 B (else 0, 2)
    for (int tile = tickets ? take_tile<WAVES>(ticket_s, lane, n_tiles) : (int)blockIdx.x * WAVES + wave; tile < n_tiles;
         tile = tickets ? take_tile<WAVES>(ticket_s, lane, n_tiles) : tile + (int)gridDim.x * WAVES) {
        const int b = tile / tiles_per_clip;
        const int f = 16 * (tile - b * tiles_per_clip) + fl;  // this lane's output frame
        const bool valid = f < T;
        const float* const row = p.x + ((int64_t)b * T + (valid ? f : 0)) * K + 4 * (lg & 1);
        f32x4_t acc[G::RT];
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // the row's 16-B pieces are requested PH groups at a time, all of a phase before its first product (a wave's loads are otherwise
        // issued one per group behind the previous group's products: too few bytes in flight for a kernel that is half HBM-bound)
        constexpr int PH = G::NQ <= 18 ? G::NQ : (G::NQ + 1) / 2;
#pragma unroll
        for (int q0 = 0; q0 < G::NQ; q0 += PH) {
            f32x4_t xv[PH];
#pragma unroll
            for (int i = 0; i < PH; ++i)
                if (q0 + i < G::NQ) xv[i] = *reinterpret_cast<const f32x4_t*>(row + 8 * (q0 + i));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PH; ++i) {
                if (q0 + i >= G::NQ) continue;
                const int q = q0 + i;
                const float e0 = odd ? xv[i][1] : xv[i][0], e1 = odd ? xv[i][3] : xv[i][2];
#pragma unroll
                for (int rt = 0; rt < G::RT; ++rt) {
                    const f32x2 wv = *reinterpret_cast<const f32x2*>(wl + (rt * G::NQ + q) * 512);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, e0, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, e1, acc[rt], 0, 0, 0);
                }
            }
        }
        // ---- + bias, ChannelNorm (rows.hip: row_kernel<PLAIN,CN>, 12 lanes per row), store ------------------------------------------
        // Every operation is spelled with its rounding (__fadd_rn / __fmul_rn / fmaf): hipcc contracts and packs row_kernel's expressions
        // differently per instantiation — in the compiled row kernels a chunk's squared deviations are (dx dx + dy dy) + (dz dz + dw dw) with
        // separate multiplies at one chunk per lane (C = 48) and fma(dx, dx, dy dy) + fma(dz, dz, dw dw) at two (C = 96); the affine is
        // fma(w, rstd d, b) in both (ISA of rows.hip, hipcc 7.2) — and this kernel must not be reshaped on its own.
        // tests/test_gpu_blocks.py::test_down_and_k3_layers asserts the equality with the two-kernel route bit for bit.
        float4 v[G::RT];
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(par + 16 * rt + 4 * lg);
            v[rt] = make_float4(__fadd_rn(acc[rt][0], bv[0]), __fadd_rn(acc[rt][1], bv[1]), __fadd_rn(acc[rt][2], bv[2]), __fadd_rn(acc[rt][3], bv[3]));
        }
        constexpr int NCH = G::RT / 3;  // chunks per lane of the row kernel: chunk j = 4 rt + lg and, at C = 96, chunk j + 12 = tile rt + 3
        float node[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float4 c = v[n + 3 * i];
                s = __fadd_rn(s, __fadd_rn(__fadd_rn(c.x, c.y), __fadd_rn(c.z, c.w)));
            }
            node[n] = rows_sum(s);
        }
        const float mean = __fadd_rn(__fadd_rn(node[0], node[1]), node[2]) / (float)COUT;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float4 c = v[n + 3 * i];
                const float dx = __fsub_rn(c.x, mean), dy = __fsub_rn(c.y, mean), dz = __fsub_rn(c.z, mean), dw = __fsub_rn(c.w, mean);
                const float qc = NCH == 1 ? __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fadd_rn(__fmul_rn(dz, dz), __fmul_rn(dw, dw)))
                                          : __fadd_rn(fmaf(dx, dx, __fmul_rn(dy, dy)), fmaf(dz, dz, __fmul_rn(dw, dw)));
                q = i == 0 ? qc : __fadd_rn(q, qc);
            }
            node[n] = rows_sum(q);
        }
        const float var = __fadd_rn(__fadd_rn(node[0], node[1]), node[2]) / (float)COUT;
        const float rstd = 1.0f / sqrtf(__fadd_rn(var, p.eps));
        if (valid) {
            float* const dst = p.y + ((int64_t)b * T + f) * COUT + 4 * lg;
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt) {
                const float4 w = *reinterpret_cast<const float4*>(par + COUT + 16 * rt + 4 * lg);
                const float4 bb = *reinterpret_cast<const float4*>(par + 2 * COUT + 16 * rt + 4 * lg);
                *reinterpret_cast<float4*>(dst + 16 * rt) =
                    make_float4(fmaf(w.x, __fmul_rn(rstd, __fsub_rn(v[rt].x, mean)), bb.x), fmaf(w.y, __fmul_rn(rstd, __fsub_rn(v[rt].y, mean)), bb.y),
                                fmaf(w.z, __fmul_rn(rstd, __fsub_rn(v[rt].z, mean)), bb.z), fmaf(w.w, __fmul_rn(rstd, __fsub_rn(v[rt].w, mean)), bb.w));
            }
        }
    }
}

template <int K, int COUT, int WAVES>
int launch_dx_waves(hipStream_t s, const UpFusedArgs& a, int tiles_per_clip, int64_t tiles) {
    using G = DxGeo<K, COUT>;
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(down_exact_kernel<K, COUT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        configured.done();
    }
    int64_t blocks = ceil_div64(tiles, WAVES);
    const int64_t cus = l3ac_device_cu_count();
    if (blocks > cus) blocks = cus;
    const int tickets = tiles >= 4 * blocks * WAVES;  // (as launch_uf_waves: 24 -> 48 0.121 -> 0.115 ms; 48 -> 96 has two tiles per wave)
    hipLaunchKernelGGL((down_exact_kernel<K, COUT, WAVES>), dim3((unsigned)blocks), dim3(64 * WAVES), G::LDS, s, a, tiles_per_clip, (int)tiles, tickets);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

template <int K, int COUT>
int launch_dx(hipStream_t s, const UpFusedArgs& a) {
    const int tiles_per_clip = (a.frames + 15) / 16;
    const int64_t tiles = (int64_t)a.batch * tiles_per_clip;
    L3AC_REQUIRE(tiles < ((int64_t)1 << 31) - 65536, "down_exact: too many tiles");
    const double rows = (double)a.batch * a.frames;
    char name[64];
    std::snprintf(name, sizeof(name), "down_exact_kernel<%d,%d>", K, COUT);
    ProfScope prof(s, name, rows * 2.0 * K * COUT, rows * 4.0 * (K + COUT));
    // few tiles (a single clip: 34 tiles at 48 -> 96): four-wave workgroups — one wave per SIMD, each tile's 360 fp32 MFMAs on a matrix pipe of
    // its own instead of four tiles per pipe on three CUs (a tile's arithmetic does not depend on the form: the same bits)
    if (2 * ceil_div64(tiles, 16) <= l3ac_device_cu_count()) return launch_dx_waves<K, COUT, 4>(s, a, tiles_per_clip, tiles);
    return launch_dx_waves<K, COUT, 16>(s, a, tiles_per_clip, tiles);
}

}  // namespace

bool up_fused_supported(int cin, int cout) { return (cin == 256 && cout == 96) || (cin == 96 && cout == 48) || (cin == 48 && cout == 24); }

// the 1x1 conv's weight [cout][cin] as RT x K1 pieces (16 rows x 32 k, three bf16 planes; zeros beyond cout / cin)
std::vector<unsigned char> up_fused_image(const float* w, int cin, int cout) {
    std::vector<unsigned char> img;
    const int k1 = (cin + 31) / 32, rt_n = (cout + 15) / 16;
    img.reserve((size_t)rt_n * k1 * 3072);
    for (int rt = 0; rt < rt_n; ++rt)
        for (int s = 0; s < k1; ++s) ring_put_piece(img, w, cin, cout, cin, 16 * rt, 32 * s);
    return img;
}

// EnhanceBlock gate + up layer; yi / stats: the branch signals and statistics of x (enhance_branches / enhance_stats)
int launch_up_fused(hipStream_t s, const EnhW& e, const UpW& w, const float* x, const float* yi, const float* stats, float* y, int batch, int frames) {
    L3AC_REQUIRE(w.fused_img && x && y && yi && stats && batch > 0 && frames > 0 && e.c == w.cin, "up_fused: bad arguments");
    UpFusedArgs a{};
    a.x = x; a.y = y; a.batch = batch; a.frames = frames; a.scale = w.scale; a.img = w.fused_img; a.bias = w.b; a.nw = w.nw; a.nb = w.nb; a.eps = 1e-8f;
    a.yi = yi; a.stats = stats; a.in_w = e.in_w; a.in_b = e.in_b; a.gate_w = e.gate_w; a.gate_b = e.gate_b;
    if (w.cin == 256 && w.cout == 96) return launch_uf<256, 96>(s, a);
    if (w.cin == 96 && w.cout == 48) return launch_uf<96, 48>(s, a);
    if (w.cin == 48 && w.cout == 24) return launch_uf<48, 24>(s, a);
    l3ac_set_error("up_fused: %d -> %d not supported", w.cin, w.cout);
    return L3AC_EINVAL;
}

// ---- encoder down layers: Conv1d(k = stride) + ChannelNorm (up_fused_kernel<K, Cout, DOWN>) ---------------------------------------------
bool down_fused_supported(int cin, int stride, int cout) {
    const int k = cin * stride;
    return (k == 144 && cout == 48) || (k == 240 && cout == 96) || (k == 192 && cout == 96);
}
// x [batch][frames_out * stride][cin] -> y [batch][frames_out][cout]; w.fused_img = up_fused_image(w.w, stride * cin, cout)
int launch_down_fused(hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames_out) {
    L3AC_REQUIRE(w.fused_img && w.nw && w.nb && x && y && x != y && batch > 0 && frames_out > 0, "down_fused: bad arguments");
    UpFusedArgs a{};
    a.x = x; a.y = y; a.batch = batch; a.frames = frames_out; a.scale = 1; a.img = w.fused_img; a.bias = w.b; a.nw = w.nw; a.nb = w.nb; a.eps = 1e-8f;
    const int k = w.cin * w.stride;
    if (k == 144 && w.cout == 48) return launch_uf<144, 48, true>(s, a);
    if (k == 240 && w.cout == 96) return launch_uf<240, 96, true>(s, a);
    if (k == 192 && w.cout == 96) return launch_uf<192, 96, true>(s, a);
    l3ac_set_error("down_fused: %d x %d -> %d not supported", w.cin, w.stride, w.cout);
    return L3AC_EINVAL;
}

// ---- the EXACT one-kernel form of the same down layers (down_exact_kernel): fp32 weights in fragment order ------------------------------
// block (rt, q) = 64 lanes x 8 B: lane (r = lane & 15, g = lane >> 4) holds w[16 rt + r][8 q + 4 (g & 1) + (g >> 1)] and the same + 2
std::vector<unsigned char> down_exact_image(const float* w, int k, int cout) {
    const int nq = k / 8, rt_n = cout / 16;
    std::vector<unsigned char> img((size_t)rt_n * nq * 512);
    for (int rt = 0; rt < rt_n; ++rt)
        for (int q = 0; q < nq; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4, kk = 8 * q + 4 * (g & 1) + (g >> 1);
                const float v[2] = {w[(size_t)(16 * rt + r) * k + kk], w[(size_t)(16 * rt + r) * k + kk + 2]};
                std::memcpy(img.data() + ((size_t)(rt * nq + q) * 64 + lane) * 8, v, 8);
            }
    return img;
}
int launch_down_exact(hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames_out) {
    L3AC_REQUIRE(w.exact_img && w.nw && w.nb && x && y && x != y && batch > 0 && frames_out > 0, "down_exact: bad arguments");
    UpFusedArgs a{};
    a.x = x; a.y = y; a.batch = batch; a.frames = frames_out; a.scale = 1; a.img = w.exact_img; a.bias = w.b; a.nw = w.nw; a.nb = w.nb; a.eps = 1e-8f;
    const int k = w.cin * w.stride;
    if (k == 144 && w.cout == 48) return launch_dx<144, 48>(s, a);
    if (k == 240 && w.cout == 96) return launch_dx<240, 96>(s, a);
    if (k == 192 && w.cout == 96) return launch_dx<192, 96>(s, a);
    l3ac_set_error("down_exact: %d x %d -> %d not supported", w.cin, w.stride, w.cout);
    return L3AC_EINVAL;
}
