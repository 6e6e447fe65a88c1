// ConvUnit of the NARROW stages (C = 24 / 48 / 96), third form: one wave = 16 frames, weights through an LDS-DMA ring.
// Reference l3ac/modules.py:10-41 + Residual (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// conv_unit_split.hip (32 frames per wave on 32x32x16 MFMAs, 8 waves at ~220 registers = 2 per SIMD, weight chunks copied
// global -> registers -> LDS behind one workgroup barrier per chunk of 32 hidden channels) runs its matrix pipe 27-33 % busy:
// with two waves per SIMD in lock step behind the chunk barrier nothing covers a wave's fragment reads, its activation or its
// staging phases (round-2 profile: wait 0.27-0.41 of the wave cycles).  This form is built like trans_stack.hip instead:
//   * a wave owns 16 frames (one MFMA column tile) end to end; everything it keeps — the LayerNorm output as bf16x3 planes, the
//     output accumulators, one hidden pair — fits ~128 registers, so THREE to FOUR waves share a SIMD (two workgroups of six
//     waves per CU, independent of each other: their barriers do not couple);
//   * both products are weights (A) x activations (B) on v_mfma_f32_16x16x32_bf16 with exact bf16x3 operands; the hidden pair (32
//     channels x 16 frames, two accumulator tiles) goes through snake / GRN on the accumulator registers, is split there and is
//     the second product's B operand in the k order sigma (ring_common.hpp);
//   * W1 / W2 are ONE stream of fragment-ordered pieces in consumption order (per hidden pair: its 2 K1 pieces of W1, then its
//     C / 16 pieces of W2), L2-resident, staged by LDS-DMA into a ring of slots behind a counted s_waitcnt vmcnt and one raw
//     s_barrier per slot (no global -> register -> LDS copy, no per-chunk __syncthreads);
//   * the depth-wise conv reads its 7 taps straight from global memory (L1 / L2 absorb the 7-fold overlap of neighbouring
//     frames): no staging tile, no staging phase.
// Arithmetic: dw-conv as an fmaf chain over the taps (bias first), LayerNorm two-pass with eps 1e-8 and 1 / sqrt, snake with
// device_math.hpp's sin_squared, GRN with normaliser 1 — the operations of conv_unit_split.hip; the channel sums of
// the LayerNorm and the k order of the products differ, i.e. results agree to rounding, not bit for bit.
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "ring_common.hpp"
#include "split_bf16.hpp"

#include <vector>

namespace {

__device__ __forceinline__ f32x4_t ring_mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4_t acc) {
    return mfma6(a, b, acc);
}

// WAVES_ waves per workgroup, PER_CU workgroups per CU (register budget 512 / (WAVES_ PER_CU / 4) per lane), SP_ pieces per ring
// slot, RSLOTS_ ring slots.  RESIDENT: the unit's whole stream fits LDS (C = 24: 36 KB, C = 48: 126 KB) — loaded once per
// workgroup, then no DMA, no barrier: the waves run independently of each other.
template <int C, int WAVES_, int PER_CU_, int SP_, int RSLOTS_, bool RESIDENT_, int FT_ = 1, bool PRE_ = false>
struct RGeo {
    static constexpr bool PRE = PRE_;              // the next piece's fragment is read from LDS in front of the current piece's products
    static constexpr int FT = FT_;                 // 16-frame column tiles per wave: 2 = every weight fragment read from LDS feeds 12 MFMAs
    static constexpr int TF = 16 * FT_;            // frames per wave tile
    static constexpr int CT = (C + 15) / 16;       // accumulator tiles of the C channels
    static constexpr int K1 = (C + 31) / 32;       // k steps of the first product
    static constexpr int RT = CT;                  // output row tiles of the second product
    static constexpr int H4 = 4 * C;
    static constexpr int NT = H4 / 32;             // hidden pairs
    static constexpr int PP = 2 * K1 + RT;         // pieces per hidden pair: W1 (tile 0: K1, tile 1: K1), W2 (RT)
    static constexpr bool RESIDENT = RESIDENT_;
    static constexpr int SP = RESIDENT ? PP : SP_;   // pieces per ring slot
    static constexpr int SLOT = SP * 3072;
    static constexpr int RSLOTS = RESIDENT ? NT : RSLOTS_;
    static constexpr int PF = RSLOTS - 1;            // slots in flight
    static constexpr int SLOTS_PER_TILE = NT * PP / SP;
    static constexpr int DMA_WAVES = 3;              // wave p < 3 copies plane p of every piece of a slot: SP x 1 KB
    static constexpr int WAIT = (PF - 1) * SP;       // a DMA wave's copies that may stay outstanding at a step's end
    static constexpr int WAVES = WAVES_, PER_CU = PER_CU_;
    // LDS (bytes): ring | alpha, 1/alpha, gamma, beta [4][H4] | b1 [H4] | b2 [16 CT] | dw_w [7][16 CT], dw_b, ln_w, ln_b [16 CT]
    static constexpr int CP = 16 * CT;               // channels padded to whole tiles
    static constexpr int OFF_P = RSLOTS * SLOT;
    static constexpr int OFF_B1 = OFF_P + 4 * H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4 * 4;
    static constexpr int OFF_DW = OFF_B2 + CP * 4;
    static constexpr int OFF_TICKET = OFF_DW + 10 * CP * 4;  // (resident form) the workgroup's next tile ticket
    static constexpr int LDS = OFF_TICKET + 16;
    static_assert(PP % SP == 0 && (RESIDENT || WAIT <= 63) && LDS * PER_CU <= 160 * 1024 && WAVES * PER_CU <= 16, "bad geometry");
};

template <class G, int C>
__global__ __launch_bounds__(64 * G::WAVES, G::WAVES * G::PER_CU / 4) void conv_unit_ring_kernel(const ConvUnitW w, const float* __restrict__ x, float* __restrict__ y,
                                                                             const int frames, const int tiles_per_clip, const int n_tiles, const int tickets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ring[];
    float* const Ps = reinterpret_cast<float*>(smem_ring + G::OFF_P);
    float* const B1s = reinterpret_cast<float*>(smem_ring + G::OFF_B1);
    float* const B2s = reinterpret_cast<float*>(smem_ring + G::OFF_B2);
    float* const DWs = reinterpret_cast<float*>(smem_ring + G::OFF_DW);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, lg = lane >> 4;

    // ---- parameters resident for the lifetime of the workgroup (padding channels: zeros) ---------------------------------
    for (int i = tid; i < G::H4; i += 64 * G::WAVES) {
        Ps[i] = w.alpha[i];
        Ps[G::H4 + i] = w.inv_alpha[i];
        Ps[2 * G::H4 + i] = w.gamma[i];
        Ps[3 * G::H4 + i] = w.beta[i];
        B1s[i] = w.b1[i];
    }
    for (int i = tid; i < G::CP; i += 64 * G::WAVES) {
        const bool ok = i < C;
        B2s[i] = ok ? w.b2[i] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) DWs[tap * G::CP + i] = ok ? w.dw_w[tap * C + i] : 0.f;
        DWs[7 * G::CP + i] = ok ? w.dw_b[i] : 0.f;
        DWs[8 * G::CP + i] = ok ? w.ln_w[i] : 0.f;
        DWs[9 * G::CP + i] = ok ? w.ln_b[i] : 0.f;
    }
    int* const ticket_s = reinterpret_cast<int*>(smem_ring + G::OFF_TICKET);
    if (tid == 0) *ticket_s = 0;
    __syncthreads();  // (every plain load above is drained here, before the first hand-counted LDS-DMA)

    // ---- the weight stream: wave p < 3 copies plane p of every piece --------------------------------------------------
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_ring;
    const unsigned lane_off = 16u * (unsigned)lane;
    int dma_slot = 0;  // slot of the unit's stream to fetch next, modulo the stream length (wave-uniform)
    int ring_pos_w = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        if constexpr (!G::RESIDENT) {
            if (wave < G::DMA_WAVES) {
                const unsigned char* src = w.ring_img + (int64_t)dma_slot * G::SLOT + 1024 * wave;
                const unsigned dst = ring_lds + (unsigned)(ring_pos_w * G::SLOT + 1024 * wave);
#pragma unroll
                for (int i = 0; i < G::SP; ++i) ring_dma_1k(src + 3072 * i, lane_off, dst + 3072u * (unsigned)i);
            }
            dma_slot = dma_slot + 1 == G::SLOTS_PER_TILE ? 0 : dma_slot + 1;
            ring_pos_w = ring_pos_w + 1 == G::RSLOTS ? 0 : ring_pos_w + 1;
        }
    };
    auto step_sync = [&]() __attribute__((always_inline)) {
        if constexpr (!G::RESIDENT) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(G::WAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    if constexpr (G::RESIDENT) {  // the whole stream, once: by LDS-DMA, every 1-KB block of a wave in flight at once (through registers the
                                  // fill was a chain of load -> store round trips, most of what a single clip's workgroups did)
        static_assert((G::RSLOTS * G::SLOT) % 1024 == 0, "whole DMA blocks");
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_ring;
        for (int blk = __builtin_amdgcn_readfirstlane(tid >> 6); blk < G::RSLOTS * G::SLOT / 1024; blk += G::WAVES)
            ring_dma_1k(w.ring_img + 1024 * blk, 16u * (unsigned)(tid & 63), lds0 + 1024u * (unsigned)blk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
#pragma unroll
        for (int j = 0; j < G::PF; ++j) issue();
        step_sync();  // slot 0 has landed
    }
    int ring_pos_r = 0;
    const unsigned char* const ring_lane = smem_ring + 16 * lane;
    auto load_frag = [&](bf16x8 (&f)[3], const unsigned char* a) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            f[pl] = *reinterpret_cast<const bf16x8*>(a + 1024 * pl);
        }
    };
    // PRE: the fragment of the piece about to be consumed, read one piece ahead (carried across hidden pairs and tiles: the stream
    // wraps around).  A slot's barrier then sits in front of its LAST piece's products: that piece's fragment is in registers by
    // then (the lgkmcnt(0) of step_sync), and the next slot must have landed for the read-ahead.
    bf16x8 fcur[3];
    if constexpr (G::PRE) {
        issue();
        load_frag(fcur, ring_lane);
    }

    const int tile_stride = (int)gridDim.x * G::WAVES;
    // Round 6 (resident form, `tickets`): the waves of a workgroup walk their tiles independently, and a SIMD serves its oldest wave first —
    // with equal static shares the first waves of a workgroup finish early and the last ones run the end of the kernel on a thinly
    // occupied CU.  The workgroup keeps its tiles ((round r) gridDim + blockIdx) WAVES + j; a wave takes the next (r, j) by a ticket in LDS.
    for (int base = (int)blockIdx.x * G::WAVES, q = 0; base < n_tiles; base += tile_stride) {
        int tile = base + wave;
        if (G::RESIDENT && tickets) {
            if (lane == 0) q = __hip_atomic_fetch_add(ticket_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            q = __builtin_amdgcn_readfirstlane(q);
            base = ((q / G::WAVES) * (int)gridDim.x + (int)blockIdx.x) * G::WAVES;
            if (base >= n_tiles) break;
            tile = base + q % G::WAVES;
            base -= tile_stride;  // (the loop statement adds it back; the next ticket replaces it anyway)
            if (tile >= n_tiles) continue;  // the last round's missing tiles: the next ticket ends the walk
        }
        const bool tile_ok = tile < n_tiles;
        if (G::RESIDENT && !tile_ok) break;
        const int clip = tile_ok ? tile / tiles_per_clip : 0;
        const int t0 = tile_ok ? (tile - clip * tiles_per_clip) * G::TF : 0;
        const float* const xc = x + (int64_t)clip * frames * C + 4 * lg;  // (32-bit offsets inside a clip: frames * C < 2^31 / 4)
        bool frame_ok[G::FT];
        bf16x8 ap[G::FT][G::K1][3];
#pragma unroll
        for (int ft = 0; ft < G::FT; ++ft) {
            const int frame = t0 + 16 * ft + fl;             // this lane's frame (of column tile ft) inside its clip
            frame_ok[ft] = tile_ok && frame < frames;
            // ---- depth-wise conv k7 (zero padding at the clip's ends) + LayerNorm of this lane's frame: channels 16 t + 4 g + i ----
            f32x4_t a[G::CT];
            float s1 = 0.f;
#pragma unroll
            for (int t = 0; t < G::CT; ++t) {
                const bool ch_ok = 16 * t + 4 * lg < C;  // (C = 24: the upper half of tile 1 is padding)
                const int ch_off = ch_ok ? 16 * t : 0;    // padding lanes re-read tile 0 (their weights are zero)
                f32x4_t acc = *reinterpret_cast<const f32x4_t*>(DWs + 7 * G::CP + 16 * t + 4 * lg);
                f32x4_t xv[7];
                // every load is unconditional on a clamped address and masked afterwards: a guarded load is a branch and a full
                // wait each (the first build: 42 branches per tile, the whole front end serialised and 250 registers spilled)
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const int frc = min(max(frame + tap - 3, 0), frames - 1);
                    xv[tap] = *reinterpret_cast<const f32x4_t*>(xc + (unsigned)(frc * C + ch_off));
                }
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const int fr = frame + tap - 3;
                    const bool ok = frame_ok[ft] && fr >= 0 && fr < frames;
                    const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(DWs + tap * G::CP + 16 * t + 4 * lg);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(ok ? xv[tap][i] : 0.f, wv[i], acc[i]);
                }
                a[t] = acc;
                s1 += (acc[0] + acc[1]) + (acc[2] + acc[3]);
                // (tried, no change in the kernel's time: the centre taps of all channel tiles loaded first — one trip to memory instead
                // of CT — and a cache-line touch of the NEXT tile's rows by LDS-DMA three hidden pairs ahead)
                // one channel tile's 7 taps in flight at a time: nothing may be hoisted across (hipcc otherwise clusters all 7 CT loads
                // at the top of the tile and spills them)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            s1 = rows_sum(s1);
            const float mean = s1 / (float)C;
            float s2 = 0.f;
#pragma unroll
            for (int t = 0; t < G::CT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = (16 * t + 4 * lg + i < C) ? a[t][i] - mean : 0.f;
                    s2 = fmaf(d, d, s2);
                }
            s2 = rows_sum(s2);
            const float rstd = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
#pragma unroll
            for (int s = 0; s < G::K1; ++s) {
                f32x4_t v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 2 * s + u;
                    v[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (t < G::CT) {
                        const f32x4_t lw = *reinterpret_cast<const f32x4_t*>(DWs + 8 * G::CP + 16 * t + 4 * lg);  // (zeros in the padding)
                        const f32x4_t lb = *reinterpret_cast<const f32x4_t*>(DWs + 9 * G::CP + 16 * t + 4 * lg);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[u][i] = frame_ok[ft] ? (a[t][i] - mean) * rstd * lw[i] + lb[i] : 0.f;
                    }
                }
                planes_of(v[0], v[1], ap[ft][s]);
            }
        }
        // ---- output accumulators start at the pw_conv2 bias ------------------------------------------------------------------
        f32x4_t yacc[G::FT][G::RT];
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) {
            const f32x4_t b2v = *reinterpret_cast<const f32x4_t*>(B2s + 16 * rt + 4 * lg);
#pragma unroll
            for (int ft = 0; ft < G::FT; ++ft) yacc[ft][rt] = b2v;
        }

        // (round 6: a wave inside its hidden-pair loop outranks the waves that are in their front end or storing — as in
        // conv_unit_wide_kernel<96>; C = 48 at 256 clips 0.759 -> 0.748 ms; the same in legacy_unit_split_kernel measured slower)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int hp = 0; hp < G::NT; ++hp) {
            f32x4_t hx[G::FT][2];  // hidden channels 32 hp + 16 u + 4 g + i of this lane's frames, starting at the pw_conv1 bias
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4_t b1v = *reinterpret_cast<const f32x4_t*>(B1s + 32 * hp + 16 * u + 4 * lg);
#pragma unroll
                for (int ft = 0; ft < G::FT; ++ft) hx[ft][u] = b1v;
            }
            bf16x8 hb[G::FT][3];
            const unsigned char* slot_a = ring_lane + (G::RESIDENT ? hp : ring_pos_r) * G::SLOT;  // (PRE: the slot of piece 0)
            bf16x8 f[3];
            auto products = [&](auto q_, const bf16x8 (&fr)[3]) __attribute__((always_inline)) {
                constexpr int q = decltype(q_)::value;
                if constexpr (q < 2 * G::K1) {
#pragma unroll
                    for (int ft = 0; ft < G::FT; ++ft) hx[ft][q / G::K1] = ring_mfma6(fr, ap[ft][q % G::K1], hx[ft][q / G::K1]);
                    if constexpr (q == 2 * G::K1 - 1) {
                        // snake + GRN (normaliser 1) on the accumulator registers (layers.py:29-33, :112-115), then the planes
#pragma unroll
                        for (int ft = 0; ft < G::FT; ++ft) {
                            f32x4_t o[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const float* pp = Ps + 32 * hp + 16 * u + 4 * lg;
                                const f32x4_t al = *reinterpret_cast<const f32x4_t*>(pp);
                                const f32x4_t ia = *reinterpret_cast<const f32x4_t*>(pp + G::H4);
                                const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(pp + 2 * G::H4);
                                const f32x4_t be = *reinterpret_cast<const f32x4_t*>(pp + 3 * G::H4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float sv = snake_act(hx[ft][u][i], al[i], ia[i]);
                                    o[u][i] = fmaf(ga[i], sv, be[i]) + sv;
                                }
                            }
                            planes_of(o[0], o[1], hb[ft]);
                        }
                    }
                } else {
                    constexpr int rt = q - 2 * G::K1;
#pragma unroll
                    for (int ft = 0; ft < G::FT; ++ft) yacc[ft][rt] = ring_mfma6(fr, hb[ft], yacc[ft][rt]);
                }
            };
            ring_static_for<G::PP>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (G::PRE) {
                    bf16x8 fnext[3];
                    if constexpr ((q + 1) % G::SP == 0) {  // the slot's last piece: the next fragment comes from the next slot
                        if constexpr (G::RESIDENT) {
                            slot_a = ring_lane + (hp + 1 == G::NT ? 0 : hp + 1) * G::SLOT;
                        } else {
                            ring_pos_r = ring_pos_r + 1 == G::RSLOTS ? 0 : ring_pos_r + 1;
                            step_sync();  // every wave holds its last fragment of this slot; the next slot has landed
                            issue();      // refill the slot just released
                            slot_a = ring_lane + ring_pos_r * G::SLOT;
                        }
                        load_frag(fnext, slot_a);
                    } else {
                        load_frag(fnext, slot_a + ((q + 1) % G::SP) * 3072);
                    }
                    products(q_, fcur);
                    if constexpr (q != 2 * G::K1 - 1) {  // (not around the activation: its parameter reads are LDS reads too)
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);           // the read-ahead ...
                        __builtin_amdgcn_sched_group_barrier(0x008, 6 * G::FT, 0);   // ... in front of this piece's products
                    }
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fcur[pl] = fnext[pl];
                } else {
                    if constexpr (q % G::SP == 0) {  // a slot step begins: refill the slot consumed one step ago
                        issue();
                        slot_a = ring_lane + (G::RESIDENT ? hp : ring_pos_r) * G::SLOT;
                    }
                    load_frag(f, slot_a + (q % G::SP) * 3072);
                    products(q_, f);
                    if constexpr ((q + 1) % G::SP == 0) {
                        ring_pos_r = ring_pos_r + 1 == G::RSLOTS ? 0 : ring_pos_r + 1;
                        step_sync();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // one fragment live at a time (left alone the scheduler front-loads a slot's reads: 250 spills)
            });
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- residual + store --------------------------------------------------------------------------------------------------
#pragma unroll
        for (int ft = 0; ft < G::FT; ++ft) {
            const int frame = t0 + 16 * ft + fl;
            if (frame_ok[ft]) {
                const float* xrow = xc + (int64_t)frame * C;
                float* yrow = y + ((int64_t)clip * frames + frame) * C + 4 * lg;
#pragma unroll
                for (int rt = 0; rt < G::RT; ++rt) {
                    if (16 * rt + 4 * lg < C) {
                        const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(xrow + 16 * rt);
                        *reinterpret_cast<f32x4_t*>(yrow + 16 * rt) = xv + yacc[ft][rt];
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // leave no LDS-DMA in flight behind the workgroup
}

template <class G, int C>
int launch_ring(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames, const char* name) {
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_ring_kernel<G, C>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        configured.done();
    }
    const int tiles_per_clip = (frames + G::TF - 1) / G::TF;
    const int64_t tiles = (int64_t)batch * tiles_per_clip;
    L3AC_REQUIRE(tiles < ((int64_t)1 << 31) - 65536, "conv_unit_ring: too many tiles");
    int64_t blocks = ceil_div64(tiles, G::WAVES);
    const int64_t places = (int64_t)l3ac_device_cu_count() * G::PER_CU;
    if (blocks > places) blocks = places;
    const double rows = (double)batch * frames;
    ProfScope prof(s, name, rows * (14.0 * C + 16.0 * C * C), rows * 8.0 * C);
    const int tickets = tiles >= 4 * blocks * G::WAVES;  // (C = 48 at 256 clips: 0.865 -> 0.795 ms, profiles/r06/tickets_ab.txt)
    hipLaunchKernelGGL((conv_unit_ring_kernel<G, C>), dim3((unsigned)blocks), dim3(64 * G::WAVES), G::LDS, s, w, x, y, frames, tiles_per_clip, (int)tiles, tickets);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace


bool conv_unit_ring_supported(int c) { return c == 24 || c == 48; }
// the width the pipeline routes to this kernel by default (C = 24 stays on conv_unit_split_kernel, which is faster there — see the
// table in launch_conv_unit_ring; C = 96, rounds 2-3, runs on conv_unit_wide_kernel<96> since round 4 and its ring form was retired in
// round 5); the context option "narrow_ring" = 2 routes every supported width here (tests)
bool conv_unit_ring_preferred(int c) { return c == 48; }

// x must not alias y (neighbouring tiles read each other's halo frames)
int launch_conv_unit_ring(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames) {
    L3AC_REQUIRE(x != y && w.ring_img && batch > 0 && frames > 0, "conv_unit_ring: bad arguments");
    // Geometries measured on the 256-clip step (profiles/r03/README.md; ms for the stage's launches, conv_unit_split_kernel beside):
    //   C = 48 (2 launches)  split 0.89 | resident 16 x 1: 0.83-0.85, 12 x 1: 0.88 | ring 8 x 2: 0.93 | 32 frames per wave, 8 / 12 x 1: 0.94 / 0.90
    //                        | fragments read one piece ahead 16 x 1: 0.84, 12 x 1: 0.87
    //   C = 24 (1 launch)    split 0.46-0.48 | resident 12 x 1: 0.57-0.59, 6 x 2: 0.79-0.81 | 32 frames per wave 16 x 1: 0.55: not routed here
    // Neither 32 frames per wave (half the fragment reads per frame) nor reading one fragment ahead moves the time: the kernel is
    // not bound by the LDS reads.  What bounds it, from round 3's timing-only builds at C = 96 (1.50 ms): without the products 1.04;
    // without snake / GRN / split / conv taps 0.99; with neither 0.70; also without the weight stream and its barriers 0.53, also without
    // the fragment reads 0.37 (one read of x, LayerNorm, one write of y).  The parts add up to 1.78: little of the vector work (VALU
    // active 0.74 of the cycles at C = 48) hides under the products (matrix pipe 0.37), and at C <= 48 the vector work — 20
    // instructions of activation and 5.5 of operand split per hidden element — is the larger of the two: these widths are bound by
    // vector issue, not by the matrix pipe.
    // Small grids (a single clip: the streaming chunk): at most one wave per SIMD of the chip — workgroups of FOUR waves, one per CU,
    // so that every wave has its SIMD to itself (at 16 x 1 the few workgroups of a clip stack their waves four deep on a dozen CUs
    // while the rest of the chip idles).  Per frame the same operations in the same order: the same bits (tested: a clip alone
    // against the same clip inside a large batch).
    const int64_t tiles16 = (int64_t)batch * ((frames + 15) / 16);
    if (w.c == 48 && tiles16 <= 4LL * l3ac_device_cu_count())
        return launch_ring<RGeo<48, 4, 1, 0, 0, true>, 48>(s, w, x, y, batch, frames, "conv_unit_ring_kernel<48>");
    switch (w.c) {
        case 24: return launch_ring<RGeo<24, 12, 1, 0, 0, true>, 24>(s, w, x, y, batch, frames, "conv_unit_ring_kernel<24>");
        case 48:  // 126 KB stream: resident, one workgroup of 16 waves per CU, no barrier after the prologue
            return launch_ring<RGeo<48, 16, 1, 0, 0, true>, 48>(s, w, x, y, batch, frames, "conv_unit_ring_kernel<48>");
        default:
            l3ac_set_error("conv_unit_ring: C=%d not supported", w.c);
            return L3AC_EINVAL;
    }
}

// The unit's weight stream in consumption order: per hidden pair hp its 2 K1 pieces of W1 (rows 32 hp + 16 u, k step s) and its
// C / 16 pieces of W2 (rows 16 rt, k = the pair's 32 hidden channels).  w1 [4C][C], w2 [C][4C]; rows / columns beyond C: zeros.
std::vector<unsigned char> conv_unit_ring_image(const float* w1, const float* w2, int c) {
    const int k1 = (c + 31) / 32, rt_n = (c + 15) / 16, nt = 4 * c / 32;
    std::vector<unsigned char> img;
    img.reserve((size_t)nt * (2 * k1 + rt_n) * 3072);
    for (int hp = 0; hp < nt; ++hp) {
        for (int u = 0; u < 2; ++u)
            for (int s = 0; s < k1; ++s) ring_put_piece(img, w1, c, 4 * c, c, 32 * hp + 16 * u, 32 * s);
        for (int rt = 0; rt < rt_n; ++rt) ring_put_piece(img, w2, 4 * c, c, 4 * c, 16 * rt, 32 * hp);
    }
    return img;
}
