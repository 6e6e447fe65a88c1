// Fused ConvUnit for the WIDE stages (C = 128 / 192 / 256), both channel contractions on the bf16 matrix cores at fp32
// accuracy ("bf16x3", split_bf16.hpp); reference l3ac/modules.py:10-41 + Residual (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// What it replaces: dwconv_ln_kernel -> gemm_split (C -> 4C, snake + GRN) -> gemm_split (4C -> C, + residual), i.e. three
// launches whose 4C-wide hidden tensor made two trips through the fabric (944 MB per unit at C = 256 for 256 x 1 s clips)
// and whose A operands were re-split by every column block (4 to 16 times).
//
// Design (gfx950).  tools/experiments/conv_unit_wide_v1.hip is the first version (32 frames per wave, ONE wave per SIMD with
// 512 registers): correct, but its stamps showed every non-MFMA phase exposed — 30 % of a pass in loads / stores, the
// activation's VALU work only partly hidden, nothing to cover a barrier or an LDS wait.  This version is built around
// having TWO waves per SIMD that are never in the same phase:
//   * One wave owns 16 frames end to end (v_mfma_f32_16x16x32_bf16: D[16 x 16] += A[16 x 32] . B[32 x 16]).  Products are
//     evaluated transposed — hidden channel on the accumulator's rows (registers), frame on its lanes — so a pair of hidden
//     tiles X^T (2 x 16 hidden x 16 frames, 8 accumulator registers) goes through snake / GRN in registers, is split into
//     bf16 planes there and IS the B operand of the second product (k index 8 g + j of lane group g <-> hidden
//     16 (j >> 2) + 4 g + (j & 3) =: sigma(g, j); the weight images are stored in that order).  Resident per wave: the split
//     LayerNorm output (3 planes x C/32 k steps x 4 = 3C/8 registers) and the output accumulators (C/4): 160 registers at
//     C = 256, so a wave fits in 256 and a 512-thread workgroup puts two waves on every SIMD.
//   * Weights: ONE cyclic stream of fragment-ordered bf16x3 images (conv_unit_wide_image), slots of C/32 k steps
//     (24 KB at C = 256): per 32-hidden block b the slots W1(tile 2b+2), W1(tile 2b+3), W2(block b, first half of the
//     output tiles), W2(block b, second half).  All 8 waves read the same slot at the same time: LDS-DMA
//     (global_load_lds, no staging registers) into a ring of 4 slots, 3 in flight behind a counted s_waitcnt vmcnt and one
//     raw s_barrier per slot.  The image (3.1 MB at C = 256) stays L2-resident: every CU walks it in step.
//   * The sum over hidden blocks may start ANYWHERE in the cycle.  Waves 0-3 (group A) and 4-7 (group B) run the same
//     program half a period apart: period = 5 iterations of "memory phase" (store the finished tile, fetch + depth-wise
//     conv + LayerNorm + split the next one; the wave only passes the step barriers) + NB + 1 iterations of compute.
//     While one wave of a SIMD loads, stores or runs VALU code, the other one owns the matrix pipe.
//   * Tiles are 16 consecutive GLOBAL rows; taps outside a lane's clip read a row of zeros (no masks, no branches).
// Algorithmic work per frame: 16 C^2 + 14 C FLOP, 8 C bytes (x in, y out).
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "split_bf16.hpp"

#include <vector>

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int C>
struct WGeo {
    static constexpr int H4 = 4 * C;
    static constexpr int NB = H4 / 32;        // 32-hidden blocks = iterations of the cyclic stream
    static constexpr int NK = C / 32;         // k steps (of 32) of the first product
    static constexpr int CT = C / 16;         // output tiles (of 16 channels)
    static constexpr int NCH = C / 16;        // 16-channel chunks of the depth-wise conv staging
    static constexpr int SLOT = NK * 3072;    // one hidden tile of W1 = half the output tiles of one block of W2
    static constexpr int NSTEP = 4;           // slots per iteration == ring size
    static constexpr int PF = NSTEP - 1;      // slots in flight
    static constexpr int RING = NSTEP * SLOT;
    static constexpr int TOTAL = NB * NSTEP;  // slots of one cycle of the stream
    static constexpr int PIECES = SLOT / 1024;             // 1-KB LDS-DMA pieces of a slot, dealt over the 8 waves
    static constexpr int PMAX = (PIECES + 7) / 8, PMIN = PIECES / 8;
    static constexpr int WAIT = PMIN * (PF - 1);           // own pieces that may stay outstanding when a step ends
    static constexpr int MEM_IT = 5;                       // iterations of the memory phase
    static constexpr int PERIOD_IT = NB + 1 + MEM_IT;      // iterations per tile and group
    static constexpr int OFFSET_IT = PERIOD_IT / 2;        // group B runs this far behind group A
    // LDS (bytes): alpha, 1/alpha, gamma, beta [4][H4] | b1 [H4] | b2 [C] | dw_w [7][C], dw_b, ln_w, ln_b | ring | staging
    // (the tables first: every table access is then one per-lane base + a 16-bit immediate)
    static constexpr int OFF_P = 0;
    static constexpr int OFF_B1 = OFF_P + 4 * H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4 * 4;
    static constexpr int OFF_DW = OFF_B2 + C * 4;
    static constexpr int OFF_RING = OFF_DW + 10 * C * 4;
    static constexpr int OFF_XS = OFF_RING + RING;         // per wave: 2 buffers of [32 rows][64 B] (16 channels)
    static constexpr int XBUF = 2048;
    static constexpr int LDS = OFF_XS + 8 * 2 * XBUF;
    static constexpr int WARM = (22 * C * 4 + 1023) / 1024; // 1-KB pieces of the 22 rows a tile needs
    static_assert(C % 64 == 0 && NK >= 2 && CT % 4 == 0 && PERIOD_IT % 2 == 0, "bad geometry");
    static_assert(PMIN >= 1 && WAIT + 2 <= 63, "vmcnt field too narrow");
    static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
    static_assert(2 * SLOT < 65536 && OFF_RING < 65536 && OFF_RING % 16 == 0, "table / ring offsets must fit the ds immediate");
    static_assert(WARM <= NB, "not enough iterations to warm the next tile's rows");
    static_assert(CT / 4 + NCH / 2 + NK / 2 <= MEM_IT * NSTEP, "memory phase too short for its work");
};

__device__ float g_zero_row[512];  // what out-of-clip depth-wise taps read (zero-initialised)

#ifdef L3AC_WIDE_STAMPS  // diagnostic build (tools/wide_stamps.py): s_memtime at the phase boundaries of waves 0 and 4
__device__ unsigned long long g_wide_stamps[256 * 2 * 16 * 4];
#define WIDE_STAMP(slot)                                                                                                      \
    do {                                                                                                                      \
        if (lane == 0 && (wave & 3) == 0 && pass < 16)                                                                        \
            g_wide_stamps[(((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + pass) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WIDE_STAMP(slot) do { } while (0)
#endif

// element j of lane group g of an MFMA k step <-> index inside the 32-block (both products; the images are built with it)
__host__ __device__ inline int wide_sigma(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// one 1-KB LDS-DMA piece, executed only while idx < limit (both wave-uniform): lane l copies 16 B from its own source pointer
// to lds_dst + 16 l.  M0 is written in the statement that uses it (guide §5.7); the test and the branch live INSIDE the
// statement, so the surrounding code stays one basic block; the copy is invisible to hipcc's s_waitcnt bookkeeping and is
// counted by hand.
__device__ __forceinline__ void dma16_if(const unsigned char* gsrc, unsigned lds_dst, int idx, int limit) {
    unsigned keep;
    asm volatile(
        "s_cmp_lt_i32 %3, %4\n\ts_cbranch_scc0 .Lwide_skip_%=\n\t"
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n"
        ".Lwide_skip_%=:"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst), "s"(idx), "s"(limit)
        : "memory", "scc");
}
__device__ __forceinline__ void dma16(const unsigned char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int C>
__global__ __launch_bounds__(512, 2) void conv_unit_wide_kernel(const ConvUnitW w, const float* __restrict__ x,
                                                              float* __restrict__ y, const int64_t rows, const int frames) {
    using G = WGeo<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
    unsigned char* ring = smem_wide + G::OFF_RING;
    float* Ps = reinterpret_cast<float*>(smem_wide + G::OFF_P);
    float* B1s = reinterpret_cast<float*>(smem_wide + G::OFF_B1);
    float* B2s = reinterpret_cast<float*>(smem_wide + G::OFF_B2);
    float* DWs = reinterpret_cast<float*>(smem_wide + G::OFF_DW);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;   // 0: waves 0-3, 1: waves 4-7 (the second wave of every SIMD)
    const int lm = lane & 15;      // frame within the tile (MFMA column)
    const int lg = lane >> 4;      // lane group: channels / hidden rows 4 lg + {0..3} of every 16
    // LDS addressing: a handful of per-lane byte offsets kept opaque to the optimiser, everything else an immediate of the
    // ds_read (16-bit field).  Left alone, hipcc materialises every base + constant combination in a register of its own
    // and spills them: ~90 registers at C = 256, reloaded — with a vmcnt(0) that drains the LDS-DMA queue — inside the loop.
    auto opaque = [](int v) __attribute__((always_inline)) -> int {
        asm volatile("" : "+v"(v));
        return v;
    };
    const int ring_lo = opaque(G::OFF_RING + 16 * lane);                // ring positions 0, 1
    const int ring_hi = opaque(G::OFF_RING + 16 * lane + 2 * G::SLOT);  // ring positions 2, 3
    // 16 x lane group, recomputed where it is needed (two instructions) instead of living in a register for the whole kernel
    auto lg16_now = [&]() __attribute__((always_inline)) -> int {
        return opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 0x30);
    };

    // ---- the weight stream: wave w copies pieces w, w + 8, ... of every slot ----------------------------------------
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned char* const src_lane = w.wide_img + 1024 * wave + 16 * lane;
    int gs = 0;  // steps taken so far = slots consumed or passed (wave-uniform); ring position of a step = gs % 4, known statically
    // One step = one slot of the stream.  Every wave, whatever its phase, requests its share of slot gs + PF (into the ring
    // position of the slot the previous step finished with) and joins the barrier that ends the step.
    auto issue = [&](int ring_pos) __attribute__((always_inline)) {
#ifndef L3AC_WIDE_NODMA
        const int slot = (gs + G::PF) % G::TOTAL;
        const unsigned char* src = src_lane + (int64_t)slot * G::SLOT;
        const unsigned dst = ring_lds + (unsigned)(ring_pos * G::SLOT) + 1024u * (unsigned)wave;
#pragma unroll
        for (int i = 0; i < G::PMAX; ++i) {
            if (i < G::PMIN) dma16(src + 8192 * i, dst + 8192u * i);
            else dma16_if(src + 8192 * i, dst + 8192u * i, wave + 8 * i, G::PIECES);
        }
#endif
    };
    // end of a step: this wave's pieces of the NEXT slot have landed, then everybody's have, and everybody is done reading the
    // slot that the next step's DMA overwrites
    auto step_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");
#ifndef L3AC_WIDE_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        ++gs;
    };

    // ---- parameters resident for the lifetime of the workgroup ---------------------------------------------------
    for (int i = tid; i < G::H4; i += 512) {
        Ps[i] = w.alpha[i];
        Ps[G::H4 + i] = w.inv_alpha[i];
        Ps[2 * G::H4 + i] = w.gamma[i];
        Ps[3 * G::H4 + i] = w.beta[i];
        B1s[i] = w.b1[i];
    }
    for (int i = tid; i < C; i += 512) {
        B2s[i] = w.b2[i];
        DWs[7 * C + i] = w.dw_b[i];
        DWs[8 * C + i] = w.ln_w[i];
        DWs[9 * C + i] = w.ln_b[i];
    }
    for (int i = tid; i < 7 * C; i += 512) DWs[i] = w.dw_w[i];
    __syncthreads();
    // prologue of the stream: slots 0 .. PF-1, slot 0 awaited
    {
#pragma unroll
        for (int j = 0; j < G::PF; ++j) {
            gs = j - G::PF;
            issue(j);
        }
        gs = 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    auto idle_iteration = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::NSTEP; ++i) {
            issue((i + G::PF) % G::NSTEP);
            step_sync();
        }
    };

    // One k step (32 of the reduction) = 6 plane products a_i . b_j, i + j <= 2, smallest first; the weight planes (A operand)
    // are read from the ring when needed: the partner wave of the SIMD covers the LDS latency.
    auto plane = [&](int ring_pos, int piece, int pl) __attribute__((always_inline)) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem_wide + (ring_pos < 2 ? ring_lo : ring_hi) + (ring_pos & 1) * G::SLOT + piece * 3072 + 1024 * pl);
    };
    auto kstep = [&](f32x4_t acc, int ring_pos, int piece, const bf16x8 (&bq)[3]) __attribute__((always_inline)) -> f32x4_t {
        const bf16x8 p0 = plane(ring_pos, piece, 0), p1 = plane(ring_pos, piece, 1), p2 = plane(ring_pos, piece, 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p2, bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, bq[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p0, bq[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p0, bq[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p0, bq[0], acc, 0, 0, 0);
#ifndef L3AC_WIDE_FREE_SCHED
        // one k step's planes at a time: the partner wave covers the LDS latency, and at C = 256 the registers a deeper
        // prefetch would take do not exist (VALU / SALU work may still move across)
        __builtin_amdgcn_sched_barrier(0x6);
#endif
        return acc;
    };

    const int64_t n_tiles = (rows + 15) / 16;
    const int64_t tile_stride = (int64_t)gridDim.x * 8;
    const int n_pass = (int)((n_tiles + tile_stride - 1) / tile_stride);

    // group B starts half a period late (and group A finishes half a period early): both take the same number of steps
    if (group == 1) {
#pragma unroll 1
        for (int i = 0; i < G::OFFSET_IT; ++i) idle_iteration();
    }

    bf16x8 ap[G::NK][3];  // LayerNorm output of this wave's tile, split: k step s, plane
    f32x4_t yacc[G::CT];  // output accumulators: tile ct, rows = channels 16 ct + 4 lg + r
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) yacc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned xs_lds = ring_lds - (unsigned)G::OFF_RING + (unsigned)(G::OFF_XS + 2 * G::XBUF * wave);
    const unsigned char* xs = smem_wide + G::OFF_XS + 2 * G::XBUF * wave;

#pragma unroll 1
    for (int pass = 0; pass <= n_pass; ++pass) {
        const int64_t row0 = ((int64_t)pass * tile_stride + (int64_t)blockIdx.x * 8 + wave) * 16;  // this pass's tile
        const int64_t row = row0 + lm;
        const bool fetch = pass < n_pass;            // (the last trip only stores the last tile)
        const bool row_ok = fetch && row < rows;
        const int64_t prev_row = row - tile_stride * 16;  // the tile finished by the previous pass
        const bool store = pass > 0 && prev_row < rows;
        WIDE_STAMP(0);

        // =============================== memory phase: MEM_IT iterations of the stream ===============================
        // unit u of the phase (one per step): store CT/4 output tiles of the finished tile | stage + convolve two 16-channel
        // chunks of the next tile | LayerNorm + split two k steps | nothing
        {
            constexpr int U_STORE = G::CT / 4, U_CONV = G::NCH / 2, U_SPLIT = G::NK / 2;
            const int lg_dw = G::OFF_DW + lg16_now();  // dw_w [7][C], dw_b, ln_w, ln_b: this lane's 4 channels of chunk 0
            // -- staging plan of the next tile: 22 rows (16 frames + 3 either side) + rows of zeros, 16 channels per chunk,
            //    LDS image [32 rows][4 slots of 16 B], slot = quad ^ ((row >> 2) & 3)
            const float* xsrc[2];
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int u = 64 * pc + lane, r = u >> 2, quad = (u & 3) ^ ((r >> 2) & 3);
                const int64_t rg = row0 - 3 + r;
                xsrc[pc] = (fetch && r < 22 && rg >= 0 && rg < rows) ? x + rg * C + 4 * quad : g_zero_row + 4 * quad;
            }
            auto stage = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    dma16(reinterpret_cast<const unsigned char*>(xsrc[pc] + 16 * c), xs_lds + (unsigned)((c & 1) * G::XBUF + 1024 * pc));
            };
            const int t = row_ok ? (int)(row % frames) : 0;  // frame inside its clip
            int xoff[7];  // LDS byte offset of this lane's tap reads inside a chunk (tap row or a zero row, its own quad)
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const bool ok = row_ok && t + tap - 3 >= 0 && t + tap - 3 < frames;
                const int r = ok ? lm + tap : 22;
                xoff[tap] = 64 * r + 16 * (lg ^ ((r >> 2) & 3));
            }
            float a[4 * G::NCH];  // depth-wise conv output: chunk c, channel 16 c + 4 lg + e
            float s1 = 0.f, mean = 0.f, rstd = 0.f;
#pragma unroll
            for (int it = 0; it < G::MEM_IT; ++it) {
#pragma unroll
                for (int i = 0; i < G::NSTEP; ++i) {
                    const int u = G::NSTEP * it + i;
                    if (u < U_STORE) {
                        // ---- residual + store of the finished tile (xtract/nn/layers.py:59-62): 4 channels of one frame per lane
                        if (store) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int ct = 4 * u + k;
                                const float4 xr = *reinterpret_cast<const float4*>(x + prev_row * C + 16 * ct + 4 * lg);
                                *reinterpret_cast<float4*>(y + prev_row * C + 16 * ct + 4 * lg) =
                                    make_float4(xr.x + yacc[ct][0], xr.y + yacc[ct][1], xr.z + yacc[ct][2], xr.w + yacc[ct][3]);
                            }
                        }
                        if (u == U_STORE - 1) {  // the staging buffers are free from here on: first two chunks of the next tile
                            stage(0);
                            stage(1);
                        }
                    } else if (u < U_STORE + U_CONV) {
                        // ---- depth-wise conv k7 (modules.py:33) on chunks 2v, 2v + 1
                        const int v = u - U_STORE;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int c = 2 * v + h;
                            // chunk c has landed once only younger copies are outstanding: the ring pieces requested since (at least
                            // PMIN) and, except for the last chunk, its successor's two pieces
                            if (c + 1 < G::NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + G::PMIN) : "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PMIN) : "memory");
                            const unsigned char* xb_ = xs + (c & 1) * G::XBUF;
                            const unsigned char* dwp = smem_wide + lg_dw + 64 * c;  // this lane's 4 channels of chunk c
                            float4 acc = *reinterpret_cast<const float4*>(dwp + 7 * C * 4);
#pragma unroll
                            for (int tap = 0; tap < 7; ++tap) {
                                const float4 xv = *reinterpret_cast<const float4*>(xb_ + xoff[tap]);
                                const float4 wv = *reinterpret_cast<const float4*>(dwp + tap * C * 4);
                                acc.x = fmaf(xv.x, wv.x, acc.x);
                                acc.y = fmaf(xv.y, wv.y, acc.y);
                                acc.z = fmaf(xv.z, wv.z, acc.z);
                                acc.w = fmaf(xv.w, wv.w, acc.w);
                            }
                            a[4 * c] = acc.x; a[4 * c + 1] = acc.y; a[4 * c + 2] = acc.z; a[4 * c + 3] = acc.w;
                            s1 += (acc.x + acc.y) + (acc.z + acc.w);
                            if (c + 2 < G::NCH) {  // refill the buffer just consumed (this wave's reads of it have returned: they fed the fmas)
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                stage(c + 2);
                            }
                        }
                    } else if (u < U_STORE + U_CONV + U_SPLIT) {
                        // ---- LayerNorm over the frame's C channels (layers.py:80), then the bf16x3 split of two k steps
                        const int v = u - U_STORE - U_CONV;
                        if (v == 0) {
                            s1 += __shfl_xor(s1, 16, 64);
                            s1 += __shfl_xor(s1, 32, 64);
                            mean = s1 / (float)C;
                            float s2 = 0.f;
#pragma unroll
                            for (int k = 0; k < 4 * G::NCH; ++k) {
                                const float d = a[k] - mean;
                                s2 = fmaf(d, d, s2);
                            }
                            s2 += __shfl_xor(s2, 16, 64);
                            s2 += __shfl_xor(s2, 32, 64);
                            rstd = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int s = 2 * v + h;  // k step s = chunks 2s (elements 0-3) and 2s + 1 (elements 4-7)
                            unsigned p[3][4];
#pragma unroll
                            for (int hc = 0; hc < 2; ++hc) {  // chunk 2s + hc = elements 4 hc .. 4 hc + 3
                                const int c = 2 * s + hc;
                                const float4 lw = *reinterpret_cast<const float4*>(smem_wide + lg_dw + 64 * c + 8 * C * 4);
                                const float4 lb = *reinterpret_cast<const float4*>(smem_wide + lg_dw + 64 * c + 9 * C * 4);
                                const float v0 = row_ok ? (a[4 * c] - mean) * rstd * lw.x + lb.x : 0.f;
                                const float v1 = row_ok ? (a[4 * c + 1] - mean) * rstd * lw.y + lb.y : 0.f;
                                const float v2 = row_ok ? (a[4 * c + 2] - mean) * rstd * lw.z + lb.z : 0.f;
                                const float v3 = row_ok ? (a[4 * c + 3] - mean) * rstd * lw.w + lb.w : 0.f;
                                split2(v0, v1, p[0][2 * hc], p[1][2 * hc], p[2][2 * hc]);
                                split2(v2, v3, p[0][2 * hc + 1], p[1][2 * hc + 1], p[2][2 * hc + 1]);
                            }
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) ap[s][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
                        }
                    }
                    issue((i + G::PF) % G::NSTEP);
                    step_sync();
                }
            }
        }
        WIDE_STAMP(1);
        if (!fetch) break;  // (uniform) the last trip had only a tile to store

        // =============================== compute phase: NB + 1 iterations of the stream ===============================
        {
            const int lg_b2 = G::OFF_B2 + lg16_now();
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                const float4 b2 = *reinterpret_cast<const float4*>(smem_wide + lg_b2 + 64 * ct);
                yacc[ct] = f32x4_t{b2.x, b2.y, b2.z, b2.w};
            }
        }
        // The NEXT tile's rows (22 x C floats, contiguous) are pulled towards the chip while this one computes: one 1-KB LDS-DMA
        // copy per iteration into the (idle) staging buffer.  Nobody reads it; the staging of the next memory phase then finds
        // the lines in the Infinity Cache instead of waiting for HBM.
        const int64_t next_row0 = row0 + tile_stride * 16;
        const int64_t warm_lo = (next_row0 - 3 < 0 ? 0 : next_row0 - 3) * C;  // in floats
        const int warm_n = (pass + 1 < n_pass && next_row0 < rows + 3) ? G::WARM : 0;
        const int64_t x_last = rows * C - 4;

        // the stream is at iteration (gs / 4) % NB: its W1 slots belong to hidden block b0 = that + 1
        int blk = ((gs >> 2) + 1) % G::NB;
        f32x4_t xa[2];         // hidden tile pair: activated (read) during the step that accumulates its successor, then replaced
        unsigned xbp[3][4];    // its bf16x3 planes: the B operand of the second product
#pragma unroll
        for (int h = 0; h < 2; ++h) xa[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int k = 0; k < 4; ++k) xbp[pl][k] = 0u;
        // (params of block b: per-lane byte offset 16 lg + 128 b behind the table's start, the rest immediates)
        auto bias1 = [&](int b, int h) __attribute__((always_inline)) -> f32x4_t {
            const float4 v = *reinterpret_cast<const float4*>(smem_wide + (G::OFF_B1 + lg16_now() + 128 * b) + 64 * h);
            return f32x4_t{v.x, v.y, v.z, v.w};
        };
        // snake + GRN (normaliser == 1; layers.py:29-33, :112-115) on rows (r, r + 1) of tile h of block b, and the bf16x3 split
        // of the pair into elements 4 h + r, 4 h + r + 1 of the second product's B operand
        auto act_pair = [&](int b, int h, int r) __attribute__((always_inline)) {
            const unsigned char* pp = smem_wide + (G::OFF_P + lg16_now() + 128 * b) + 64 * h + 4 * r;
            const f32x2 al = *reinterpret_cast<const f32x2*>(pp);
            const f32x2 ia = *reinterpret_cast<const f32x2*>(pp + 4 * G::H4);
            const f32x2 ga = *reinterpret_cast<const f32x2*>(pp + 8 * G::H4);
            const f32x2 be = *reinterpret_cast<const f32x2*>(pp + 12 * G::H4);
            f32x2 hv;
            hv.x = xa[h][r];
            hv.y = xa[h][r + 1];
#ifdef L3AC_WIDE_NOACT
            const f32x2 o = hv + al * 0.f + ia * 0.f + ga * 0.f + be * 0.f;
#else
            const f32x2 sv = snake_act2(hv, al, ia);
            const f32x2 o = __builtin_elementwise_fma(ga, sv, be) + sv;
#endif
            split2(o.x, o.y, xbp[0][2 * h + (r >> 1)], xbp[1][2 * h + (r >> 1)], xbp[2][2 * h + (r >> 1)]);
        };
        auto warm = [&](int it) __attribute__((always_inline)) {
            int64_t off = warm_lo + 256 * it + 4 * lane;
            off = off < x_last ? off : x_last;
            dma16_if(reinterpret_cast<const unsigned char*>(x + off), xs_lds, it, warm_n);
        };

#pragma unroll 1
        for (int it = 0; it <= G::NB; ++it) {
            const bool first = it == 0, last = it == G::NB;
            const int prev = blk == 0 ? G::NB - 1 : blk - 1;  // the block whose tiles sit in xa
            warm(it);
            // ---- steps 0, 1: first product of block `blk` (one hidden tile per slot) beside the activation of block `prev` ----
            if (!last) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    issue((h + G::PF) % G::NSTEP);
                    f32x4_t acc = bias1(blk, h);
#pragma unroll
                    for (int s = 0; s < G::NK; ++s) {
                        acc = kstep(acc, h, s, ap[s]);
                        if (s == G::NK / 2 - 1) act_pair(prev, h, 0);  // (block `prev` is garbage in the first iteration: unused)
                        if (s == G::NK - 1) act_pair(prev, h, 2);
                    }
                    xa[h] = acc;  // (both act_pair calls of this step have read the old tile)
                    step_sync();
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    issue((h + G::PF) % G::NSTEP);
                    act_pair(prev, h, 0);
                    act_pair(prev, h, 2);
                    step_sync();
                }
            }
            // ---- steps 2, 3: second product of block `prev` (half of the output tiles per slot) ---------------------------------
            if (!first) {
                bf16x8 xb[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xb[pl] = __builtin_bit_cast(bf16x8, u32x4{xbp[pl][0], xbp[pl][1], xbp[pl][2], xbp[pl][3]});
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    issue((2 + h + G::PF) % G::NSTEP);
#pragma unroll
                    for (int k = 0; k < G::CT / 2; ++k) yacc[G::CT / 2 * h + k] = kstep(yacc[G::CT / 2 * h + k], 2 + h, k, xb);
                    step_sync();
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    issue((2 + h + G::PF) % G::NSTEP);
                    step_sync();
                }
            }
            blk = blk + 1 == G::NB ? 0 : blk + 1;
        }
        WIDE_STAMP(2);
    }
    if (group == 0) {
#pragma unroll 1
        for (int i = 0; i < G::OFFSET_IT; ++i) idle_iteration();
    }
    // leave no LDS-DMA in flight behind the workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C>
int launch_wide(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int64_t rows, int frames, const char* name) {
    using G = WGeo<C>;
    static bool configured = false;
    if (!configured) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_wide_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        configured = true;
    }
    const int64_t tiles = ceil_div64(rows, 16);
    int64_t blocks = ceil_div64(tiles, 8);
    if (blocks > 256) blocks = 256;
    ProfScope prof(s, name, (double)rows * (14.0 * C + 16.0 * C * C), (double)rows * 8.0 * C);
    hipLaunchKernelGGL((conv_unit_wide_kernel<C>), dim3((unsigned)blocks), dim3(512), G::LDS, s, w, x, y, rows, frames);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

#ifdef L3AC_WIDE_STAMPS
extern "C" int l3ac_debug_wide_stamps(unsigned long long* out, int n) {  // diagnostic builds only (not part of the ABI)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

bool conv_unit_wide_supported(int c) { return c == 128 || c == 192 || c == 256; }

// x must not alias y (tiles read their neighbours' frames for the depth-wise taps)
int launch_conv_unit_wide(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames) {
    L3AC_REQUIRE(x != y && w.wide_img && batch > 0 && frames > 0, "conv_unit_wide: bad arguments");
    const int64_t rows = (int64_t)batch * frames;
    switch (w.c) {
        case 128: return launch_wide<128>(s, w, x, y, rows, frames, "conv_unit_wide_kernel<128>");
        case 192: return launch_wide<192>(s, w, x, y, rows, frames, "conv_unit_wide_kernel<192>");
        case 256: return launch_wide<256>(s, w, x, y, rows, frames, "conv_unit_wide_kernel<256>");
        default:
            l3ac_set_error("conv_unit_wide: C=%d not supported", w.c);
            return L3AC_EINVAL;
    }
}

// The cyclic weight stream, iteration b = 0 .. 4C/32 - 1 (indices mod 4C/32):
//     W1(tile 2(b+1)) | W1(tile 2(b+1) + 1) | W2(block b, output tiles 0 .. C/32 - 1) | W2(block b, output tiles C/32 .. C/16 - 1)
// every slot C/32 x 3 KB, every 1-KB plane fragment in MFMA A-operand order (lane = 16 g + i holds 8 bf16 at byte 16 lane):
//   W1(tile h), k step s, plane p:  W1[16 h + i][32 s + sigma(g, j)]
//   W2(block b), output tile ct, plane p:  W2[16 ct + i][32 b + sigma(g, j)]
std::vector<unsigned char> conv_unit_wide_image(const float* w1, const float* w2, int c) {
    const int h4 = 4 * c, nb = h4 / 32, nk = c / 32, ct_n = c / 16;
    std::vector<unsigned char> img((size_t)nb * 4 * nk * 3072, 0);
    auto put = [&](unsigned char* dst, const float* src_row_base, int64_t ld, int col0) {  // one k step / output tile: 3 planes x 1 KB
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 8; ++j) {
                    uint16_t pl[3];
                    split3_host(src_row_base[(int64_t)i * ld + col0 + wide_sigma(g, j)], pl);
                    for (int p = 0; p < 3; ++p) std::memcpy(dst + 1024 * p + 16 * (16 * g + i) + 2 * j, &pl[p], 2);
                }
    };
    unsigned char* out = img.data();
    for (int b = 0; b < nb; ++b) {
        for (int h = 0; h < 2; ++h) {
            const int tile = (2 * (b + 1) + h) % (2 * nb);
            for (int s = 0; s < nk; ++s, out += 3072) put(out, w1 + (size_t)16 * tile * c, c, 32 * s);
        }
        for (int ct = 0; ct < ct_n; ++ct, out += 3072) put(out, w2 + (size_t)16 * ct * h4, h4, 32 * b);
    }
    return img;
}
