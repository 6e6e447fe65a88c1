// Fused ConvUnit for the WIDE stages (C = 128 / 192 / 256), both channel contractions on the bf16 matrix cores at fp32
// accuracy ("bf16x3", split_bf16.hpp); reference l3ac/modules.py:10-41 + Residual (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// What it replaces: dwconv_ln_kernel -> gemm_split (C -> 4C, snake + GRN) -> gemm_split (4C -> C, + residual), i.e. three
// launches whose 4C-wide hidden tensor made two trips through the fabric (944 MB per unit at C = 256 for 256 x 1 s clips),
// whose A operands were re-split by every column block (4 to 16 times), and which ran the MFMA pipe ~40 % busy.
//
// Design (gfx950)
//   * One WAVE owns 32 frames end to end, as in conv_unit_split.hip: products are evaluated transposed (hidden channel
//     on the accumulator's rows = registers, frame on its lanes), so a hidden tile X^T (32 hidden x 32 frames) goes
//     through snake / GRN on the accumulator registers, is split there and is DIRECTLY the B operand of the second
//     product.  Nothing of the hidden tensor ever leaves the register file.
//   * Register file as the main store (512 KB per CU against 160 KB of LDS): a workgroup is 4 waves, ONE per SIMD, each
//     with up to 512 registers: the split LayerNorm output (3 planes x C/16 k steps x 4 = 0.75 C registers) and the
//     output accumulators (C/2 registers) of the wave's 32 frames stay resident for the whole unit.
//   * Weights: W1 / W2 come as ONE stream of fragment-ordered bf16x3 images in exactly the order the wave consumes them
//     (built at model load: conv_unit_wide_image), 12-KB slots = 4 k steps of W1 or 2 output tiles of W2.  All four
//     waves read the same stream, so it is staged once per workgroup: global_load_lds (LDS-DMA, no staging registers,
//     3 x 1 KB per wave per slot) into a ring of C/32 slots, NSTEP - 1 slots in flight behind a counted s_waitcnt vmcnt
//     and one raw s_barrier per slot.  The image (3.1 MB at C = 256) stays L2-resident: every CU walks it in step.
//   * Inside a hidden-tile iteration the first product of tile nt+1 (MFMA) is interleaved in program order with the
//     activation + split of tile nt (VALU), then the second product of tile nt runs: 192 MFMAs per iteration at C = 256,
//     the matrix pipe is the only pipe that is ever full.
//   * Tiles are 32 consecutive GLOBAL rows (clip boundaries handled by masking the depth-wise taps), so there are no
//     partial tiles except the very last one.
// Algorithmic work per frame: 16 C^2 + 14 C FLOP, 8 C bytes (x in, y out).
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "split_bf16.hpp"

#include <vector>

namespace {

template <int C>
struct WGeo {
    static constexpr int H4 = 4 * C;
    static constexpr int NT = H4 / 32;      // hidden tiles
    static constexpr int NS1 = C / 16;      // k steps of the first product
    static constexpr int CT = C / 32;       // output tiles
    static constexpr int KQ = C / 8;        // channel quads per lane half
    // depth-wise conv staging (per wave, inside the idle weight ring): chunks of 32 channels x 40 rows (38 halo'd frames,
    // one row of zeros, one spare), 128 B per row, XOR-swizzled 16-B slots
    static constexpr int XCH = 32;          // channels per chunk
    static constexpr int NCH = C / XCH;     // chunks
    static constexpr int XCHUNK = 40 * 128; // bytes of one staged chunk = 5 LDS-DMA pieces
    static constexpr int KS = 4;            // k steps of W1 per slot
    static constexpr int SLOT = KS * 3 * 1024;  // = 2 output tiles of W2 (2 x 2 steps x 3 planes x 1 KB)
    static constexpr int NA = NS1 / KS;     // slots of one W1 tile == slots of one W2 tile (= CT / 2)
    static constexpr int NSTEP = 2 * NA;    // slots per hidden-tile iteration == ring size
    static constexpr int PF = NSTEP - 1;    // slots in flight
    static constexpr int RING = NSTEP * SLOT;
    static constexpr int XREGION = RING / 4;                        // per-wave share of the ring during the staging phase
    static constexpr int XBUF = XREGION / XCHUNK < 4 ? XREGION / XCHUNK : 4;  // staged chunks in flight
    static constexpr int TOTAL = NT * NSTEP;  // slots of the whole stream
    static constexpr int WAIT = 3 * (PF - 2); // this wave's DMA instructions that may stay outstanding at a step's end
    // LDS (bytes): ring | alpha, 1/alpha, gamma, beta [4][H4] | b1 [H4] | b2 [C] | dw_w [7][C], dw_b, ln_w, ln_b
    static constexpr int OFF_P = RING;
    static constexpr int OFF_B1 = OFF_P + 4 * H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4 * 4;
    static constexpr int OFF_DW = OFF_B2 + C * 4;
    static constexpr int OFF_DUMP = OFF_DW + 10 * C * 4;   // 1 KB per wave: target of the cache-warming copies (never read)
    static constexpr int LDS = OFF_DUMP + 4 * 1024;
    static constexpr int WARM = (38 * C * 4 + 1023) / 1024; // 1-KB pieces of the 38 rows a tile needs
    static_assert(C % 64 == 0 && NS1 % KS == 0 && CT % 2 == 0 && NA == CT / 2, "bad geometry");
    static_assert(PF >= 3 && WAIT <= 63, "ring too small / vmcnt field too narrow");
    static_assert(XBUF >= 2 && 5 * XBUF <= 63 && XREGION % 16 == 0, "staging buffers");
    static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
    static_assert(WARM <= 2 * (NT - 1), "not enough hidden-tile iterations to warm the next tile's rows");
};

__device__ float g_zero_row[512];  // what out-of-clip depth-wise taps read (zero-initialised)

#ifdef L3AC_WIDE_STAMPS  // diagnostic build (tools/wide_stamps.py): s_memtime at the phase boundaries of every pass of wave 0
__device__ unsigned long long g_wide_stamps[256 * 16 * 8];
#define WIDE_STAMP(slot)                                                                                      \
    do {                                                                                                      \
        if (lane == 0 && wave == 0 && pass_no < 16)                                                           \
            g_wide_stamps[((size_t)blockIdx.x * 16 + pass_no) * 8 + (slot)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define WIDE_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// one 1-KB LDS-DMA piece: lane l copies 16 B from its own source pointer to lds_dst + 16 l (guide §5.7: M0 is written
// in the statement that uses it; the copy is invisible to hipcc's s_waitcnt bookkeeping and is counted by hand)
__device__ __forceinline__ void dma16(const unsigned char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the same, executed only while idx < limit (both wave-uniform).  The test and the branch live INSIDE the statement, so the
// surrounding code stays one basic block and hipcc keeps interleaving MFMA and VALU work across it.
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

template <int C>
__global__ __launch_bounds__(256, 1) void conv_unit_wide_kernel(const ConvUnitW w, const float* __restrict__ x,
                                                              float* __restrict__ y, const int64_t rows, const int frames) {
    using G = WGeo<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
    unsigned char* ring = smem_wide;
    float* Ps = reinterpret_cast<float*>(smem_wide + G::OFF_P);
    float* B1s = reinterpret_cast<float*>(smem_wide + G::OFF_B1);
    float* B2s = reinterpret_cast<float*>(smem_wide + G::OFF_B2);
    float* DWs = reinterpret_cast<float*>(smem_wide + G::OFF_DW);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 31;  // frame within the tile (MFMA column)
    const int lh = lane >> 5;

    // ---- the weight stream: this wave copies bytes [3072 wave, 3072 wave + 3072) of every slot ------------------
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned char* const src_lane = w.wide_img + 3072 * wave + 16 * lane;
    int dma_slot = 0;  // next slot of the stream to fetch (wave-uniform)
    auto issue = [&](int ring_pos) __attribute__((always_inline)) {
#ifdef L3AC_WIDE_NODMA  // timing experiment: no weight stream (wrong results)
        return;
#endif
        // the stream ends with the pass (the ring then serves the next pass's staging): past the last slot nothing is issued
        const unsigned char* src = src_lane + (int64_t)dma_slot * G::SLOT;
        const unsigned dst = ring_lds + (unsigned)(ring_pos * G::SLOT) + 3072u * (unsigned)wave;
        dma16_if(src, dst, dma_slot, G::TOTAL);
        dma16_if(src + 1024, dst + 1024u, dma_slot, G::TOTAL);
        dma16_if(src + 2048, dst + 2048u, dma_slot, G::TOTAL);
        ++dma_slot;
    };
    // end of a step: this wave's pieces of the slot after next have landed, then everybody's have, and everybody is done
    // reading the slot that the next step's DMA overwrites
    auto step_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");
#ifndef L3AC_WIDE_NOBAR  // timing experiment: no step barrier (racy, wrong results)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };
    // ---- parameters resident for the lifetime of the workgroup ---------------------------------------------------
    for (int i = tid; i < G::H4; i += 256) {
        Ps[i] = w.alpha[i];
        Ps[G::H4 + i] = w.inv_alpha[i];
        Ps[2 * G::H4 + i] = w.gamma[i];
        Ps[3 * G::H4 + i] = w.beta[i];
        B1s[i] = w.b1[i];
    }
    for (int i = tid; i < C; i += 256) {
        B2s[i] = w.b2[i];
        DWs[7 * C + i] = w.dw_b[i];
        DWs[8 * C + i] = w.ln_w[i];
        DWs[9 * C + i] = w.ln_b[i];
    }
    for (int i = tid; i < 7 * C; i += 256) DWs[i] = w.dw_w[i];
    __syncthreads();  // (plain loads above: hipcc drains them here, the DMA pieces with them)

    // One k step (16 of the reduction) = 6 plane products a_i . b_j, i + j <= 2, with the weight planes P[i] (A operand,
    // from the ring) held in three rotating register quads: each plane's next value (the following k step's, possibly in
    // the next slot) is fetched as soon as its last product of this step has issued — P[0] after 3 products, P[1] after 5,
    // P[2] after 6 — so every ds_read has >= 3 MFMAs (96 cycles) of cover and no fragment is ever waited for, at no extra
    // registers.  sched_barrier(0x6) lets only VALU / SALU instructions cross: the MFMA and ds_read order below is exactly
    // what runs, while the compiler stays free to spread the activation's VALU work between the MFMAs.
    // (Sum order inside a step: largest plane products first; each term still meets the fp32 accumulator once.)
    bf16x8 P[3];
    auto plane = [&](int ring_pos, int piece, int pl) __attribute__((always_inline)) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(ring + ring_pos * G::SLOT + piece * 3072 + 1024 * pl + 16 * lane);
    };
    auto kstep = [&](f32x16_t acc, const bf16x8 (&bq)[3], bool more, int next_pos, int next_piece) __attribute__((always_inline)) -> f32x16_t {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[0], bq[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[0], bq[1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[0], bq[2], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        if (more) P[0] = plane(next_pos, next_piece, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[1], bq[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[1], bq[1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        if (more) P[1] = plane(next_pos, next_piece, 1);
        __builtin_amdgcn_sched_barrier(0x6);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P[2], bq[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        if (more) P[2] = plane(next_pos, next_piece, 2);
        __builtin_amdgcn_sched_barrier(0x6);
        return acc;
    };

    const int64_t n_tiles = (rows + 31) / 32;
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // every wave of the block runs the same number of passes (block barriers inside)
    int pass_no = 0;
    (void)pass_no;
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_tiles; base += tile_stride, ++pass_no) {
        WIDE_STAMP(0);
        const int64_t row0 = (base + wave) * 32;       // first global row of this wave's tile
        const int64_t row = row0 + lj;                // this lane's global row
        const bool row_ok = row < rows;
        const int t = row_ok ? (int)(row % frames) : 0;  // frame inside its clip

        // ---- depth-wise conv k7 + LayerNorm for this lane's frame, channels 8q + 4 lh + {0..3} (modules.py:33-35) ----
        // The 38 rows a tile needs (32 frames + 3 either side) are fetched ONCE, whole 128-B lines at a time, by LDS-DMA into
        // this wave's quarter of the (idle) weight ring, 32 channels per chunk, XBUF chunks in flight; the 7 taps of a lane
        // are then LDS reads.  LDS image of a chunk: [40 rows][8 slots of 16 B], slot = quad ^ swz(row) (conflict-free for
        // 16 consecutive rows); row 38 holds zeros and is what a tap outside the lane's clip reads (zero padding,
        // modules.py:19-20), so the loop has neither masks nor branches.
        bf16x8 ap[G::NS1][3];
        {
            const unsigned xreg_lds = ring_lds + (unsigned)(G::XREGION * wave);
            const unsigned char* xreg = ring + G::XREGION * wave;
            const float* xsrc[5];  // this lane's source of piece p of chunk 0 (chunk c: + 32 c floats; the zero row is long enough)
#pragma unroll
            for (int pc = 0; pc < 5; ++pc) {
                const int u = 64 * pc + lane, r = u >> 3, quad = (u & 7) ^ ((r ^ (r >> 3)) & 7);
                const int64_t rg = row0 - 3 + r;
                xsrc[pc] = (r < 38 && rg >= 0 && rg < rows) ? x + rg * C + 4 * quad : g_zero_row + 4 * quad;
            }
            auto stage = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                for (int pc = 0; pc < 5; ++pc)
                    dma16(reinterpret_cast<const unsigned char*>(xsrc[pc] + G::XCH * c), xreg_lds + (unsigned)((c % G::XBUF) * G::XCHUNK + 1024 * pc));
            };
#pragma unroll
            for (int c = 0; c < G::XBUF && c < G::NCH; ++c) stage(c);
            // LDS byte offsets of this lane's reads inside a chunk: tap row (or the zero row) x the lane's two quads per q
            int xoff[7][4];
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const bool ok = row_ok && t + tap - 3 >= 0 && t + tap - 3 < frames;
                const int r = ok ? lj + tap : 38;
                const int sw = (r ^ (r >> 3)) & 7;
#pragma unroll
                for (int ql = 0; ql < 4; ++ql) xoff[tap][ql] = 128 * r + 16 * ((2 * ql + lh) ^ sw);
            }
            float a[4 * G::KQ];
            float s1 = 0.f;
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                // chunk c has landed once at most the younger chunks' pieces are outstanding
                constexpr int XB = G::XBUF;
                const int younger = (c + XB < G::NCH ? XB - 1 : G::NCH - 1 - c) * 5;
                if (younger >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                else if (younger == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (younger == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned char* xb_ = xreg + (c % XB) * G::XCHUNK;
#pragma unroll
                for (int ql = 0; ql < 4; ++ql) {
                    // (fully unrolled so that a[] and yacc stay in registers: the fence keeps hipcc from hoisting every quad's
                    // 14 LDS reads to the top of the chunk)
                    if (ql & 1) __builtin_amdgcn_sched_barrier(0);
                    const int q = 4 * c + ql;
                    const int k0 = 8 * q + 4 * lh;
                    float4 acc = *reinterpret_cast<const float4*>(DWs + 7 * C + k0);
#pragma unroll
                    for (int tap = 0; tap < 7; ++tap) {
                        const float4 xv = *reinterpret_cast<const float4*>(xb_ + xoff[tap][ql]);
                        const float4 wv = *reinterpret_cast<const float4*>(DWs + tap * C + k0);
                        acc.x = fmaf(xv.x, wv.x, acc.x);
                        acc.y = fmaf(xv.y, wv.y, acc.y);
                        acc.z = fmaf(xv.z, wv.z, acc.z);
                        acc.w = fmaf(xv.w, wv.w, acc.w);
                    }
                    a[4 * q] = acc.x; a[4 * q + 1] = acc.y; a[4 * q + 2] = acc.z; a[4 * q + 3] = acc.w;
                    s1 += (acc.x + acc.y) + (acc.z + acc.w);
                }
                if (c + XB < G::NCH) {  // refill the buffer just consumed (this wave's reads of it have returned: they fed the fmas)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    stage(c + XB);
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            const float mean = s1 / (float)C;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4 * G::KQ; ++i) {
                const float d = a[i] - mean;
                s2 = fmaf(d, d, s2);
            }
            s2 += __shfl_xor(s2, 32, 64);
            const float rstd = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
            // LayerNorm affine, then split: k step s of lane half lh = a[8s .. 8s+7] = channels split_sigma(s, lh, j)
#pragma unroll
            for (int s = 0; s < G::NS1; ++s) {
                unsigned p[3][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int idx = 8 * s + 2 * j + e;
                        const int ch = 8 * (idx >> 2) + 4 * lh + (idx & 3);
                        v[e] = row_ok ? (a[idx] - mean) * rstd * DWs[8 * C + ch] + DWs[9 * C + ch] : 0.f;
                    }
                    split2(v[0], v[1], p[0][j], p[1][j], p[2][j]);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ap[s][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
            }
        }
        WIDE_STAMP(1);
        // ---- output accumulators start at the pw_conv2 bias ------------------------------------------------------
        f32x16_t yacc[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ct][r] = B2s[32 * ct + rowmap(r, lh)];

        // ---- the weight stream starts over: every wave is done with its staging area (= the ring), then the first PF slots are
        // requested and the first two awaited
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_slot = 0;
#pragma unroll
        for (int j = 0; j < G::PF; ++j) issue(j);

        // the slot at ring position 0 (and the one after it) must have landed: first pass = the prologue's copies,
        // later passes = guaranteed by the previous pass's last step
        step_sync();

        auto bias1 = [&](int nt) __attribute__((always_inline)) -> f32x16_t {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = B1s[32 * nt + rowmap(r, lh)];
            return acc;
        };
        // snake + GRN (normaliser == 1; layers.py:29-33, :112-115) on rows (r, r + 1) of hidden tile nt, in place, and the
        // bf16x3 split of the pair into element (r & 7) / 2 of k step r / 8 of the second product's B operand
        unsigned xbp[2][3][4];
        auto act_pair = [&](f32x16_t& xacc, int nt, int r) __attribute__((always_inline)) {
#ifdef L3AC_WIDE_NOACT  // timing experiment: split only, no snake / GRN (wrong results)
            split2(xacc[r], xacc[r + 1], xbp[r >> 3][0][(r & 7) >> 1], xbp[r >> 3][1][(r & 7) >> 1], xbp[r >> 3][2][(r & 7) >> 1]);
            return;
#endif
            const float* pp = Ps + 32 * nt + rowmap(r, lh);
            const f32x2 al = *reinterpret_cast<const f32x2*>(pp);
            const f32x2 ia = *reinterpret_cast<const f32x2*>(pp + G::H4);
            const f32x2 ga = *reinterpret_cast<const f32x2*>(pp + 2 * G::H4);
            const f32x2 be = *reinterpret_cast<const f32x2*>(pp + 3 * G::H4);
            f32x2 hv;
            hv.x = xacc[r];
            hv.y = xacc[r + 1];
#ifdef L3AC_WIDE_SCALAR_ACT  // experiment: one-element-per-instruction activation (packed fp32 ops are dearer beside MFMAs)
            f32x2 o;
            {
                const float s0 = snake_act(hv.x, al.x, ia.x), s1v = snake_act(hv.y, al.y, ia.y);
                o.x = fmaf(ga.x, s0, be.x) + s0;
                o.y = fmaf(ga.y, s1v, be.y) + s1v;
            }
#else
            const f32x2 sv = snake_act2(hv, al, ia);
            const f32x2 o = __builtin_elementwise_fma(ga, sv, be) + sv;
#endif
            split2(o.x, o.y, xbp[r >> 3][0][(r & 7) >> 1], xbp[r >> 3][1][(r & 7) >> 1], xbp[r >> 3][2][(r & 7) >> 1]);
        };
        // second product of hidden tile nt from the ring slots OFF .. OFF + NA - 1 (2 output tiles per slot); P holds the
        // first fragment on entry and, unless LAST_OF_PASS, the next slot's first fragment on exit
        auto second_product = [&](auto off_, auto last_) __attribute__((always_inline)) {
            constexpr int OFF = decltype(off_)::value;
            constexpr bool LAST_OF_PASS = decltype(last_)::value;
            bf16x8 xb[2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    xb[s][pl] = __builtin_bit_cast(bf16x8, u32x4{xbp[s][pl][0], xbp[s][pl][1], xbp[s][pl][2], xbp[s][pl][3]});
#pragma unroll
            for (int i = 0; i < G::NA; ++i) {
                issue((OFF + i + G::PF) % G::NSTEP);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) {  // piece = 2 (output tile within the slot) + k step
                    const bool last_piece = pc == 3;
                    // the slot after this one landed a step ago: its first fragment is fetched across the barrier
                    const int npos = !last_piece ? OFF + i : (i + 1 < G::NA ? OFF + i + 1 : G::NA);
                    yacc[2 * i + (pc >> 1)] = kstep(yacc[2 * i + (pc >> 1)], xb[pc & 1], !(LAST_OF_PASS && i + 1 == G::NA && last_piece), npos,
                                                    last_piece ? 0 : pc + 1);
                }
                step_sync();
            }
        };

        WIDE_STAMP(2);
        // ---- first product of hidden tile 0 (ring slots 0 .. NA-1): nothing to overlap with -------------------------
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) P[pl] = plane(0, 0, pl);
        f32x16_t xacc = bias1(0);
#pragma unroll
        for (int i = 0; i < G::NA; ++i) {
            issue((i + G::PF) % G::NSTEP);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
                xacc = kstep(xacc, ap[G::KS * i + ks], true, ks + 1 < G::KS ? i : i + 1, ks + 1 < G::KS ? ks + 1 : 0);  // i + 1 == NA: iteration 0
            step_sync();
        }

        WIDE_STAMP(3);
        // The NEXT pass's rows (38 x C floats, contiguous) are pulled towards the chip while this pass computes: two 1-KB LDS-DMA
        // copies per hidden-tile iteration into a dump area nobody reads.  Their only purpose is that the staging phase of
        // the next pass finds the lines in the Infinity Cache instead of waiting for HBM with the matrix cores idle.
        const int64_t next_row0 = row0 + tile_stride * 32;
        const int64_t warm_lo = (next_row0 - 3 < 0 ? 0 : next_row0 - 3) * C;  // in floats
        const int warm_n = next_row0 < rows + 3 ? G::WARM : 0;                // pieces to copy (0: no next pass)
        const int64_t x_last = rows * C - 4;                                  // last 16-B unit of x
        const unsigned dump_lds = ring_lds + (unsigned)(G::OFF_DUMP + 1024 * wave);
#pragma unroll 1
        for (int nt = 0; nt + 1 < G::NT; ++nt) {
            {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int64_t off = warm_lo + 256 * (2 * nt + e) + 4 * lane;
                    off = off < x_last ? off : x_last;
                    dma16_if(reinterpret_cast<const unsigned char*>(x + off), dump_lds, 2 * nt + e, warm_n);
                }
            }
            // ---- first product of tile nt+1 (slots NA .. 2NA-1) beside the activation of tile nt ----------------------
            f32x16_t xnext = bias1(nt + 1);
#pragma unroll
            for (int i = 0; i < G::NA; ++i) {
                issue((G::NA + i + G::PF) % G::NSTEP);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    constexpr int STEPS = G::KS * G::NA;  // k steps of this phase; 8 activation pairs are dealt over them
                    const int s = G::KS * i + ks;
                    const int npos = ks + 1 < G::KS ? G::NA + i : (i + 1 < G::NA ? G::NA + i + 1 : 0);
                    xnext = kstep(xnext, ap[s], true, npos, ks + 1 < G::KS ? ks + 1 : 0);
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr)
                        if (pr * STEPS / 8 == s) act_pair(xacc, nt, 2 * pr);
                }
                step_sync();
            }
            // ---- second product of tile nt (slots 0 .. NA-1) --------------------------------------------------------
            second_product(std::integral_constant<int, 0>{}, std::false_type{});
            xacc = xnext;
        }
        WIDE_STAMP(4);
        // ---- last hidden tile: activation alone, second product from slots NA .. 2NA-1 ------------------------------
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) act_pair(xacc, G::NT - 1, 2 * pr);
        second_product(std::integral_constant<int, G::NA>{}, std::true_type{});

        WIDE_STAMP(5);
        // ---- residual + store (xtract/nn/layers.py:59-62).  The accumulators hold, per lane, 4 channels of ONE frame for each of
        // the 4 C/32 (tile, group) pairs: stored directly that is 32 B per row per instruction.  Instead every 32-channel tile
        // goes through this wave's (now idle) quarter of the ring — written as [32 frames][8 slots of 16 B] with the staging
        // swizzle, read back row-major — so that the residual load and the store move whole 128-B lines, 8 rows per instruction.
        {
            unsigned char* tbuf = ring + G::XREGION * wave;
            const int er = lane >> 3, es = lane & 7;  // row within a group of 8, 16-B slot
            float4 xres[G::CT][4];
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t rr = row0 + 8 * i + er;
                    xres[ct][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rr < rows) xres[ct][i] = *reinterpret_cast<const float4*>(x + rr * C + 32 * ct + 4 * es);
                }
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                unsigned char* tb = tbuf + (ct & 1) * 4096;  // two tiles in flight
                const int swl = (lj ^ (lj >> 3)) & 7;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(tb + 128 * lj + 16 * ((2 * g + lh) ^ swl)) =
                        make_float4(yacc[ct][4 * g], yacc[ct][4 * g + 1], yacc[ct][4 * g + 2], yacc[ct][4 * g + 3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 8 * i + er;
                    const int64_t rr = row0 + r;
                    const float4 v = *reinterpret_cast<const float4*>(tb + 128 * r + 16 * (es ^ ((r ^ (r >> 3)) & 7)));
                    const float4 xr = xres[ct][i];
                    if (rr < rows)
                        *reinterpret_cast<float4*>(y + rr * C + 32 * ct + 4 * es) = make_float4(xr.x + v.x, xr.y + v.y, xr.z + v.z, xr.w + v.w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    WIDE_STAMP(6);
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
    const int64_t tiles = ceil_div64(rows, 32);
    int64_t blocks = ceil_div64(tiles, 4);
    if (blocks > 256) blocks = 256;
    ProfScope prof(s, name, (double)rows * (14.0 * C + 16.0 * C * C), (double)rows * 8.0 * C);
    hipLaunchKernelGGL((conv_unit_wide_kernel<C>), dim3((unsigned)blocks), dim3(256), G::LDS, s, w, x, y, rows, frames);
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

// The weight stream in consumption order: W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W1(NT-1) W2(NT-2) | W2(NT-1), where
// W1(nt) / W2(nt) are the fragment-ordered tile images of conv_unit_w1_image / conv_unit_w2_image (conv_unit_split.hip):
//   W1(nt): k step s, plane p, lane half h, row r (32): 8 bf16 = W1[32 nt + r][split_sigma(s, h, j)]      (C/16 x 3 KB)
//   W2(nt): output tile ct, k step s (2), plane, half, row: 8 bf16 = W2[32 ct + r][32 nt + split_sigma(s, h, j)]  (C/32 x 6 KB)
std::vector<unsigned char> conv_unit_wide_image(const float* w1, const float* w2, int c) {
    const std::vector<unsigned char> i1 = conv_unit_w1_image(w1, c), i2 = conv_unit_w2_image(w2, c);
    const size_t t1 = (size_t)(c / 16) * 3072, t2 = (size_t)192 * c;
    const int nt_n = 4 * c / 32;
    std::vector<unsigned char> img;
    img.reserve(i1.size() + i2.size());
    auto put = [&](const std::vector<unsigned char>& v, size_t off, size_t n) { img.insert(img.end(), v.begin() + off, v.begin() + off + n); };
    put(i1, 0, t1);
    for (int nt = 0; nt + 1 < nt_n; ++nt) {
        put(i1, (size_t)(nt + 1) * t1, t1);
        put(i2, (size_t)nt * t2, t2);
    }
    put(i2, (size_t)(nt_n - 1) * t2, t2);
    return img;
}
