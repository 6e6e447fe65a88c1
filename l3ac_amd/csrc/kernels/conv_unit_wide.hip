// ConvUnit of the WIDE stages (C = 128 / 192 / 256; round 4: also C = 96, see WGeo::WG_PER_CU / FRONT) with the hidden tensor kept in registers, both channel contractions on the
// bf16 matrix cores at fp32 accuracy ("bf16x3", split_bf16.hpp); reference l3ac/modules.py:10-41 + Residual
// (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// What it replaces: dwconv_ln_kernel -> gemm_split (C -> 4C, snake + GRN) -> gemm_split (4C -> C, + residual), whose 4C-wide
// hidden tensor made two trips through the fabric (944 MB per unit at C = 256 for 256 x 1 s clips) and whose A operands were
// re-split by every column block (4 to 16 times).
//
// Two kernels:
//   dwconv_ln_split_kernel   memory-bound front end: depth-wise conv k7 + LayerNorm, the result split ONCE into its three bf16
//                            planes and written in the MFMA fragment order of the main kernel (6 B per element).
//   conv_unit_wide_kernel    One WAVE owns 32 frames end to end, as in conv_unit_split.hip: products are evaluated transposed
//                            (hidden channel on the accumulator's rows = registers, frame on its lanes), so a hidden tile X^T
//                            (32 hidden x 32 frames) goes through snake / GRN on the accumulator registers, is split there
//                            and is DIRECTLY the B operand of the second product: nothing of the hidden tensor ever leaves
//                            the register file.  The register file is the main store (512 KB per CU against 160 KB of LDS):
//                            a workgroup is 4 waves, ONE per SIMD with up to 512 registers, holding the split LayerNorm
//                            output (3 planes x C/16 k steps x 4 = 0.75 C registers) and the output accumulators (C/2) of
//                            its 32 frames for the whole unit.  W1 / W2 come as ONE stream of fragment-ordered bf16x3 images
//                            in consumption order (conv_unit_wide_image), 12-KB slots = 4 k steps of W1 or 2 output tiles of
//                            W2, staged once per workgroup by LDS-DMA (global_load_lds, 3 x 1 KB per wave per slot) into a
//                            ring of C/32 slots, C/32 - 1 in flight behind a counted s_waitcnt vmcnt and one raw s_barrier
//                            per slot; the stream wraps around from one tile to the next.  The image (3.1 MB at C = 256)
//                            stays L2-resident: every CU walks it in step.  The first product of hidden tile nt+1 (MFMA) is
//                            interleaved in program order with the activation + split of tile nt (VALU).
// Why the front end is a kernel of its own: with one wave per SIMD nothing covers a load phase, and measured inside the main
// kernel (tools/wide_stamps.py on tools/experiments/conv_unit_wide_v1.hip (retired: git show 1a7dadb:tools/experiments/conv_unit_wide_v1.hip), conv_unit_wide_v2.hip) the depth-wise conv's 38
// rows per tile cost 17-32 % of a pass however they were fetched; as a full-occupancy streaming kernel the same work is
// HBM-bound and the main kernel's prologue becomes 48 coalesced 16-B loads per lane.
// Tiles are 32 consecutive GLOBAL rows (clip boundaries only matter to the front end's taps).
// Algorithmic work per frame: 16 C^2 + 14 C FLOP; bytes: front end 4 C in + 6 C out, main kernel 6 C + 4 C in, 4 C out.
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "ring_common.hpp"
#include "split_bf16.hpp"
#define L3AC_DIAG_UNIT_CONV_UNIT_WIDE
#include "diag.hpp"

#include <vector>

namespace {

template <int C>
struct WGeo {
    static constexpr int H4 = 4 * C;
    static constexpr int NT = H4 / 32;      // hidden tiles
    static constexpr int NS1 = C / 16;      // k steps of the first product
    static constexpr int CT = C / 32;       // output tiles of 32 channels (epilogue)
    static constexpr int RT = C / 16;       // output row tiles of 16 channels (second product)
    static constexpr int KQ = C / 8;        // channel quads per lane half
    static constexpr int KS = C >= 256 ? 8 : C % 64 == 0 ? 4 : 3;  // pieces (weight fragments of 3 planes x 1 KB = 12 MFMAs) per ring slot: the
                                                 // step barrier, the DMA statement and the counted wait come once per slot, so fewer, larger slots
                                                 // where the ring still holds four of them.  Every copying wave moves whole 1-KB blocks, 3 or 6 of
                                                 // them: all four waves when KS is a multiple of 4; C = 96 (six pieces per product) has slots of
                                                 // three pieces = 9 KB, copied by the first three waves
    static constexpr int SLOT = KS * 3 * 1024;
    static constexpr int NA = NS1 / KS;     // slots of one W1 tile == slots of one W2 tile
    static constexpr int DMA_N = KS == 8 ? 6 : 3;  // 1-KB LDS-DMA instructions per copying wave and slot
    static constexpr int COPY_WAVES = SLOT / (DMA_N * 1024);
    // workgroups per CU.  C = 96 has 3.4 vector instructions per MFMA where two hide behind one: with ONE wave per SIMD the hidden-tile loop
    // ran at 32 cycles per MFMA (stamps: 4.6 k cycles per hidden tile against 2.3 k of products).  A SIMD issues vector instructions of TWO
    // waves at 1.8 x the rate of one (tools/probes issue_probe2), and the second workgroup's products run under the first one's activation
    static constexpr int WG_PER_CU = C <= 96 ? 2 : 1;
    // the front end (depth-wise conv + LayerNorm + split) inside the main kernel's pass prologue instead of dwconv_ln_split_kernel: with two
    // workgroups per CU the other workgroup's loop covers its loads, and at C = 96 the separate kernel's 10 C bytes per frame through HBM
    // cost half of what the main kernel does (0.68 against 1.12 ms at 256 clips)
    static constexpr bool FRONT = WG_PER_CU == 2;
    // units handed out by a counter instead of lock-step passes (conv_unit_wide_kernel, 'DYN')
    static constexpr bool DYN = WG_PER_CU == 2;
    static constexpr int HALF_POS = 49152 / SLOT;  // ring positions per opaque LDS base (fragment offsets must fit 16 bits)
    static constexpr int NSTEP = 2 * NA;    // slots per hidden-tile iteration == ring size
    static constexpr int PF = NSTEP - 1;    // slots in flight
    static constexpr int RING = NSTEP * SLOT;
    static constexpr int TOTAL = NT * NSTEP;  // slots of the whole stream
    static constexpr int AP_AGPR_FROM = C >= 256 ? NS1 - 6 : NS1;  // k steps of the LayerNorm operand kept in AGPRs
    static constexpr int WAIT = DMA_N * (PF - 2);  // this wave's DMA instructions that may stay outstanding at a step's end
    // LDS (bytes): alpha, 1/alpha, gamma, beta [4][H4] | b1 [H4] | b2 [C] | ring | epilogue transposition buffers.  The tables
    // come first so that their reads (one per activation pair, the tile index in the address) reach with the 16-bit
    // immediate of ds_read from one lane base
    static constexpr int OFF_P = 0;
    static constexpr int OFF_B1 = OFF_P + 4 * H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4 * 4;
    static constexpr int OFF_RING = OFF_B2 + C * 4;
    static constexpr int OFF_TB = OFF_RING + RING;            // epilogue transposition buffers: 2 x 4 KB per wave
    static constexpr int OFF_NEXT = OFF_TB + 4 * 8192;        // (DYN) the next unit of the workgroup, written by thread 0
    static constexpr int LDS = OFF_NEXT + (WG_PER_CU == 2 ? 16 : 0);
    static_assert(C % 32 == 0 && NS1 % KS == 0 && NS1 % 2 == 0 && SLOT % (DMA_N * 1024) == 0 && COPY_WAVES <= 4 && NSTEP <= 2 * HALF_POS, "bad geometry");
    static_assert(PF >= 3 && WAIT <= 63, "ring too small / vmcnt field too narrow");
    static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
};


__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// this wave's quarter of one ring slot: N 1-KB LDS-DMA pieces, lane l copying 16 B from base + 16 l + 1024 i to lds_dst + 16 l + 1024 i.
// One statement = one M0 write for all pieces (the instruction offset is added to the global AND the LDS address; both are
// passed pre-biased by +3072 when N = 6 so that the offsets fit the signed 13-bit field), a wave-uniform 64-bit base in SGPRs
// and one loop-invariant 32-bit lane offset: no vector address arithmetic per piece (guide 5.7: M0 is written in the statement
// that uses it; the copies are invisible to hipcc's s_waitcnt bookkeeping and are counted by hand).
// NT: non-temporal policy on the copies (aux = nt).  For a stream that ONE workgroup per XCD or so reads once from beyond L2 (the
// half-tile form: a single clip) the copies land sooner (guide, price list 'nt-weights'); never for the batch form, whose 256
// workgroups re-read the image from L2 in step.
template <int N, bool NT = false>
__device__ __forceinline__ void dma_slot_quarter(const unsigned char* base, unsigned lane_off, unsigned lds_dst) {
    static_assert(N == 3 || N == 6, "3 KB or 6 KB per wave");
    unsigned keep;
#define L3AC_DMA3(POL)                                                     \
    asm volatile(                                                          \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"             \
        "global_load_lds_dwordx4 %1, %2" POL "\n\t"                        \
        "global_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t"            \
        "global_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\t"            \
        "s_mov_b32 m0, %0"                                                 \
        : "=&s"(keep)                                                      \
        : "v"(lane_off), "s"(base), "s"(lds_dst)                           \
        : "memory")
#define L3AC_DMA6(POL)                                                     \
    asm volatile(                                                          \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"             \
        "global_load_lds_dwordx4 %1, %2 offset:-3072" POL "\n\t"           \
        "global_load_lds_dwordx4 %1, %2 offset:-2048" POL "\n\t"           \
        "global_load_lds_dwordx4 %1, %2 offset:-1024" POL "\n\t"           \
        "global_load_lds_dwordx4 %1, %2" POL "\n\t"                        \
        "global_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t"            \
        "global_load_lds_dwordx4 %1, %2 offset:2048" POL "\n\t"            \
        "s_mov_b32 m0, %0"                                                 \
        : "=&s"(keep)                                                      \
        : "v"(lane_off), "s"(base + 3072), "s"(lds_dst + 3072u)            \
        : "memory")
    if constexpr (N == 3) {
        if constexpr (NT) L3AC_DMA3(" nt"); else L3AC_DMA3("");
    } else {
        if constexpr (NT) L3AC_DMA6(" nt"); else L3AC_DMA6("");
    }
#undef L3AC_DMA3
#undef L3AC_DMA6
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// ---- the activation of one PAIR of hidden rows (two elements per lane), cut into stages of at most three single-issue vector
// instructions so that the kernel can place them one by one into the gaps behind its MFMAs (one wave per SIMD: ~5
// single-issue instructions hide behind a 32-cycle MFMA; packed-fp32 instructions are an anti-lever there — guide,
// 'price of one filler beside MFMAs' — so every operation is written per element and the file is built with
// -fno-slp-vectorize).  The arithmetic is device_math.hpp's snake_act2 / sin_squared2 followed by GRN with normaliser 1 and
// split_bf16.hpp's split2, operation for operation: the results are bit-identical to those helpers'.
struct ActPair {
    float h[2], t[2], n[2], r[2], z[2], p[2], c[2], o[2], sv[2];
    float al[2], ia[2], ga[2], be[2];
    int ni[2];
    unsigned p0, p1;
};
constexpr int ACT_STAGES = 27;
// tab: this lane's row of the interleaved parameter table for the pair (alpha, alpha', 1/alpha, 1/alpha' | gamma, gamma', beta,
// beta'); h0/h1: the pair's pre-activations; out0..2: the pair's packed bf16 planes
template <int T>
__device__ __forceinline__ void act_stage(ActPair& a, const float* tab, float h0, float h1, unsigned& out0, unsigned& out1, unsigned& out2) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
#define L3AC_BOTH(expr) _Pragma("unroll") for (int e = 0; e < 2; ++e) { expr; }
    if constexpr (T == 0) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(tab);
        a.al[0] = v.x, a.al[1] = v.y, a.ia[0] = v.z, a.ia[1] = v.w;
    } else if constexpr (T == 1) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(tab + 4);
        a.ga[0] = v.x, a.ga[1] = v.y, a.be[0] = v.z, a.be[1] = v.w;
    } else if constexpr (T == 2) {
        a.h[0] = h0, a.h[1] = h1;
    } else if constexpr (T == 3) {
        L3AC_BOTH(a.t[e] = a.al[e] * a.h[e])
    } else if constexpr (T == 4) {
        L3AC_BOTH(a.t[e] = __builtin_amdgcn_fmed3f(a.t[e], -SIN2_ARG_MAX, SIN2_ARG_MAX))
    } else if constexpr (T == 5) {
        L3AC_BOTH(a.n[e] = a.t[e] * 0.636619772367581343f)
    } else if constexpr (T == 6) {
        L3AC_BOTH(a.n[e] = __builtin_rintf(a.n[e]))
    } else if constexpr (T == 7) {
        L3AC_BOTH(a.r[e] = __builtin_fmaf(a.n[e], -1.57079637050628662109375f, a.t[e]))
    } else if constexpr (T == 8) {
        L3AC_BOTH(a.r[e] = __builtin_fmaf(a.n[e], 4.37113900018624283e-8f, a.r[e]))
    } else if constexpr (T == 9) {
        L3AC_BOTH(a.z[e] = a.r[e] * a.r[e])
    } else if constexpr (T == 10) {
        L3AC_BOTH(a.p[e] = __builtin_fmaf(a.z[e], -1.9515295891e-4f, 8.3321608736e-3f))
    } else if constexpr (T == 11) {
        L3AC_BOTH(a.p[e] = __builtin_fmaf(a.p[e], a.z[e], -1.6666654611e-1f))
    } else if constexpr (T == 12) {
        L3AC_BOTH(a.p[e] = a.p[e] * a.z[e])
    } else if constexpr (T == 13) {
        L3AC_BOTH(a.p[e] = __builtin_fmaf(a.p[e], a.r[e], a.r[e]))  // sin(r), |r| <= pi/4
    } else if constexpr (T == 14) {
        L3AC_BOTH(a.p[e] = a.p[e] * a.p[e])
    } else if constexpr (T == 15) {
        L3AC_BOTH(a.c[e] = __builtin_fmaf(a.p[e], -2.0f, 1.0f))  // cos(2r)
    } else if constexpr (T == 16) {
        L3AC_BOTH(a.ni[e] = (int)a.n[e])
    } else if constexpr (T == 17) {
        L3AC_BOTH(a.ni[e] = (int)((unsigned)a.ni[e] << 31))
    } else if constexpr (T == 18) {
        L3AC_BOTH(a.c[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, a.c[e]) ^ a.ni[e]))  // cos(2u) = (-1)^n cos(2r)
    } else if constexpr (T == 19) {
        L3AC_BOTH(a.c[e] = __builtin_fmaf(a.c[e], -0.5f, 0.5f))  // sin(u)^2 = (1 - cos 2u) / 2
    } else if constexpr (T == 20) {
        L3AC_BOTH(a.sv[e] = __builtin_fmaf(a.ia[e], a.c[e], a.h[e]))  // snake
    } else if constexpr (T == 21) {
        L3AC_BOTH(a.o[e] = __builtin_fmaf(a.ga[e], a.sv[e], a.be[e]))  // GRN, normaliser 1
    } else if constexpr (T == 22) {
        L3AC_BOTH(a.o[e] = a.o[e] + a.sv[e])
    } else if constexpr (T == 23) {
        const f32x2_t v = {a.o[0], a.o[1]};
        a.p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        out0 = a.p0;
        a.t[0] = __builtin_bit_cast(float, a.p0 << 16), a.t[1] = __builtin_bit_cast(float, a.p0 & 0xffff0000u);
    } else if constexpr (T == 24) {
        L3AC_BOTH(a.r[e] = a.o[e] - a.t[e])
        const f32x2_t v = {a.r[0], a.r[1]};
        a.p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        out1 = a.p1;
    } else if constexpr (T == 25) {
        a.t[0] = __builtin_bit_cast(float, a.p1 << 16), a.t[1] = __builtin_bit_cast(float, a.p1 & 0xffff0000u);
    } else {
        static_assert(T == ACT_STAGES - 1, "stage list out of step with ACT_STAGES");
        L3AC_BOTH(a.z[e] = a.r[e] - a.t[e])
        const f32x2_t v = {a.z[0], a.z[1]};
        out2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    }
#undef L3AC_BOTH
}

// one of the six plane products of a fragment pair, in mfma_split's order (split_bf16.hpp), on v_mfma_f32_16x16x32_bf16: D[16 x 16]
// += A[16 x 32] B[32 x 16]; A: lane (row = lane & 15, k group = lane >> 4) holds 8 k values; B: lane (column = lane & 15, k group);
// D: lane (column = lane & 15), registers = rows 4 (lane >> 4) .. + 3.  (Why this shape: DESIGN.md 3.1, 'MFMA shape'.)
template <int M>
__device__ __forceinline__ f32x4_t mfma_plane(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4_t acc) {
    constexpr int IA[6] = {2, 1, 0, 1, 0, 0}, IB[6] = {0, 1, 2, 0, 1, 0};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[IA[M]], b[IB[M]], acc, 0, 0, 0);
}

// The front end inside the main kernel (WGeo::FRONT): LayerNorm(dw_conv7(x)) of this wave's frames, computed directly in the B-operand
// layout — lane (frame n = lane & 15, k group g = lane >> 4) owns channels 32 b + 8 g .. + 7 of its frame for every k block b, so the
// conv is 7 taps x 2 x 16 B of the lane's own channel run per block (rows of a neighbouring clip are the conv's zero padding, as in
// dwconv_ln_split_kernel), the LayerNorm sums are C/4 registers plus the four k groups of the frame (rows_sum, a fixed order: the same
// bits for a clip alone and inside any batch), and the split planes are the operand registers themselves.
// one output channel of the depth-wise conv for 16 consecutive frames held one per lane of a 16-lane row: acc += sum over taps of
// w[tap] * x[frame + tap - 3].  A neighbouring frame is a neighbouring LANE (DPP row shift); the three frames beyond either end of the row
// come from the registers that hold the previous / next 16 frames (lanes 13..15 of `prev`, lanes 0..2 of `next`) by the complementary shift,
// written only into the bank of the boundary lanes.  A DPP lane whose source lies outside the row is switched off (no bound_ctrl), so each
// lane executes exactly its seven fused multiply-adds, in tap order: the same bits as the per-tap loads of the general path below.
// (s_nop: a vector write of a DPP source needs two wait states before the DPP read, and hipcc does not see into the statement.)
__device__ __forceinline__ float dw_taps_dpp(float acc, float own, float prev, float next, const float (&w)[7]) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %1, %4 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %4 row_shl:13 row_mask:0xf bank_mask:0x1\n\t"
        "v_fmac_f32_dpp %0, %1, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %5 row_shl:14 row_mask:0xf bank_mask:0x1\n\t"
        "v_fmac_f32_dpp %0, %1, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %6 row_shl:15 row_mask:0xf bank_mask:0x1\n\t"
        "v_fmac_f32 %0, %1, %7\n\t"
        "v_fmac_f32_dpp %0, %1, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %8 row_shr:15 row_mask:0xf bank_mask:0x8\n\t"
        "v_fmac_f32_dpp %0, %1, %9 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %9 row_shr:14 row_mask:0xf bank_mask:0x8\n\t"
        "v_fmac_f32_dpp %0, %1, %10 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %10 row_shr:13 row_mask:0xf bank_mask:0x8"
        : "+v"(acc)
        : "v"(own), "v"(prev), "v"(next), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]));
    return acc;
}

// The front end inside the main kernel (WGeo::FRONT): LayerNorm(dw_conv7(x)) of this wave's frames, computed directly in the B-operand
// layout — lane (frame n = lane & 15, k group g = lane >> 4) owns channels 32 b + 8 g .. + 7 of its frame for every k block b.  The
// LayerNorm sums are C/4 registers plus the four k groups of the frame (rows_sum, a fixed order: the same bits for a clip alone and inside
// any batch), and the split planes are the operand registers themselves.  The conv has two forms with the same bits per frame:
//   general   7 taps x 2 x 16 B of the lane's own channel run per block, rows of a neighbouring clip being the conv's zero padding (as in
//             dwconv_ln_split_kernel): 28 loads per block and frame half;
//   interior  (wave-uniform test: the tile and three frames either side lie inside one clip — all but ~1 % of the tiles of a 2700-frame
//             clip) every row is loaded ONCE and the taps are lane shifts (dw_taps_dpp): 8 row loads per block instead of 28.
template <int C, int FH, int N>
__device__ __forceinline__ void wide_front(const ConvUnitW& w, const float* __restrict__ x, const int64_t rows, const int frames, const int64_t row0,
                                           const bool tile_ok, const int ln, const int lg, bf16x8 (&ap)[N][3]) {
    constexpr int KB = C / 32;
    static_assert(N == (FH == 2 ? 2 * KB : KB), "operand planes of FH frame halves");
    float cv[FH][KB][8];
    bool interior = false;
    if (tile_ok) {
        const int t0 = (int)(row0 % frames);
        interior = t0 >= 3 && t0 + 16 * FH + 3 <= frames;  // (inside one clip, hence inside the tensor)
    }
    if (__builtin_amdgcn_readfirstlane((int)interior)) {
        // every row piece of the tile is requested before the first tap (round 6: the loads were issued block by block behind lane-dependent
        // branches, three memory round trips per pass).  Of the frame groups either side only lanes 13..15 / 0..2 are used; the other lanes
        // re-request their own row of the tile's first group (the same lines: no extra traffic, no branch).
        f32x4_t grp[KB][FH + 2][2];
        const float* const px0 = x + (row0 - 16 + ln) * C + 8 * lg;  // frame group -1: its lanes 13..15 are the left halo
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int gi = 0; gi < FH + 2; ++gi) {
                const bool wanted = gi == 0 ? ln >= 13 : gi == FH + 1 ? ln <= 2 : true;
                const float* const q = px0 + 32 * b + (wanted ? (int64_t)gi * 16 * C : (int64_t)16 * C);
#pragma unroll
                for (int h = 0; h < 2; ++h) grp[b][gi][h] = *reinterpret_cast<const f32x4_t*>(q + 4 * h);
            }
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const int c0 = 32 * b + 8 * lg;
            f32x4_t wt[7][2], bs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bs[h] = *reinterpret_cast<const f32x4_t*>(w.dw_b + c0 + 4 * h);
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) wt[tap][h] = *reinterpret_cast<const f32x4_t*>(w.dw_w + tap * C + c0 + 4 * h);
            }
#pragma unroll
            for (int fh = 0; fh < FH; ++fh)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float wv[7] = {wt[0][h][e], wt[1][h][e], wt[2][h][e], wt[3][h][e], wt[4][h][e], wt[5][h][e], wt[6][h][e]};
                        cv[fh][b][4 * h + e] = dw_taps_dpp(bs[h][e], grp[b][fh + 1][h][e], grp[b][fh][h][e], grp[b][fh + 2][h][e], wv);
                    }
        }
    } else {
#pragma unroll
        for (int fh = 0; fh < FH; ++fh) {
            const int64_t r = row0 + 16 * fh + ln;
            const bool ok = tile_ok && r < rows;
            const int64_t rc = ok ? r : 0;
            const int t = (int)(rc % frames);
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int c0 = 32 * b + 8 * lg;
                f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(w.dw_b + c0), a1 = *reinterpret_cast<const f32x4_t*>(w.dw_b + c0 + 4);
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const int tt = t + tap - 3;
                    const bool in_clip = ok && tt >= 0 && tt < frames;
                    const float* px = x + (rc + (in_clip ? tap - 3 : 0)) * C + c0;
                    const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(px), x1 = *reinterpret_cast<const f32x4_t*>(px + 4);
                    const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(w.dw_w + tap * C + c0), w1 = *reinterpret_cast<const f32x4_t*>(w.dw_w + tap * C + c0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a0[e] = fmaf(in_clip ? x0[e] : 0.f, w0[e], a0[e]);
                        a1[e] = fmaf(in_clip ? x1[e] : 0.f, w1[e], a1[e]);
                    }
                }
                // (left alone hipcc requests the 126 row and weight pieces of all six (frame half, block) pairs first, spills them as they
                // arrive — s_waitcnt vmcnt(0) + scratch_store after every load — and the ~1.4 % of the tiles that touch a clip boundary hold
                // their workgroup's slot barriers for most of a pass)
                // (the sums pass through the statement: the taps are finished, and their operands dead, before the next pair's loads)
                asm volatile("" : "+v"(a0), "+v"(a1)::"memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) cv[fh][b][e] = a0[e], cv[fh][b][4 + e] = a1[e];
            }
        }
    }
#pragma unroll
    for (int fh = 0; fh < FH; ++fh) {
        const bool ok = tile_ok && row0 + 16 * fh + ln < rows;
        float s1 = 0.f;
#pragma unroll
        for (int b = 0; b < KB; ++b)
            s1 += ((cv[fh][b][0] + cv[fh][b][1]) + (cv[fh][b][2] + cv[fh][b][3])) + ((cv[fh][b][4] + cv[fh][b][5]) + (cv[fh][b][6] + cv[fh][b][7]));
        const float mean = rows_sum(s1) / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cv[fh][b][e] -= mean;
                s2 = fmaf(cv[fh][b][e], cv[fh][b][e], s2);
            }
        const float rstd = 1.0f / sqrtf(rows_sum(s2) / (float)C + 1e-8f);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const int c0 = 32 * b + 8 * lg;
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4_t lw = *reinterpret_cast<const f32x4_t*>(w.ln_w + c0 + 4 * h), lb = *reinterpret_cast<const f32x4_t*>(w.ln_b + c0 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * h + e] = ok ? cv[fh][b][4 * h + e] * rstd * lw[e] + lb[e] : 0.f;  // frames past the end: zeros
            }
            unsigned p[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(o[2 * j], o[2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[FH == 2 ? 2 * b + fh : b][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
        }
    }
}

// FH: frame halves (16 frames each) a wave owns.  2: the 32-frame tiles described above.  1 (round 4): HALF tiles, for grids that leave
// most of the chip idle — a single clip (the streaming chunk) is 29 tiles at C = 256, eight workgroups, and each wave's tile is 82 us
// of MFMAs on its one SIMD however fast the weights arrive.  With 16 frames per wave there are twice the waves, every weight piece
// meets one column group instead of two, and the activation has only its "frame half 0" stream (beside the second product of the
// previous hidden tile).  Per element the same products in the same order: the same bits as FH = 2 (tested).
// Non-temporal policy on the main kernel's STREAMED data — the operand planes, the residual rows, the
// output rows — so that they do not evict the weight image from L2 (PMC: the C = 256 unit fetched 1.96 x its algorithmic bytes, the
// excess being ~250 re-fetches of the 3.1 MB image per launch).  Measured (profiles/r04/wide_nt.md, two interleaved rounds on one
// box): all three on, the three C = 256 units 3.16-3.21 -> 3.06-3.07 ms, the next unit's front end 0.47 -> 0.44, the step - 1.7 %.
// (The half-tile form's weight copies with the non-temporal policy: measured, no gain — 86 vs 83 us at C = 256, 51 vs 52 at C = 192.)
// tail_tiles (FHK = 2 only): the LAST tail_tiles 32-frame tiles are left out of the lock-step passes and run afterwards as half tiles on
// the first workgroups — 256 x 900 frames at C = 256 are 7 full passes + 32 tiles, and an eighth pass for 0.4 % of the tiles cost 12 %
// of the unit; as 64 half tiles on 16 workgroups it costs half a pass.
template <int C, int FHK = 2>
__global__ __launch_bounds__(256, WGeo<C>::WG_PER_CU) void conv_unit_wide_kernel(const ConvUnitW w, const unsigned char* __restrict__ planes,
                                                              const float* __restrict__ x, float* __restrict__ y, const int64_t rows,
                                                              const int64_t tail_tiles, const int frames, int* __restrict__ counters) {
    using G = WGeo<C>;
    static_assert(FHK == 1 || FHK == 2, "one or two frame halves per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
    unsigned char* ring = smem_wide + G::OFF_RING;
    float* Pt = reinterpret_cast<float*>(smem_wide + G::OFF_P);
    float* B1s = reinterpret_cast<float*>(smem_wide + G::OFF_B1);
    float* B2s = reinterpret_cast<float*>(smem_wide + G::OFF_B2);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15;  // frame within a 16-frame half of the tile (MFMA column) / weight row of a fragment
    const int lg = lane >> 4;  // k group of an operand fragment; row group 4 lg .. 4 lg + 3 of an accumulator tile

    // LDS addressing of the fragment reads: two per-lane byte offsets kept opaque to the optimiser (ring positions 0-3, 4-7), the
    // rest an immediate of the ds_read (16-bit field).  Left alone, hipcc keeps every base + constant combination beyond 64 KB in
    // a register of its own: 23 registers at C = 256, paid for with AGPR copies of the operands inside the loop.
    auto opaque = [](int v) __attribute__((always_inline)) -> int {
        asm volatile("" : "+v"(v));
        return v;
    };
    const int ring_lo = opaque(G::OFF_RING + 16 * lane);
    const int ring_hi = opaque(G::OFF_RING + 16 * lane + G::HALF_POS * G::SLOT);
    static_assert((G::HALF_POS - 1) * G::SLOT + (G::KS - 1) * 3072 + 2 * 1024 + 1024 < 65536, "fragment offsets must fit the ds_read immediate");

    // ---- the weight stream: copying wave v moves bytes [1024 DMA_N v, 1024 DMA_N (v + 1)) of every slot ------------------
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned char* const src_wave = w.wide_img + (G::DMA_N * 1024) * wave;  // wave-uniform
    const unsigned lane_off = 16u * (unsigned)lane;
    int dma_slot = 0;  // next slot of the stream to fetch (wave-uniform)
    auto issue = [&](int ring_pos) __attribute__((always_inline)) {
        if (G::COPY_WAVES == 4 || wave < G::COPY_WAVES)  // (a wave without copies meets the same counted waits with nothing outstanding)
            dma_slot_quarter<G::DMA_N>(src_wave + (int64_t)dma_slot * G::SLOT, lane_off, ring_lds + (unsigned)(ring_pos * G::SLOT) + (unsigned)(G::DMA_N * 1024) * (unsigned)wave);
        dma_slot = dma_slot + 1 == G::TOTAL ? 0 : dma_slot + 1;
    };
    // end of a step: this wave's pieces of the slot after next have landed, then everybody's have, and everybody is done
    // reading the slot that the next step's DMA overwrites
    auto step_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int j = 0; j < G::PF; ++j) issue(j);

    // ---- parameters resident for the lifetime of the workgroup ---------------------------------------------------
    // activation parameters, interleaved per PAIR of hidden channels (2c, 2c+1): alpha x2, 1/alpha x2 | gamma x2, beta x2
    for (int i = tid; i < G::H4; i += 256) {
        float* row = Pt + 8 * (i >> 1) + (i & 1);
        row[0] = w.alpha[i];
        row[2] = w.inv_alpha[i];
        row[4] = w.gamma[i];
        row[6] = w.beta[i];
        B1s[i] = w.b1[i];
    }
    for (int i = tid; i < C; i += 256) B2s[i] = w.b2[i];
    __syncthreads();  // (plain loads above: hipcc drains them here, the DMA pieces with them)

    const int64_t tiles32 = (rows + 31) / 32;
    const int64_t n_tiles = FHK == 2 ? tiles32 - tail_tiles : 2 * tiles32;  // units of the lock-step passes: 32-frame tiles, or 16-frame halves
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // Passes run in lock step across the chip, so every workgroup would fetch its operand planes and store its tile at the
    // same moment: HBM idles through the loops and saturates in between (the store phase measured 24 k cycles = 10.7 B / clk /
    // CU).  Workgroups that run one pass fewer than the busiest ones (at 256 x 900 frames: all but 8) have a whole pass of
    // slack, so they start up to 0.9 pass late, in 8 steps by blockIdx / 8 (i.e. evenly inside every XCD): their memory
    // phases then fall into other workgroups' compute phases.  Results do not depend on it.
    {
        const int64_t wg_passes = (n_tiles + 3) / 4;
        const int64_t max_passes = (wg_passes + gridDim.x - 1) / gridDim.x;
        const int64_t my_passes = (wg_passes - blockIdx.x + gridDim.x - 1) / gridDim.x;
        constexpr long long PASS_CYCLES = (long long)G::NT * G::NSTEP * (12 * G::KS) * 16 * 6 / 5 + 35000;
        // (with a tail of half tiles the workgroups that do NOT run it have half a pass of slack)
        const bool tail_slack = FHK == 2 && tail_tiles > 0 && (int64_t)blockIdx.x * 4 >= 2 * tail_tiles;
        const long long delay = my_passes < max_passes ? (long long)((blockIdx.x >> 3) & 7) * (PASS_CYCLES * 9 / 80)
                                : tail_slack        ? (long long)((blockIdx.x >> 3) & 7) * (PASS_CYCLES * 9 / 160)
                                                    : 0;
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        if (!(G::DYN && FHK == 2 && counters))
            while ((long long)__builtin_amdgcn_s_memtime() - t0 < delay) __builtin_amdgcn_s_sleep(16);
    }
    // every wave of the block runs the same number of passes (block barriers inside)
    int pass_no = 0;
    (void)pass_no;
    // one unit of FH frame halves: `tile` counts units of 16 FH frames from row 0, units >= unit_limit are computed but not stored
    auto run_unit = [&](auto fh_, const int64_t tile, const int64_t unit_limit) __attribute__((always_inline)) {
        constexpr int FH = decltype(fh_)::value;
        constexpr int GPP = 6 * FH;  // MFMA gaps per weight piece
        WIDE_STAMP(0);
        const int64_t row0 = tile * (16 * FH);
        const bool tile_ok = tile < unit_limit;
        const int64_t tile32 = FH == 2 ? tile : tile >> 1;  // the 32-frame tile of the operand-plane image
        const int fsel = FH == 2 ? 0 : (int)(tile & 1);     // FH = 1: which frame half of it

        // ---- LayerNorm(dw_conv7(x)) of this tile, already split and in fragment order (dwconv_ln_split_kernel): k step s,
        //      plane p is one 1-KB block, 16 B per lane
        // (block s = 2 b + fh of the image: k block b, frame half fh; FH = 1 keeps the blocks of its own half: ap[b])
        bf16x8 ap[G::NS1 / (3 - FH)][3];
        if constexpr (G::FRONT) {
            // (the lane number made opaque per pass: what the front end derives from it — row pointers, parameter addresses — is then computed
            // here instead of being hoisted out of the pass loop, where at 256 registers for the loop every such value was spilled and came back
            // through scratch behind an s_waitcnt vmcnt(0), one memory round trip each)
            int lane_f = lane;
            asm volatile("" : "+v"(lane_f));
            wide_front<C, FH>(w, x, rows, frames, row0, tile_ok, lane_f & 15, lane_f >> 4, ap);
        } else {
            const unsigned char* src = planes + (tile_ok ? tile32 : 0) * (int64_t)(G::NS1 * 3072) + 16 * lane + (FH == 1 ? fsel * 3072 : 0);
#pragma unroll
            for (int s = 0; s < G::NS1 / (3 - FH); ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x8* q = reinterpret_cast<const bf16x8*>(src + ((FH == 1 ? 2 * s : s) * 3 + pl) * 1024);
                    ap[s][pl] = __builtin_nontemporal_load(q);
                }
        }

        // At C = 256 the 192 operand registers, the 160 accumulators and the activation's working set exceed the 256 arch VGPRs:
        // left alone hipcc parks part of ap in AGPRs and copies it back (4 v_accvgpr_read per MFMA) inside the loop.  An MFMA
        // reads its B operand from an AGPR just as well: the last k steps are moved there for good, once per pass.
        if constexpr (FH == 2) {
#pragma unroll
            for (int s = G::AP_AGPR_FROM; s < G::NS1; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+a"(ap[s][pl]));
        }

        WIDE_STAMP(1);
        // ---- output accumulators start at the pw_conv2 bias: yacc[rt][fh][i] = y[channel 16 rt + 4 lg + i][frame 16 fh + ln]
        f32x4_t yacc[G::RT][FH];
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) {
            const f32x4_t b2v = *reinterpret_cast<const f32x4_t*>(B2s + 16 * rt + 4 * lg);
#pragma unroll
            for (int fh = 0; fh < FH; ++fh) yacc[rt][fh] = b2v;
        }

        // the slot at ring position 0 (and the one after it) must have landed: first pass = the prologue's copies,
        // later passes = guaranteed by the previous pass's last step
        step_sync();

        // a hidden tile of 32 channels x 32 frames = four 16 x 16 accumulator tiles, index 2 hh + fh (hidden half, frame half):
        // element i of tile (hh, fh) = hidden channel 16 hh + 4 lg + i at frame 16 fh + ln
        struct XTile { f32x4_t q[4]; };
        auto bias1 = [&](int nt) __attribute__((always_inline)) -> XTile {
            XTile t;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(B1s + 32 * nt + 16 * hh + 4 * lg);
                t.q[2 * hh] = v;
                t.q[2 * hh + 1] = v;
            }
            return t;
        };
        // the activated hidden tile as the second product's B operand (snake + GRN with normaliser 1, layers.py:29-33, :112-115,
        // then the bf16x3 split): xbp[fh][plane][w], word w = 2 hh + ip = hidden channels 16 hh + 4 lg + 2 ip, + 1 of frame 16 fh + ln
        // — the k order sigma(lg, j) = (j < 4 ? 4 lg + j : 16 + 4 lg + j - 4) that the W2 image is built in
        unsigned xbp[2][3][4];
        unsigned xbq[3][4];  // frame half 0 of the NEXT tile while xbp[0] still feeds the running second product (a separate array: updating
                             // xbp[0] in place made hipcc keep both generations of every tuple and spill 240 registers)
        // One ring slot = one "slot step" of 12 KS MFMAs (v_mfma_f32_16x16x32_bf16, 16 cycles): KS pieces of 3 planes x 1 KB, a piece =
        // one weight fragment (16 rows x 32 k) used against both frame halves, 6 plane products each.  First product: piece
        // (k block b, hidden half hh); second product: piece = output row tile rt.  Every instruction of a slot step is placed
        // into one of its 12 KS MFMA gaps and a sched_barrier(0) closes every window of three gaps (inside it a sched_group_barrier
        // pattern of one MFMA + two fillers): gap g = (piece, frame half, plane
        // product m) holds the MFMA, at three gaps of a piece one fragment read of the NEXT piece (the last piece fetches the first fragment of the following slot, which landed a step ago),
        // in gap 0 the DMA issue, and the stages of the activation stream that fall to it.  The activation of a hidden tile is
        // 8 pairs x ACT_STAGES stages: frame half 0 runs beside the SECOND product of the previous tile, frame half 1 beside
        // the FIRST product of the next tile, so the vector work is spread over all 2 NA x 12 KS gaps of an iteration.
        bf16x8 fb[2][3];
        auto frag1 = [&](int ring_pos, int piece, int pl) __attribute__((always_inline)) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(smem_wide + (ring_pos < G::HALF_POS ? ring_lo : ring_hi) + (ring_pos % G::HALF_POS) * G::SLOT + piece * 3072 + 1024 * pl);
        };
        ActPair ast[4];
        // stages of the half-tile stream (4 pairs x ACT_STAGES) that fall to gap `gap` of the half phase's 12 KS NA gaps; pair P of
        // frame half HALF = hidden half P >> 1, register pair P & 1 of accumulator tile 2 (P >> 1) + HALF
        auto act_gap = [&](auto gap_, auto half_, const XTile& xa, const float* tab) __attribute__((always_inline)) {
            constexpr int GAP = decltype(gap_)::value, HALF = decltype(half_)::value;
            constexpr int TOTAL_ST = 4 * ACT_STAGES, GAPS = GPP * G::KS * G::NA;
            constexpr int LO = GAP * TOTAL_ST / GAPS, HI = (GAP + 1) * TOTAL_ST / GAPS;
            static_for<HI - LO>([&](auto k_) {
                constexpr int ST = LO + decltype(k_)::value, P = ST / ACT_STAGES, T = ST % ACT_STAGES;
                constexpr int Q = 2 * (P >> 1) + HALF, E = 2 * (P & 1);
                // table row of channels 32 nt + 16 hh + 4 lg + 2 ip, + 1: 8 floats per pair; tab carries nt and lg
                const float* row = tab + 64 * (P >> 1) + 8 * (P & 1);
                if constexpr (HALF == 0)
                    act_stage<T>(ast[P], row, xa.q[Q][E], xa.q[Q][E + 1], xbq[0][P], xbq[1][P], xbq[2][P]);
                else
                    act_stage<T>(ast[P], row, xa.q[Q][E], xa.q[Q][E + 1], xbp[1][0][P], xbp[1][1][P], xbp[1][2][P]);
            });
        };
        // PHASE 0: first product, pieces 4 I .. 4 I + 3 = (k block, hidden half) of a hidden tile from ring slot POS into acc.
        // PHASE 1: second product, output row tiles 4 I .. 4 I + 3 from ring slot POS with B operand xb[frame half].
        // NEXT: the ring slot whose first fragment is fetched for the following slot step (-1: none).  ACT: -1 none, else the
        // frame half (0 / 1) of the activation stream of xa that runs in the gaps.
        auto slot_step = [&](auto phase_, auto pos_, auto next_, auto i_, auto act_, int issue_pos, XTile& acc,
                             const bf16x8 (&xb)[2][3], const XTile& xa, const float* tab) __attribute__((always_inline)) {
            constexpr int PHASE = decltype(phase_)::value, POS = decltype(pos_)::value, NEXT = decltype(next_)::value;
            constexpr int I = decltype(i_)::value, ACT = decltype(act_)::value;
            static_for<GPP * G::KS>([&](auto g_) {
                constexpr int g = decltype(g_)::value, pc = g / GPP, fh = (g % GPP) / 6, m = g % 6;
                constexpr int piece = G::KS * I + pc;
                // (the fragment double buffer alternates with the PIECE, not with its position in the slot: slots of three pieces)
                if constexpr (PHASE == 0)
                    acc.q[2 * (piece & 1) + fh] = mfma_plane<m>(fb[piece & 1], ap[FH == 2 ? 2 * (piece >> 1) + fh : (piece >> 1)], acc.q[2 * (piece & 1) + fh]);
                else
                    yacc[piece][fh] = mfma_plane<m>(fb[piece & 1], xb[fh], yacc[piece][fh]);
                // the next piece's planes in the order 0, 1, 2: its first MFMA takes plane 2 of the weights, the YOUNGEST read, so the one
                // s_waitcnt in front of it covers all three (LDS returns in order) instead of one wait per plane
                constexpr int FSTEP = FH == 2 ? 3 : 2;  // gaps between the three fragment reads of the next piece
                if constexpr ((g % GPP) % FSTEP == 0 && (g % GPP) / FSTEP < 3) {
                    constexpr int pl = (g % GPP) / FSTEP;
                    if constexpr (pc + 1 < G::KS)
                        fb[(piece + 1) & 1][pl] = frag1(POS, pc + 1, pl);
                    else if constexpr (NEXT >= 0)
                        fb[(piece + 1) & 1][pl] = frag1(NEXT, 0, pl);  // (a product has an even number of pieces: the next one starts at fb[0] again)
                }
                if constexpr (g == 0) issue(issue_pos);
                if constexpr (ACT >= 0) act_gap(std::integral_constant<int, GPP * G::KS * I + g>{}, std::integral_constant<int, ACT>{}, xa, tab);
                // a wall (sched_barrier) closes every window of WALL gaps (after every gap, or every second, hipcc's register allocation
                // goes over the edge at C = 256); inside a window: one MFMA, then at most two fillers — a 16-cycle MFMA holds the issue
                // port for 8, so a gap costs max(16, 8 + 4 fillers) cycles: two per gap are free, the third and fourth cost 4 cycles each
                // (measured: the loop's cycles follow that sum over the gaps of the compiled stream)
                constexpr int WALL = 3;
                if constexpr (g % WALL == WALL - 1) {
                    static_for<WALL>([&](auto) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x106, 2, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            step_sync();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto make_xb = [&](bf16x8 (&xb)[2][3]) __attribute__((always_inline)) {
#pragma unroll
            for (int fh = 0; fh < FH; ++fh)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    xb[fh][pl] = __builtin_bit_cast(bf16x8, u32x4{xbp[fh][pl][0], xbp[fh][pl][1], xbp[fh][pl][2], xbp[fh][pl][3]});
        };
        using IC_none = std::integral_constant<int, -1>;
        const float* const tab_lane = Pt + 16 * lg;  // + 128 nt: the lane's rows of hidden tile nt
        bf16x8 xb[2][3] = {};
        XTile xdummy = {};

        WIDE_STAMP(2);
        // (round 6, two workgroups per CU: a wave inside its product loop outranks the other workgroup's wave while that one is in its front
        // end or epilogue — otherwise the OLDER wave's 1 300 front-end instructions are served before the younger wave's products:
        // 1.27 -> 1.24 ms for the three C = 96 units, profiles/r06/wide96_sched/prio_ab.txt; priority 3, or the younger workgroup only: no better)
        if constexpr (G::WG_PER_CU == 2) __builtin_amdgcn_s_setprio(1);
        // ---- first product of hidden tile 0 (ring slots 0 .. NA-1), then frame half 0 of its activation: nothing to overlap
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[0][pl] = frag1(0, 0, pl);
        XTile xacc = bias1(0);
        static_for<G::NA>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            slot_step(std::integral_constant<int, 0>{}, std::integral_constant<int, i>{}, std::integral_constant<int, i + 1>{}, i_, IC_none{},
                      (i + G::PF) % G::NSTEP, xacc, xb, xdummy, tab_lane);
        });
        static_for<4 * ACT_STAGES>([&](auto st_) {
            constexpr int ST = decltype(st_)::value, P = ST / ACT_STAGES, T = ST % ACT_STAGES, Q = 2 * (P >> 1), E = 2 * (P & 1);
            act_stage<T>(ast[P], tab_lane + 64 * (P >> 1) + 8 * (P & 1), xacc.q[Q][E], xacc.q[Q][E + 1], xbp[0][0][P], xbp[0][1][P], xbp[0][2][P]);
        });

        WIDE_STAMP(3);
#pragma unroll 1
        for (int nt = 0; nt + 1 < G::NT; ++nt) {
            // ---- A: first product of tile nt+1 (slots NA .. 2NA-1) beside frame half 1 of tile nt's activation ------------
            XTile xnext = bias1(nt + 1);
            const float* const tab_a = tab_lane + 128 * nt;
            static_for<G::NA>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                slot_step(std::integral_constant<int, 0>{}, std::integral_constant<int, G::NA + i>{},
                          std::integral_constant<int, (G::NA + i + 1) % G::NSTEP>{}, i_, std::integral_constant<int, FH == 2 ? 1 : -1>{},
                          (G::NA + i + G::PF) % G::NSTEP, xnext, xb, xacc, tab_a);
            });
            // ---- B: second product of tile nt (slots 0 .. NA-1) beside frame half 0 of tile nt+1's activation ---------------
            make_xb(xb);
            const float* const tab_b = tab_a + 128;
            static_for<G::NA>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                slot_step(std::integral_constant<int, 1>{}, std::integral_constant<int, i>{}, std::integral_constant<int, i + 1>{}, i_,
                          std::integral_constant<int, 0>{}, (i + G::PF) % G::NSTEP, xdummy, xb, xnext, tab_b);
            });
            xacc = xnext;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 4; ++j) xbp[0][pl][j] = xbq[pl][j];
        }
        WIDE_STAMP(4);
        // ---- the residual rows of this wave's tile, all 4 C bytes per frame at once: the operand planes are dead from here on, so
        //      their registers hold the 8 x C/32 row pieces while the last second product runs (fetched tile by tile inside the
        //      store loop they cost 8 serial memory latencies: 21 k of a pass's 263 k cycles)
        int lane_e = lane;
        if constexpr (G::WG_PER_CU == 2) asm volatile("" : "+v"(lane_e));  // (as for the front end: the epilogue's addresses are computed here)
        const int er = lane_e >> 3, es = lane_e & 7;  // row within a group of 8, 16-B slot
        const int ln_e = G::WG_PER_CU == 2 ? lane_e & 15 : ln, lg_e = G::WG_PER_CU == 2 ? lane_e >> 4 : lg;
        float4 xres[G::CT][2 * FH];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int i = 0; i < 2 * FH; ++i) {
                const int64_t rr = row0 + 8 * i + er;
                const bool ok = tile_ok && rr < rows;
                {
                    typedef float f4nt __attribute__((ext_vector_type(4)));
                    const f4nt* q = reinterpret_cast<const f4nt*>(x + (ok ? rr : 0) * C + 32 * ct + 4 * es);
                    const f4nt v = __builtin_nontemporal_load(q);
                    xres[ct][i] = make_float4(v.x, v.y, v.z, v.w);
                }
            }
        // ---- last hidden tile: frame half 1 of its activation alone, second product from slots NA .. 2NA-1 ------------------
        if constexpr (FH == 2) {
            static_for<4 * ACT_STAGES>([&](auto st_) {
                constexpr int ST = decltype(st_)::value, P = ST / ACT_STAGES, T = ST % ACT_STAGES, Q = 2 * (P >> 1) + 1, E = 2 * (P & 1);
                act_stage<T>(ast[P], tab_lane + 128 * (G::NT - 1) + 64 * (P >> 1) + 8 * (P & 1), xacc.q[Q][E], xacc.q[Q][E + 1], xbp[1][0][P],
                             xbp[1][1][P], xbp[1][2][P]);
            });
        }
        make_xb(xb);
        static_for<G::NA>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            // the stream wraps: the slot after the pass's last one is ring position 0, the first slot of the next pass
            slot_step(std::integral_constant<int, 1>{}, std::integral_constant<int, G::NA + i>{},
                      std::integral_constant<int, (G::NA + i + 1) % G::NSTEP>{}, i_, IC_none{}, (G::NA + i + G::PF) % G::NSTEP, xdummy, xb,
                      xdummy, tab_lane);
        });

        if constexpr (G::WG_PER_CU == 2) __builtin_amdgcn_s_setprio(0);
        WIDE_STAMP(5);
        // ---- residual + store (xtract/nn/layers.py:59-62).  The accumulators hold, per lane, 4 channels of ONE frame for each of
        // the 4 C/32 (tile, group) pairs: stored directly that is 32 B per row per instruction.  Instead every 32-channel tile
        // goes through a 4-KB LDS buffer of this wave — written as [32 frames][8 slots of 16 B], slot = quad ^ swz(frame), read
        // back row-major — so that the residual load and the store move whole 128-B lines, 8 rows per instruction.
        {
            unsigned char* tbuf = smem_wide + G::OFF_TB + 8192 * wave;
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                unsigned char* tb = tbuf + (ct & 1) * 4096;  // two tiles in flight
                // this lane's 4 x 16 B of the 32-channel tile: row tiles 2 ct, 2 ct + 1 (channels 16 rtl + 4 lg .. + 3 = slot 4 rtl + lg),
                // frames ln and 16 + ln
#pragma unroll
                for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
                    for (int fh = 0; fh < FH; ++fh) {
                        const int fr = 16 * fh + ln_e;
                        const f32x4_t v = yacc[2 * ct + rtl][fh];
                        *reinterpret_cast<float4*>(tb + 128 * fr + 16 * ((4 * rtl + lg_e) ^ ((fr ^ (fr >> 3)) & 7))) = make_float4(v.x, v.y, v.z, v.w);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 2 * FH; ++i) {
                    const int r = 8 * i + er;
                    const int64_t rr = row0 + r;
                    const float4 v = *reinterpret_cast<const float4*>(tb + 128 * r + 16 * (es ^ ((r ^ (r >> 3)) & 7)));
                    if (tile_ok && rr < rows) {
                        typedef float f4nt __attribute__((ext_vector_type(4)));
                        const f4nt o = {xres[ct][i].x + v.x, xres[ct][i].y + v.y, xres[ct][i].z + v.z, xres[ct][i].w + v.w};
                        __builtin_nontemporal_store(o, reinterpret_cast<f4nt*>(y + rr * C + 32 * ct + 4 * es));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    if constexpr (G::DYN && FHK == 2) {
        if (counters) {
            // DYN (round 6, two workgroups per CU): units handed out by a counter.  The two waves that share a SIMD are not served alike — the
            // OLDER one (the workgroup dispatched first: blockIdx < gridDim / 2) issues as if it were alone (stamps: hidden-tile loop 54 k
            // cycles, a pass 81 k) and the younger one gets what is left (loop 106 k, pass 133 k) until the older workgroup has finished,
            // so with equal shares the first half of the grid ended after 0.7 of the kernel and the second half ran the rest alone
            // (profiles/r06/wide96_sched/).  Unit u < G4 = a group of four 32-frame tiles; after those, groups of four half tiles of the
            // last tail_tiles tiles (a shorter last unit per workgroup).  A workgroup's first unit is its blockIdx; thread 0 fetches the
            // next one at the START of a unit (the answer is not needed before its end) and hands it over through LDS.  Which workgroup
            // computes a tile does not enter the tile's arithmetic: the same bits.  counters[0] = units handed out beyond the first gridDim,
            // counters[1] = workgroups that have left; the last one to leave zeroes both for the next launch.
            int* const next_s = reinterpret_cast<int*>(smem_wide + G::OFF_NEXT);
            const int64_t g4 = (n_tiles + 3) / 4, total = g4 + (2 * tail_tiles + 3) / 4;
            int64_t u = blockIdx.x;
            while (u < total) {
                int nxt = 0;
                if (tid == 0) nxt = __hip_atomic_fetch_add(counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (int)gridDim.x;
                if (u < g4)
                    run_unit(std::integral_constant<int, 2>{}, 4 * u + wave, n_tiles);
                else
                    run_unit(std::integral_constant<int, 1>{}, 2 * n_tiles + 4 * (u - g4) + wave, 2 * tiles32);
                if (tid == 0) *next_s = nxt;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                u = __builtin_amdgcn_readfirstlane(*next_s);  // (the next write is a whole unit and its barriers away)
                ++pass_no;
            }
            WIDE_STAMP(6);
            if (tid == 0) {
                const int left = __hip_atomic_fetch_add(counters + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (left == (int)gridDim.x - 1) {
                    __hip_atomic_store(counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(counters + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_tiles; base += tile_stride, ++pass_no)
        run_unit(std::integral_constant<int, FHK>{}, base + wave, n_tiles);  // (every wave of the block runs the same number of passes)
    if constexpr (FHK == 2) {
        // the tail: half tiles 2 n_tiles .. 2 tiles32 - 1 (the weight stream continues where the last pass left it)
        for (int64_t base = (int64_t)blockIdx.x * 4; base < 2 * tail_tiles; base += tile_stride, ++pass_no)
            run_unit(std::integral_constant<int, 1>{}, 2 * n_tiles + base + wave, 2 * tiles32);
    }
    WIDE_STAMP(6);
    // leave no LDS-DMA in flight behind the workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- front end: depth-wise conv k7 + LayerNorm (modules.py:33-35, layers.py:80) -> three bf16 planes in the main kernel's
// fragment order.  One thread = one frame x one channel quad, one wave per frame, one 32-frame tile per workgroup (8 frames
// per wave); taps outside the frame's clip are the conv's zero padding.  Plane image: [tile of 32 frames][block s = 2 b + fh]
// [plane][1 KB] for k block b (32 channels) and frame half fh; inside a block lane (n = frame & 15, kg) owns 16 B = channels
// 32 b + 8 kg .. + 7 of frame 16 fh + n — the B operand of v_mfma_f32_16x16x32_bf16: channel quad Q of frame f lands in block
// 2 (Q >> 3) + (f >> 4) at byte 16 (16 ((Q >> 1) & 3) + (f & 15)) + 8 (Q & 1).
template <int C>
__global__ __launch_bounds__(256) void dwconv_ln_split_kernel(const ConvUnitW w, const float* __restrict__ x, unsigned char* __restrict__ planes,
                                                              const int64_t rows, const int frames) {
    constexpr int QN = C / 4;  // channel quads of a frame: one WAVE per frame (lanes >= QN idle), so that the LayerNorm sums are
                               // a fixed-order butterfly inside the wave — the same bits on every run and for every batch
    constexpr int IMG = (C / 16) * 3072;  // plane image of one 32-frame tile
    static_assert(QN <= 64 && IMG <= 64 * 1024, "one wave per frame, one tile image in LDS");
    // One workgroup = one tile: the 8-B pieces are assembled in LDS and leave as whole 1-KB blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char img[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const bool lane_ok = lane < QN;
    const int q = lane_ok ? lane : 0;
    const float4 dwb = *reinterpret_cast<const float4*>(w.dw_b + 4 * q);
    const float4 lw = *reinterpret_cast<const float4*>(w.ln_w + 4 * q);
    const float4 lb = *reinterpret_cast<const float4*>(w.ln_b + 4 * q);
    float4 dww[7];
#pragma unroll
    for (int tap = 0; tap < 7; ++tap) dww[tap] = *reinterpret_cast<const float4*>(w.dw_w + tap * C + 4 * q);
    const int t_tile = (int)((tile * 32) % frames);  // one 64-bit modulo per thread, not one per frame (~100 instructions each)
    // A wave's 8 frames are consecutive, so their 7-row windows overlap: a rolling register window loads each input row once
    // (14 loads for 8 frames instead of 56).  A row belongs to the frame's clip iff its time index stays inside [0, frames).
    const int lj0 = 8 * wv;
    auto load_row = [&](int d) -> float4 {  // input row (first frame of the wave) + d, zero outside that row's clip or the tensor
        const int64_t r = tile * 32 + lj0 + d;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_ok && r >= 0 && r < rows) v = *reinterpret_cast<const float4*>(x + r * C + 4 * q);
        return v;
    };
    float4 win[7];  // win[(k + 4) % 7] = row k - 3 ... the slot of input row d is (d + 7) % 7
#pragma unroll
    for (int d = -3; d <= 2; ++d) win[(d + 7) % 7] = load_row(d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lj = lj0 + i;
        const int64_t row = tile * 32 + lj;
        const bool ok = row < rows && lane_ok;
        const int t = row < rows ? (int)((unsigned)(t_tile + lj) % (unsigned)frames) : 0;
        win[(i + 3) % 7] = load_row(i + 3);  // overwrites row i - 4
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            acc = dwb;
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                // rows of a neighbouring clip (t + tap - 3 outside [0, frames)) are the conv's zero padding
                const bool in_clip = t + tap - 3 >= 0 && t + tap - 3 < frames;
                const float4 xv = in_clip ? win[(i + tap - 3 + 7) % 7] : make_float4(0.f, 0.f, 0.f, 0.f);
                acc.x = fmaf(xv.x, dww[tap].x, acc.x);
                acc.y = fmaf(xv.y, dww[tap].y, acc.y);
                acc.z = fmaf(xv.z, dww[tap].z, acc.z);
                acc.w = fmaf(xv.w, dww[tap].w, acc.w);
            }
        }
        const float s1 = wave_sum((acc.x + acc.y) + (acc.z + acc.w));  // idle lanes contribute exact zeros
        const float mean = s1 / (float)C;
        const float dx = acc.x - mean, dy = acc.y - mean, dz = acc.z - mean, dw_ = acc.w - mean;
        const float s2 = wave_sum(ok ? fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, dw_ * dw_))) : 0.f);
        const float rstd = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
        unsigned p[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};  // frames past the end: zeros
        if (ok) {
            split2(dx * rstd * lw.x + lb.x, dy * rstd * lw.y + lb.y, p[0][0], p[1][0], p[2][0]);
            split2(dz * rstd * lw.z + lb.z, dw_ * rstd * lw.w + lb.w, p[0][1], p[1][1], p[2][1]);
        }
        if (lane_ok) {
            unsigned char* dst = img + (2 * (q >> 3) + (lj >> 4)) * 3072 + 16 * (16 * ((q >> 1) & 3) + (lj & 15)) + 8 * (q & 1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + 1024 * pl) = make_uint2(p[pl][0], p[pl][1]);
        }
    }
    __syncthreads();
    unsigned char* out = planes + tile * (int64_t)IMG;
    // (plain stores: the planes are the main kernel's next read; non-temporal ones cost the front end 3 %: profiles/r04/wide_nt.md)
    for (int o = 16 * threadIdx.x; o < IMG; o += 16 * 256) *reinterpret_cast<u32x4*>(out + o) = *reinterpret_cast<const u32x4*>(img + o);
}

// ---- the SLICED form (round 5): few frames, many CUs -------------------------------------------------------------------------------
// conv_unit_wide_kernel gives every wave 16 or 32 frames END TO END: 6 144 MFMAs per 16 frames at C = 256 — 47 us on the wave's one
// SIMD however fast the weights arrive — and a single clip (the streaming chunk: 900 frames at C = 256, 178 at C = 192) fills 15 / 3
// workgroups: 80 / 49 us with 240 CUs idle.  Here the unit's two products are two launches over (frame tiles x channel slices), the
// hidden tensor going through L2 once as the bf16x3 operand blocks the second product reads (6 B per element, 5.5 MB for one clip):
//   wide_sliced_hidden_kernel   workgroup = 4 frame tiles of 16 (one per wave) x GH hidden tiles of 32: W1(nt) — the SAME pieces of the
//                               SAME image (conv_unit_wide_image) — copied to LDS in one shot, b1 + products over the k blocks in order,
//                               act_stage, split: block (nt, tile) of the hidden image = the registers xb of the fused kernel.
//   wide_sliced_out_kernel      workgroup = 2 frame tiles x 2 output row tiles of 16 channels (one pair per wave): per hidden tile nt a
//                               12-KB ring slot = W2(nt)'s two pieces + the two tiles' hidden blocks, 11 slots in flight by LDS-DMA
//                               behind counted waits; b2 + products over nt in order, + residual.
// Every accumulator sees the same operands in the same order as in conv_unit_wide_kernel<C, 1>: the SAME BITS (tested at block
// level and by the batch-invariance tests: a clip alone takes this form, inside a large batch the fused one).
template <int C>
struct SGeo {
    using G = WGeo<C>;
    static constexpr int PIECES = C / 16;        // pieces of W1(nt) (k block b, hidden half hh) / of W2(nt) (output row tile rt)
    static constexpr int T1 = PIECES * 3072;     // bytes of W1(nt) = bytes of W2(nt)
    static constexpr int GH = C >= 256 ? 2 : C <= 96 ? 2 : 1;  // hidden tiles per workgroup of the first kernel (C = 96: its front end is
                                                                // computed by every workgroup of a frame group — six of them: − 2 us against four hidden tiles each)
    static constexpr int LDS1 = GH * T1;
    static constexpr int RING = 12, PF = RING - 1, SLOT = 4 * 3072;
    static constexpr int LDS2 = RING * SLOT;
    static_assert(G::NT % GH == 0 && G::RT % 2 == 0 && G::NT >= RING && LDS1 <= 160 * 1024 && LDS2 <= 160 * 1024, "bad geometry");
    // byte offsets inside conv_unit_wide_image: W1(0) | W1(1) W2(0) | ... | W1(NT-1) W2(NT-2) | W2(NT-1)
    __host__ __device__ static constexpr int64_t w1_off(int nt) { return nt == 0 ? 0 : (int64_t)T1 * (2 * nt - 1); }
    __host__ __device__ static constexpr int64_t w2_off(int nt) { return nt == G::NT - 1 ? (int64_t)T1 * (2 * G::NT - 1) : (int64_t)T1 * 2 * (nt + 1); }
};

// hid: [hidden tile nt][frame tile of 16][plane][1 KB]: lane (frame n = lane & 15, k group g = lane >> 4) owns 16 B = the hidden channels
// 32 nt + sigma(g, j) of its frame, the B operand of the second product (conv_unit_wide_image, W2)
// (WGeo::FRONT, C = 96: no plane image — the front end runs here, wide_front as in the fused kernel's pass prologue: x / rows / frames)
template <int C>
__global__ __launch_bounds__(256, 1) void wide_sliced_hidden_kernel(const ConvUnitW w, const unsigned char* __restrict__ planes,
                                                                    unsigned char* __restrict__ hid, const int64_t tiles16, const int64_t tiles_pad,
                                                                    const float* __restrict__ x, const int64_t rows, const int frames) {
    using G = WGeo<C>;
    using S = SGeo<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4;
    constexpr int HG = G::NT / S::GH;
    const int nt0 = (int)(blockIdx.x % (unsigned)HG) * S::GH;
    const int64_t tile = (int64_t)(blockIdx.x / (unsigned)HG) * 4 + wave;
    const bool tile_ok = tile < tiles16;
    // the weights of this workgroup's hidden tiles, all in flight at once: the 3-KB pieces of every tile dealt round-robin to the waves
    {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_wide;
#pragma unroll
        for (int g = 0; g < S::GH; ++g)
#pragma unroll
            for (int j = 0; j < (S::PIECES + 3) / 4; ++j) {
                const int piece = wave + 4 * j;  // (wave-uniform)
                if (S::PIECES % 4 == 0 || piece < S::PIECES)
                    dma_slot_quarter<3>(w.wide_img + S::w1_off(nt0 + g) + 3072 * piece, 16u * (unsigned)lane, lds0 + (unsigned)(g * S::T1 + 3072 * piece));
            }
    }
    // LayerNorm(dw_conv7(x)) of this wave's 16 frames, split (dwconv_ln_split_kernel): k block b, plane p = one 1-KB block
    bf16x8 ap[C / 32][3];
    if constexpr (G::FRONT) {
        wide_front<C, 1>(w, x, rows, frames, tile * 16, tile_ok, lane & 15, lg, ap);
    } else {
        const int64_t tl = tile_ok ? tile : 0;
        const unsigned char* src = planes + (tl >> 1) * (int64_t)(G::NS1 * 3072) + (int)(tl & 1) * 3072 + 16 * lane;
#pragma unroll
        for (int b = 0; b < C / 32; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[b][pl] = *reinterpret_cast<const bf16x8*>(src + (2 * b * 3 + pl) * 1024);
    }
    // activation parameters of the lane's 8 hidden channels per tile (pair P = hidden half P >> 1, register pair P & 1: channels
    // 32 nt + 16 (P >> 1) + 4 lg + 2 (P & 1), + 1) in act_stage's table order, and the pw_conv1 bias the accumulators start at
    alignas(16) float tab[S::GH][4][8];
    f32x4_t b1v[S::GH][2];
#pragma unroll
    for (int g = 0; g < S::GH; ++g) {
#pragma unroll
        for (int P = 0; P < 4; ++P) {
            const int ch = 32 * (nt0 + g) + 16 * (P >> 1) + 4 * lg + 2 * (P & 1);
            const float2 al = *reinterpret_cast<const float2*>(w.alpha + ch), ia = *reinterpret_cast<const float2*>(w.inv_alpha + ch);
            const float2 ga = *reinterpret_cast<const float2*>(w.gamma + ch), be = *reinterpret_cast<const float2*>(w.beta + ch);
            tab[g][P][0] = al.x, tab[g][P][1] = al.y, tab[g][P][2] = ia.x, tab[g][P][3] = ia.y;
            tab[g][P][4] = ga.x, tab[g][P][5] = ga.y, tab[g][P][6] = be.x, tab[g][P][7] = be.y;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) b1v[g][hh] = *reinterpret_cast<const f32x4_t*>(w.b1 + 32 * (nt0 + g) + 16 * hh + 4 * lg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int g = 0; g < S::GH; ++g) {
        f32x4_t q[2] = {b1v[g][0], b1v[g][1]};
        const unsigned char* wl = smem_wide + g * S::T1 + 16 * lane;
        static_for<S::PIECES>([&](auto pc_) {
            constexpr int piece = decltype(pc_)::value;  // (k block piece >> 1, hidden half piece & 1)
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(wl + piece * 3072 + pl * 1024);
            static_for<6>([&](auto m_) { q[piece & 1] = mfma_plane<decltype(m_)::value>(fb, ap[piece >> 1], q[piece & 1]); });
        });
        ActPair ast[4];
        unsigned out[3][4];
        static_for<4 * ACT_STAGES>([&](auto st_) {
            constexpr int ST = decltype(st_)::value, P = ST / ACT_STAGES, T = ST % ACT_STAGES, E = 2 * (P & 1);
            act_stage<T>(ast[P], tab[g][P], q[P >> 1][E], q[P >> 1][E + 1], out[0][P], out[1][P], out[2][P]);
        });
        if (tile_ok) {
            unsigned char* dst = hid + ((int64_t)(nt0 + g) * tiles_pad + tile) * 3072 + 16 * lane;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 1024) = u32x4{out[pl][0], out[pl][1], out[pl][2], out[pl][3]};
        }
    }
}

template <int C>
__global__ __launch_bounds__(256, 1) void wide_sliced_out_kernel(const ConvUnitW w, const unsigned char* __restrict__ hid, const float* __restrict__ x,
                                                                 float* __restrict__ y, const int64_t rows, const int64_t tiles16, const int64_t tiles_pad) {
    using G = WGeo<C>;
    using S = SGeo<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    constexpr int RP = G::RT / 2;
    const int rp = (int)(blockIdx.x % (unsigned)RP);
    const int64_t fp = blockIdx.x / (unsigned)RP;
    const int ft = wave & 1, rg = wave >> 1;  // this wave's frame tile and output row tile of the workgroup's 2 x 2
    const int64_t tile = 2 * fp + ft;
    const bool tile_ok = tile < tiles16;
    const int rt = 2 * rp + rg;
    const int64_t row = tile * 16 + ln;
    const bool row_ok = tile_ok && row < rows;
    // the residual: this lane's four channels of its frame (a plain load, older than every copy below: the counted waits cover it)
    const f32x4_t xres = *reinterpret_cast<const f32x4_t*>(x + (row_ok ? row : 0) * C + 16 * rt + 4 * lg);
    // the pw_conv2 bias the accumulator starts at, by scalar loads (a vector load here would sit in front of the counted waits)
    f32x4_t yacc;
    {
        const float* b2t = w.b2 + 16 * rt;  // wave-uniform
        unsigned sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = __builtin_bit_cast(unsigned, b2t[i]);
        // (bit masks, not ?: chains: hipcc turned those into nested branches)
        const unsigned m0 = 0u - (unsigned)(lg == 0), m1 = 0u - (unsigned)(lg == 1), m2 = 0u - (unsigned)(lg == 2), m3 = 0u - (unsigned)(lg == 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) yacc[i] = __builtin_bit_cast(float, (sv[i] & m0) | (sv[4 + i] & m1) | (sv[8 + i] & m2) | (sv[12 + i] & m3));
    }
    // this wave's 3 KB of every slot: waves 0, 1 = the two W2 pieces, waves 2, 3 = the two hidden blocks
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_wide;
    const unsigned char* const src0 = wave < 2 ? w.wide_img + (int64_t)(2 * rp + wave) * 3072 : hid + (2 * fp + (wave - 2)) * 3072;
    const int64_t hid_stride = tiles_pad * 3072;
    auto issue = [&](auto nt_) __attribute__((always_inline)) {
        constexpr int nt = decltype(nt_)::value;
        const unsigned char* src = src0 + (wave < 2 ? S::w2_off(nt) : (int64_t)nt * hid_stride);
        dma_slot_quarter<3>(src, 16u * (unsigned)lane, ring_lds + (unsigned)((nt % S::RING) * S::SLOT + 3072 * wave));
    };
    static_for<S::PF>([&](auto j_) { issue(j_); });
    // fragments one hidden tile ahead: fw = this wave's W2 piece (A operand), fx = its frame tile's hidden block (B operand)
    bf16x8 fw[2][3], fx[2][3];
    auto fetch = [&](auto nt_) __attribute__((always_inline)) {
        constexpr int nt = decltype(nt_)::value;
        const unsigned char* slot = smem_wide + (nt % S::RING) * S::SLOT + 16 * lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            fw[nt & 1][pl] = *reinterpret_cast<const bf16x8*>(slot + rg * 3072 + pl * 1024);
            fx[nt & 1][pl] = *reinterpret_cast<const bf16x8*>(slot + (2 + ft) * 3072 + pl * 1024);
        }
    };
    // slot nt has landed for every wave, and every wave's reads of the slot before it are complete: `issued` slots requested so far
    auto landed = [&](auto nt_, auto issued_) __attribute__((always_inline)) {
        constexpr int after = decltype(issued_)::value - 1 - decltype(nt_)::value;  // slots requested after slot nt
        static_assert(after >= 0 && 3 * after <= 63, "counted wait out of range");
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * after) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    landed(std::integral_constant<int, 0>{}, std::integral_constant<int, S::PF>{});
    fetch(std::integral_constant<int, 0>{});
    static_for<G::NT>([&](auto nt_) {
        constexpr int nt = decltype(nt_)::value;
        if constexpr (nt + S::PF < G::NT) issue(std::integral_constant<int, nt + S::PF>{});  // into the position of slot nt - 1
        if constexpr (nt + 1 < G::NT) {
            constexpr int issued = nt + S::PF + 1 < G::NT ? nt + S::PF + 1 : G::NT;
            landed(std::integral_constant<int, nt + 1>{}, std::integral_constant<int, issued>{});
            fetch(std::integral_constant<int, nt + 1>{});
        }
        static_for<6>([&](auto m_) { yacc = mfma_plane<decltype(m_)::value>(fw[nt & 1], fx[nt & 1], yacc); });
    });
    // residual + store (xtract/nn/layers.py:59-62): the lane's 16 B of its frame
    if (row_ok) *reinterpret_cast<f32x4_t*>(y + row * C + 16 * rt + 4 * lg) = xres + yacc;
}

// frame tiles of 16 up to which the sliced form exists (its hidden image is part of the scratch: 24 C bytes per frame) and up to which
// it is taken by default (measured against the half-tile form: tools/wide_bench.py, profiles/r05/wide_sliced_sweep.txt)
constexpr int64_t SLICED_MAX_TILES = 256, SLICED_AUTO_TILES = 256;
__host__ constexpr size_t wide_planes_bytes(int c, int64_t rows) { return (size_t)((rows + 31) / 32) * 32 * (size_t)c * 6; }
__host__ constexpr int64_t sliced_tiles_pad(int64_t rows) { return ((rows + 15) / 16 + 3) / 4 * 4; }
__host__ constexpr bool sliced_exists(int c, int64_t rows) { return (rows + 15) / 16 <= SLICED_MAX_TILES; }
__host__ constexpr size_t sliced_hidden_bytes(int c, int64_t rows) { return (size_t)(4 * c / 32) * (size_t)sliced_tiles_pad(rows) * 3072; }

template <int C>
int launch_wide(hipStream_t s, const ConvUnitW& w, const float* x, float* y, unsigned char* planes, int64_t rows, int frames, int sliced_mode, int* counters) {
    using G = WGeo<C>;
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_wide_kernel<C, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_wide_kernel<C, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_sliced_hidden_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SGeo<C>::LDS1));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_sliced_out_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SGeo<C>::LDS2));
        configured.done();
    }
    char name[64];
    const int64_t tiles16 = ceil_div64(rows, 16);
    const bool sliced = sliced_exists(C, rows) && (sliced_mode == 2 || (sliced_mode == 1 && tiles16 <= SLICED_AUTO_TILES));
    if constexpr (!G::FRONT) {
        std::snprintf(name, sizeof(name), "dwconv_ln_split_kernel<%d>", C);
        ProfScope prof(s, name, (double)rows * 30.0 * C, (double)rows * 10.0 * C);
        constexpr int IMG = (C / 16) * 3072;
        hipLaunchKernelGGL((dwconv_ln_split_kernel<C>), dim3((unsigned)ceil_div64(rows, 32)), dim3(256), IMG, s, w, x, planes, rows, frames);
        L3AC_LAUNCH_CHECK();
    }
    if (sliced) {
        using S = SGeo<C>;
        const int64_t tiles_pad = sliced_tiles_pad(rows);
        unsigned char* hid = planes + (wide_planes_bytes(C, rows) + 255) / 256 * 256;
        {
            std::snprintf(name, sizeof(name), "wide_sliced_hidden_kernel<%d>", C);
            ProfScope p1(s, name, (double)rows * (8.0 * C * C + (G::FRONT ? 30.0 * C * (G::NT / S::GH) : 0.0)), (double)rows * 30.0 * C);
            hipLaunchKernelGGL((wide_sliced_hidden_kernel<C>), dim3((unsigned)(tiles_pad / 4 * (G::NT / S::GH))), dim3(256), S::LDS1, s, w, planes, hid,
                               tiles16, tiles_pad, x, rows, frames);
            L3AC_LAUNCH_CHECK();
        }
        std::snprintf(name, sizeof(name), "wide_sliced_out_kernel<%d>", C);
        ProfScope p2(s, name, (double)rows * 8.0 * C * C, (double)rows * 32.0 * C);
        hipLaunchKernelGGL((wide_sliced_out_kernel<C>), dim3((unsigned)(ceil_div64(tiles16, 2) * (G::RT / 2))), dim3(256), S::LDS2, s, w, hid, x, y, rows,
                           tiles16, tiles_pad);
        L3AC_LAUNCH_CHECK();
        return L3AC_OK;
    }
    const int64_t tiles = ceil_div64(rows, 32);
    int64_t blocks = ceil_div64(tiles, 4);
    const int cus = l3ac_device_cu_count();
    const bool half = 2 * blocks <= (int64_t)cus * G::WG_PER_CU;  // half tiles while twice the workgroups still fit one pass
    if constexpr (!G::FRONT) {
        // SLICED TAIL (round 6): a remainder of at most SLICED_MAX_TILES frame tiles behind the full passes (256 x 900 frames at C = 256: 7 full
        // passes + 32 tiles) cost 0.7 of a pass as half tiles on 16 workgroups — a wave's chain of 3 072 MFMAs however few waves run
        // (stamps: 180 k of the kernel's 1 800 k cycles, ~90 us).  The sliced form (two launches over frame tiles x channel slices, the same
        // bits) does those frames in ~25 us: the passes run on rows [0, 32 full), the two sliced launches on the rest, from the same plane
        // image.  One box, three interleaved rounds: the step 13.22 -> 13.11 ms (profiles/r06/sliced_tail_ab.txt).
        const int64_t per_pass = 4LL * 256 * G::WG_PER_CU;
        const int64_t full = tiles / per_pass * per_pass, rest = tiles - full;
        if (!half && full > 0 && rest > 0 && 2 * rest <= SLICED_MAX_TILES) {
            using S = SGeo<C>;
            const int64_t rows_main = 32 * full, rows_t = rows - rows_main;
            {
                std::snprintf(name, sizeof(name), "conv_unit_wide_kernel<%d>", C);
                ProfScope prof(s, name, (double)rows_main * 16.0 * C * C, (double)rows_main * 14.0 * C);
                hipLaunchKernelGGL((conv_unit_wide_kernel<C, 2>), dim3(256 * G::WG_PER_CU), dim3(256), G::LDS, s, w, planes, x, y, rows_main, (int64_t)0, frames,
                                   (int*)nullptr);
                L3AC_LAUNCH_CHECK();
            }
            const int64_t tiles16_t = ceil_div64(rows_t, 16), tiles_pad_t = sliced_tiles_pad(rows_t);
            const unsigned char* planes_t = planes + full * (int64_t)(G::NS1 * 3072);
            unsigned char* hid = planes + (wide_planes_bytes(C, rows) + 255) / 256 * 256;  // (conv_unit_wide_scratch_bytes reserves it)
            const float* x_t = x + rows_main * C;
            float* y_t = y + rows_main * C;
            {
                std::snprintf(name, sizeof(name), "wide_sliced_hidden_kernel<%d>", C);
                ProfScope p1(s, name, (double)rows_t * 8.0 * C * C, (double)rows_t * 30.0 * C);
                hipLaunchKernelGGL((wide_sliced_hidden_kernel<C>), dim3((unsigned)(tiles_pad_t / 4 * (G::NT / S::GH))), dim3(256), S::LDS1, s, w, planes_t, hid,
                                   tiles16_t, tiles_pad_t, x_t, rows_t, frames);
                L3AC_LAUNCH_CHECK();
            }
            std::snprintf(name, sizeof(name), "wide_sliced_out_kernel<%d>", C);
            ProfScope p2(s, name, (double)rows_t * 8.0 * C * C, (double)rows_t * 32.0 * C);
            hipLaunchKernelGGL((wide_sliced_out_kernel<C>), dim3((unsigned)(ceil_div64(tiles16_t, 2) * (G::RT / 2))), dim3(256), S::LDS2, s, w, hid, x_t, y_t, rows_t,
                               tiles16_t, tiles_pad_t);
            L3AC_LAUNCH_CHECK();
            return L3AC_OK;
        }
    }
    std::snprintf(name, sizeof(name), "conv_unit_wide_kernel<%d>", C);
    ProfScope prof(s, name, (double)rows * (16.0 * C * C + (G::FRONT ? 30.0 * C : 0.0)), (double)rows * (G::FRONT ? 8.0 : 14.0) * C);
    if (half) {
        blocks = ceil_div64(2 * tiles, 4);
        hipLaunchKernelGGL((conv_unit_wide_kernel<C, 1>), dim3((unsigned)blocks), dim3(256), G::LDS, s, w, planes, x, y, rows, (int64_t)0, frames, (int*)nullptr);
    } else {
        // A large grid runs its tiles in lock-step passes of 4 x 256 and the last pass may be nearly empty (256 x 900 frames at C = 256:
        // 7 200 tiles = 7 full passes + 32 tiles).  A remainder of at most half a pass is left out of the passes and run as HALF tiles
        // by the first workgroups, inside the same launch (conv_unit_wide_kernel, tail_tiles).
        const int64_t per_pass = 4LL * 256 * G::WG_PER_CU;
        const int64_t full = tiles / per_pass * per_pass, rest = tiles - full;
        const int64_t tail = (full > 0 && 2 * rest <= per_pass) ? rest : 0;
        if (blocks > 256 * G::WG_PER_CU) blocks = 256 * G::WG_PER_CU;

        int64_t tail_k = tail;
        int* ctr = nullptr;
        if (G::DYN && counters && tiles > 4 * blocks) {
            // units by counter (conv_unit_wide_kernel, DYN): the last tile's worth per workgroup as groups of half tiles (measured against
            // 0, 2 and 4 per workgroup: profiles/r06/wide96_sched/dyn_ab.txt)
            ctr = counters;
            tail_k = blocks;
            if (tail_k > tiles) tail_k = tiles;
        }
        hipLaunchKernelGGL((conv_unit_wide_kernel<C, 2>), dim3((unsigned)blocks), dim3(256), G::LDS, s, w, planes, x, y, rows, tail_k, frames, ctr);
    }
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace


bool conv_unit_wide_supported(int c) { return c == 96 || c == 128 || c == 192 || c == 256; }
// scratch the kernels need: the split LayerNorm output of `rows` frames (whole 32-frame tiles), 6 bytes per element, and — for row
// counts at which the sliced form exists — its hidden image behind it
size_t conv_unit_wide_scratch_bytes(int c, int64_t rows) {
    const size_t planes = wide_planes_bytes(c, rows);
    // (large row counts, C >= 128: the remainder of the last pass may run in the sliced form — launch_wide, 'sliced tail' — up to
    // SLICED_MAX_TILES frame tiles of hidden image behind the planes)
    if (sliced_exists(c, rows)) return (planes + 255) / 256 * 256 + sliced_hidden_bytes(c, rows);
    return c >= 128 ? (planes + 255) / 256 * 256 + sliced_hidden_bytes(c, 16 * SLICED_MAX_TILES) : planes;
}

// x must not alias y; `planes` = at least conv_unit_wide_scratch_bytes(c, batch * frames) bytes of scratch (checked).
// sliced_mode (context option "wide_sliced"): 0 the fused kernel always, 1 the sliced form for few frames (default), 2 wherever it exists
// counters: two zeroed ints of the context (the batch form at two workgroups per CU hands its units out by them and leaves them zeroed;
// null = lock-step passes)
int launch_conv_unit_wide(hipStream_t s, const ConvUnitW& w, const float* x, float* y, unsigned char* planes, size_t planes_bytes, int batch,
                          int frames, int sliced_mode, int* counters) {
    L3AC_REQUIRE(x != y && w.wide_img && planes && batch > 0 && frames > 0, "conv_unit_wide: bad arguments");
    const int64_t rows = (int64_t)batch * frames;
    L3AC_REQUIRE(planes_bytes >= conv_unit_wide_scratch_bytes(w.c, rows), "conv_unit_wide: scratch of %zu bytes, %zu needed (C=%d, %lld rows)",
                 planes_bytes, conv_unit_wide_scratch_bytes(w.c, rows), w.c, (long long)rows);
    L3AC_REQUIRE(ceil_div64(rows, 4) < ((int64_t)1 << 31), "conv_unit_wide: too many rows");
    switch (w.c) {
        case 96: return launch_wide<96>(s, w, x, y, planes, rows, frames, sliced_mode, counters);
        case 128: return launch_wide<128>(s, w, x, y, planes, rows, frames, sliced_mode, counters);
        case 192: return launch_wide<192>(s, w, x, y, planes, rows, frames, sliced_mode, counters);
        case 256: return launch_wide<256>(s, w, x, y, planes, rows, frames, sliced_mode, counters);
        default:
            l3ac_set_error("conv_unit_wide: C=%d not supported", w.c);
            return L3AC_EINVAL;
    }
}

// The weight stream in consumption order: W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W1(NT-1) W2(NT-2) | W2(NT-1), where W1(nt) / W2(nt) are
// the fragment-ordered images of hidden tile nt for v_mfma_f32_16x16x32_bf16 (A operand: lane (m = lane & 15, kg = lane >> 4) holds 8
// consecutive bf16 of row m), C/16 pieces of 3 planes x 1 KB each:
//   W1(nt): piece 2 b + hh (k block b, hidden half hh), plane p, lane (m, kg): W1[32 nt + 16 hh + m][32 b + 8 kg + j], j = 0..7
//   W2(nt): piece rt (output row tile), plane p, lane (m, kg): W2[16 rt + m][32 nt + sigma(kg, j)], sigma(kg, j) = j < 4 ? 4 kg + j :
//           16 + 4 kg + j - 4 — the order in which the activated accumulator tiles become the second product's B operand
namespace {
void put_split(std::vector<unsigned char>& img, size_t off, float v) {  // the three bf16 planes of v, 1 KB apart
    uint16_t pl[3];
    split3_host(v, pl);
    for (int p = 0; p < 3; ++p) std::memcpy(img.data() + off + (size_t)p * 1024, &pl[p], 2);
}
std::vector<unsigned char> wide_w1_image(const float* w1, int c) {
    const int h4 = 4 * c, pieces = c / 16;
    std::vector<unsigned char> img((size_t)(h4 / 32) * pieces * 3072, 0);
    for (int nt = 0; nt < h4 / 32; ++nt)
        for (int b = 0; b < c / 32; ++b)
            for (int hh = 0; hh < 2; ++hh)
                for (int m = 0; m < 16; ++m)
                    for (int kg = 0; kg < 4; ++kg)
                        for (int j = 0; j < 8; ++j)
                            put_split(img, ((size_t)nt * pieces + 2 * b + hh) * 3072 + (size_t)(16 * kg + m) * 16 + 2 * j,
                                      w1[(size_t)(32 * nt + 16 * hh + m) * c + 32 * b + 8 * kg + j]);
    return img;
}
std::vector<unsigned char> wide_w2_image(const float* w2, int c) {
    const int h4 = 4 * c, pieces = c / 16;
    std::vector<unsigned char> img((size_t)(h4 / 32) * pieces * 3072, 0);
    for (int nt = 0; nt < h4 / 32; ++nt)
        for (int rt = 0; rt < pieces; ++rt)
            for (int m = 0; m < 16; ++m)
                for (int kg = 0; kg < 4; ++kg)
                    for (int j = 0; j < 8; ++j) {
                        const int sigma = j < 4 ? 4 * kg + j : 16 + 4 * kg + j - 4;
                        put_split(img, ((size_t)nt * pieces + rt) * 3072 + (size_t)(16 * kg + m) * 16 + 2 * j,
                                  w2[(size_t)(16 * rt + m) * h4 + 32 * nt + sigma]);
                    }
    return img;
}
}  // namespace
std::vector<unsigned char> conv_unit_wide_image(const float* w1, const float* w2, int c) {
    const std::vector<unsigned char> i1 = wide_w1_image(w1, c), i2 = wide_w2_image(w2, c);
    const size_t t1 = (size_t)(c / 16) * 3072, t2 = t1;
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
