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
    static constexpr int QG = 2;            // channel quads whose depth-wise taps are loaded together
    static constexpr int KS = 4;            // k steps of W1 per slot
    static constexpr int SLOT = KS * 3 * 1024;  // = 2 output tiles of W2 (2 x 2 steps x 3 planes x 1 KB)
    static constexpr int NA = NS1 / KS;     // slots of one W1 tile == slots of one W2 tile (= CT / 2)
    static constexpr int NSTEP = 2 * NA;    // slots per hidden-tile iteration == ring size
    static constexpr int PF = NSTEP - 1;    // slots in flight
    static constexpr int RING = NSTEP * SLOT;
    static constexpr int TOTAL = NT * NSTEP;  // slots of the whole stream
    static constexpr int WAIT = 3 * (PF - 2); // this wave's DMA instructions that may stay outstanding at a step's end
    // LDS (bytes): ring | alpha, 1/alpha, gamma, beta [4][H4] | b1 [H4] | b2 [C] | dw_w [7][C], dw_b, ln_w, ln_b
    static constexpr int OFF_P = RING;
    static constexpr int OFF_B1 = OFF_P + 4 * H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4 * 4;
    static constexpr int OFF_DW = OFF_B2 + C * 4;
    static constexpr int LDS = OFF_DW + 10 * C * 4;
    static_assert(C % 64 == 0 && NS1 % KS == 0 && CT % 2 == 0 && NA == CT / 2, "bad geometry");
    static_assert(PF >= 3 && WAIT <= 63, "ring too small / vmcnt field too narrow");
    static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
};

__device__ float g_zero_row[512];  // what out-of-clip depth-wise taps read (zero-initialised)

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
        const unsigned char* src = src_lane + (int64_t)dma_slot * G::SLOT;
        const unsigned dst = ring_lds + (unsigned)(ring_pos * G::SLOT) + 3072u * (unsigned)wave;
        dma16(src, dst);
        dma16(src + 1024, dst + 1024u);
        dma16(src + 2048, dst + 2048u);
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

    auto frag = [&](int ring_pos, int piece, bf16x8 (&f)[3]) __attribute__((always_inline)) {  // 3 planes of one k step
        const unsigned char* p = ring + ring_pos * G::SLOT + piece * 3072 + 16 * lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(p + 1024 * pl);
    };

    const int64_t n_tiles = (rows + 31) / 32;
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // every wave of the block runs the same number of passes (block barriers inside)
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_tiles; base += tile_stride) {
        const int64_t row = (base + wave) * 32 + lj;  // this lane's global row
        const bool row_ok = row < rows;
        const int t = row_ok ? (int)(row % frames) : 0;  // frame inside its clip

        // ---- depth-wise conv k7 + LayerNorm for this lane's frame, channels 8q + 4 lh + {0..3} (modules.py:33-35) ----
        bf16x8 ap[G::NS1][3];
        {
            float a[4 * G::KQ];
            // taps that fall outside the lane's clip (zero padding, modules.py:19-20) read a row of zeros instead: no masks or
            // branches in the loop below
            const float* xr = x + (row_ok ? row : 0) * C + 4 * lh;
            const float* tap_src[7];
#pragma unroll
            for (int tap = 0; tap < 7; ++tap)
                tap_src[tap] = (row_ok && t + tap - 3 >= 0 && t + tap - 3 < frames) ? xr + (int64_t)(tap - 3) * C : g_zero_row + 4 * lh;
            float s1 = 0.f;
#pragma unroll
            for (int q = 0; q < G::KQ; ++q) {
                // the loop is fully unrolled (a[] must stay in registers); without a fence hipcc hoists all 7 KQ row loads to
                // the top (7 KQ x 4 registers).  Groups of QG quads: 7 QG loads in flight, enough to cover the L2 latency.
                if (q % G::QG == 0) __builtin_amdgcn_sched_barrier(0);
                const int k0 = 8 * q + 4 * lh;
                float4 acc = *reinterpret_cast<const float4*>(DWs + 7 * C + k0);
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const float4 xv = *reinterpret_cast<const float4*>(tap_src[tap] + 8 * q);
                    const float4 wv = *reinterpret_cast<const float4*>(DWs + tap * C + k0);
                    acc.x = fmaf(xv.x, wv.x, acc.x);
                    acc.y = fmaf(xv.y, wv.y, acc.y);
                    acc.z = fmaf(xv.z, wv.z, acc.z);
                    acc.w = fmaf(xv.w, wv.w, acc.w);
                }
                a[4 * q] = acc.x; a[4 * q + 1] = acc.y; a[4 * q + 2] = acc.z; a[4 * q + 3] = acc.w;
                s1 += (acc.x + acc.y) + (acc.z + acc.w);
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

        // ---- output accumulators start at the pw_conv2 bias ------------------------------------------------------
        f32x16_t yacc[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ct][r] = B2s[32 * ct + rowmap(r, lh)];

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
            const float* pp = Ps + 32 * nt + rowmap(r, lh);
            const f32x2 al = *reinterpret_cast<const f32x2*>(pp);
            const f32x2 ia = *reinterpret_cast<const f32x2*>(pp + G::H4);
            const f32x2 ga = *reinterpret_cast<const f32x2*>(pp + 2 * G::H4);
            const f32x2 be = *reinterpret_cast<const f32x2*>(pp + 3 * G::H4);
            f32x2 hv;
            hv.x = xacc[r];
            hv.y = xacc[r + 1];
            const f32x2 sv = snake_act2(hv, al, ia);
            const f32x2 o = __builtin_elementwise_fma(ga, sv, be) + sv;
            split2(o.x, o.y, xbp[r >> 3][0][(r & 7) >> 1], xbp[r >> 3][1][(r & 7) >> 1], xbp[r >> 3][2][(r & 7) >> 1]);
        };
        // second product of hidden tile nt from the ring slots OFF .. OFF + NA - 1 (2 output tiles per slot);
        // wf holds the first fragment on entry and, unless LAST_OF_PASS, the next slot's first fragment on exit
        auto second_product = [&](auto off_, auto last_, bf16x8 (&wf)[3]) __attribute__((always_inline)) {
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
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 cur[3];
                        if (c2 == 0 && s == 0) {
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) cur[pl] = wf[pl];
                        } else {
                            frag(OFF + i, 2 * c2 + s, cur);
                        }
                        yacc[2 * i + c2] = mfma_split(cur, xb[s], yacc[2 * i + c2]);
                    }
                // the slot after this one landed a step ago: its first fragment is fetched across the barrier
                if (!(LAST_OF_PASS && i + 1 == G::NA)) frag(i + 1 < G::NA ? OFF + i + 1 : G::NA, 0, wf);
                step_sync();
            }
        };

        // ---- first product of hidden tile 0 (ring slots 0 .. NA-1): nothing to overlap with -------------------------
        bf16x8 wf[3];
        frag(0, 0, wf);
        f32x16_t xacc = bias1(0);
#pragma unroll
        for (int i = 0; i < G::NA; ++i) {
            issue((i + G::PF) % G::NSTEP);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                bf16x8 cur[3];
                if (ks == 0) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) cur[pl] = wf[pl];
                } else {
                    frag(i, ks, cur);
                }
                xacc = mfma_split(cur, ap[G::KS * i + ks], xacc);
            }
            frag(i + 1, 0, wf);  // i + 1 == NA: the first slot of iteration 0
            step_sync();
        }

#pragma unroll 1
        for (int nt = 0; nt + 1 < G::NT; ++nt) {
            // ---- first product of tile nt+1 (slots NA .. 2NA-1) beside the activation of tile nt ----------------------
            f32x16_t xnext = bias1(nt + 1);
#pragma unroll
            for (int i = 0; i < G::NA; ++i) {
                issue((G::NA + i + G::PF) % G::NSTEP);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    constexpr int STEPS = G::KS * G::NA;  // k steps of this phase; 8 activation pairs are dealt over them
                    const int s = G::KS * i + ks;
                    bf16x8 cur[3];
                    if (ks == 0) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) cur[pl] = wf[pl];
                    } else {
                        frag(G::NA + i, ks, cur);
                    }
                    xnext = mfma_split(cur, ap[s], xnext);
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr)
                        if (pr * STEPS / 8 == s) act_pair(xacc, nt, 2 * pr);
                }
                frag(i + 1 < G::NA ? G::NA + i + 1 : 0, 0, wf);
                step_sync();
            }
            // ---- second product of tile nt (slots 0 .. NA-1) --------------------------------------------------------
            second_product(std::integral_constant<int, 0>{}, std::false_type{}, wf);
            xacc = xnext;
        }
        // ---- last hidden tile: activation alone, second product from slots NA .. 2NA-1 ------------------------------
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) act_pair(xacc, G::NT - 1, 2 * pr);
        second_product(std::integral_constant<int, G::NA>{}, std::true_type{}, wf);

        // ---- residual + store: lane (frame lj, half lh) owns channels 32 ct + 8 g + 4 lh + {0..3} -----------------
        if (row_ok) {
            const float* src = x + row * C;
            float* dst = y + row * C;
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * ct + 8 * g + 4 * lh;
                    const float4 xr = *reinterpret_cast<const float4*>(src + c0);
                    *reinterpret_cast<float4*>(dst + c0) = make_float4(xr.x + yacc[ct][4 * g], xr.y + yacc[ct][4 * g + 1],
                                                                       xr.z + yacc[ct][4 * g + 2], xr.w + yacc[ct][4 * g + 3]);
                }
        }
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
    const int64_t tiles = ceil_div64(rows, 32);
    int64_t blocks = ceil_div64(tiles, 4);
    if (blocks > 256) blocks = 256;
    ProfScope prof(s, name, (double)rows * (14.0 * C + 16.0 * C * C), (double)rows * 8.0 * C);
    hipLaunchKernelGGL((conv_unit_wide_kernel<C>), dim3((unsigned)blocks), dim3(256), G::LDS, s, w, x, y, rows, frames);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

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
