// Fused ConvUnit for the narrow stages (C = 24 / 48 / 96), reference l3ac/modules.py:10-41 + Residual
// (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// Unfused, the 4C-wide hidden tensor makes two HBM round trips and the channel contractions have K = 24..96,
// far too short for a tiled GEMM.  Here one WAVE owns 32 consecutive frames of one clip end to end:
//
//   * the products are evaluated TRANSPOSED:  X[n][m] = W1[n][:] . a[m][:]   (n = hidden channel, m = frame)
//     so the 32x32 accumulator tile has the hidden channel on its ROWS (registers) and the frame on its LANES.
//     snake / GRN are applied to the accumulator registers, and the tile is then used directly as the B operand
//     of the second product  Y[c][m] = W2[c][:] . X[:][m]  — the reduction runs over X's row index, which is
//     exactly what v_mfma_f32_32x32x2_f32 takes from a register without any lane movement (register r of lane
//     half h is row (r&3) + 8(r>>2) + 4h; the W2 fragment is fetched in the same order).  The hidden activations
//     never leave registers: no LDS transpose, no HBM traffic.
//   * C = 24 / 48: W1, W2 stay resident in LDS for the lifetime of a persistent workgroup and the waves never
//     synchronise.  C = 96: the weights (295 KB) are streamed through LDS in chunks of HC = 32 hidden channels, L2-served,
//     double-buffered: chunk c+1 is fetched into registers while chunk c is multiplied, one block barrier per chunk.
//   * lane (j, h) computes the depth-wise conv + LayerNorm of frame j for the channels k = 8q + 4h + {0..3} it
//     later feeds to the MFMAs as the B operand, from a per-wave LDS copy of the 38 input rows (halo 3 + 3).
//   * the first product of hidden tile nt+1 is issued before the activation of tile nt, so the MFMA pipe and the
//     VALU overlap inside one wave; fp32 throughout; bias vectors enter as the initial accumulator value.
//
// HBM traffic per frame: C floats in (+ 6/32 halo re-read, L2-served) and C floats out; the bound of this kernel
// is the fp32 MFMA rate (16 C^2 FLOP per frame, + the channel padding of the second product to 32 ceil(C/32)).
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "lane_sums.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C, int HC, int XH>  // XH: the input rows are staged in XH channel slices (LDS budget at C = 96)
struct Geo {
    static constexpr int H4 = 4 * C;          // hidden width
    static constexpr int NCH = H4 / HC;       // weight chunks (1 = resident)
    static constexpr int NTC = HC / 32;       // hidden tiles per chunk
    static constexpr int CT = (C + 31) / 32;  // output-channel tiles
    static constexpr int KQ = C / 8;          // k groups of the first product
    static constexpr int XC = C / XH;         // channels per staged slice
    static constexpr int XS = XC + 4;         // padded row strides (odd number of 16-B slots: conflict-free b128)
    static constexpr int W1S = C + 4;
    static constexpr int W2S = HC + 4;
    static constexpr int ROWS = 38;           // 32 frames + 3 + 3 halo
    // LDS carve (floats)
    static constexpr int NBUF = NCH == 1 ? 1 : 2;             // streamed weight chunks are double-buffered
    static constexpr int OFF_W1 = 0;                          // [HC][W1S]
    static constexpr int OFF_W2 = OFF_W1 + HC * W1S;          // [C + 1][W2S], last row all zeros (padding channels)
    static constexpr int WBUF = HC * W1S + (C + 1) * W2S;     // floats per weight buffer
    static constexpr int OFF_P = OFF_W1 + NBUF * WBUF;        // alpha[H4], 1/alpha[H4], gamma[H4], beta[H4]
    static constexpr int OFF_B1 = OFF_P + H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4;
    static constexpr int OFF_DW = OFF_B2 + 32 * CT;           // dw_w [7][C], dw_b, ln_w, ln_b
    static constexpr int OFF_XS = OFF_DW + 10 * C;
    static constexpr int xs_floats = ROWS * XS;
    static constexpr int lds_floats(int waves) { return OFF_XS + waves * xs_floats; }
    static_assert(H4 % HC == 0 && HC % 32 == 0 && C % (8 * XH) == 0 && (NCH == 1 || NCH % 2 == 0), "bad geometry");
};

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C, int WAVES, int HC, int XH>
__global__ __launch_bounds__(64 * WAVES) void conv_unit_fused_kernel(const ConvUnitW w, const float* __restrict__ x,
                                                                    float* __restrict__ y, int batch, int frames) {
    using G = Geo<C, HC, XH>;
    constexpr bool RESIDENT = G::NCH == 1;
    constexpr int THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1s = smem + G::OFF_W1;
    float* W2s = smem + G::OFF_W2;
    float* Ps = smem + G::OFF_P;
    float* B1s = smem + G::OFF_B1;
    float* B2s = smem + G::OFF_B2;
    float* DWs = smem + G::OFF_DW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xs = smem + G::OFF_XS + wave * G::xs_floats;

    auto stage_weights = [&](int chunk) {  // resident variant: hidden channels [chunk * HC, (chunk + 1) * HC), once
        for (int i = tid; i < HC * (C / 4); i += THREADS) {
            const int row = i / (C / 4), ch = i % (C / 4);
            *reinterpret_cast<float4*>(W1s + row * G::W1S + 4 * ch) =
                *reinterpret_cast<const float4*>(w.w1 + (int64_t)(chunk * HC + row) * C + 4 * ch);
        }
        for (int i = tid; i < (C + 1) * (HC / 4); i += THREADS) {
            const int row = i / (HC / 4), ch = i % (HC / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < C) v = *reinterpret_cast<const float4*>(w.w2 + (int64_t)row * G::H4 + chunk * HC + 4 * ch);
            *reinterpret_cast<float4*>(W2s + row * G::W2S + 4 * ch) = v;
        }
    };
    // streamed variant: chunk c+1 travels global -> registers while chunk c is being multiplied, and is written to the
    // OTHER LDS buffer after the products (one block barrier per chunk, no exposed L2 latency)
    constexpr int N1 = (HC * (C / 4) + THREADS - 1) / THREADS;
    constexpr int N2 = ((C + 1) * (HC / 4) + THREADS - 1) / THREADS;
    float4 wpre[RESIDENT ? 1 : N1 + N2];
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (!RESIDENT) {
#pragma unroll
            for (int k = 0; k < N1; ++k) {
                const int i = tid + THREADS * k;
                const int row = i / (C / 4), ch = i % (C / 4);
                wpre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < HC * (C / 4)) wpre[k] = *reinterpret_cast<const float4*>(w.w1 + (int64_t)(chunk * HC + row) * C + 4 * ch);
            }
#pragma unroll
            for (int k = 0; k < N2; ++k) {
                const int i = tid + THREADS * k;
                const int row = i / (HC / 4), ch = i % (HC / 4);
                wpre[N1 + k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < C) wpre[N1 + k] = *reinterpret_cast<const float4*>(w.w2 + (int64_t)row * G::H4 + chunk * HC + 4 * ch);
            }
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        if constexpr (!RESIDENT) {
            float* d1 = smem + G::OFF_W1 + buf * G::WBUF;
            float* d2 = smem + G::OFF_W2 + buf * G::WBUF;
#pragma unroll
            for (int k = 0; k < N1; ++k) {
                const int i = tid + THREADS * k;
                if (i < HC * (C / 4)) *reinterpret_cast<float4*>(d1 + (i / (C / 4)) * G::W1S + 4 * (i % (C / 4))) = wpre[k];
            }
#pragma unroll
            for (int k = 0; k < N2; ++k) {
                const int i = tid + THREADS * k;
                if (i < (C + 1) * (HC / 4)) *reinterpret_cast<float4*>(d2 + (i / (HC / 4)) * G::W2S + 4 * (i % (HC / 4))) = wpre[N1 + k];
            }
        }
    };

    // ---- parameters resident for the lifetime of the workgroup -------------------------------------------
    for (int i = tid; i < G::H4; i += THREADS) {
        Ps[i] = w.alpha[i];
        Ps[G::H4 + i] = w.inv_alpha[i];
        Ps[2 * G::H4 + i] = w.gamma[i];
        Ps[3 * G::H4 + i] = w.beta[i];
        B1s[i] = w.b1[i];
    }
    for (int i = tid; i < 32 * G::CT; i += THREADS) B2s[i] = i < C ? w.b2[i] : 0.f;
    for (int i = tid; i < 7 * C; i += THREADS) DWs[i] = w.dw_w[i];
    for (int i = tid; i < C; i += THREADS) {
        DWs[7 * C + i] = w.dw_b[i];
        DWs[8 * C + i] = w.ln_w[i];
        DWs[9 * C + i] = w.ln_b[i];
    }
    if (RESIDENT) {
        stage_weights(0);
    } else {
        load_chunk(0);
        store_chunk(0);  // made visible by the barrier of chunk step 0
    }
    __syncthreads();

    const int lj = lane & 31;  // frame within the tile (MFMA column) / weight row within a tile (A operand)
    const int lh = lane >> 5;
    const int tiles_per_clip = (frames + 31) / 32;
    const int64_t n_tiles = (int64_t)batch * tiles_per_clip;

    // input rows [t0 - 3, t0 + 35) of a tile's clip travel global -> registers -> the wave's LDS copy; with a single
    // channel slice (XH == 1) the NEXT tile's rows are fetched into registers while the current tile computes
    constexpr int NPRE = (G::ROWS * (G::XC / 4) + 63) / 64;
    int pre_row[NPRE], pre_src[NPRE], pre_dst[NPRE];  // tile-invariant: which row / chunk each of this lane's copies moves
#pragma unroll
    for (int it = 0; it < NPRE; ++it) {
        const int i = lane + 64 * it;
        const int row = i / (G::XC / 4), ch = i % (G::XC / 4);
        pre_row[it] = i < G::ROWS * (G::XC / 4) ? row - 3 : -(1 << 28);  // past the end: never a valid frame
        pre_src[it] = (row - 3) * C + 4 * ch;
        pre_dst[it] = i < G::ROWS * (G::XC / 4) ? row * G::XS + 4 * ch : -1;
    }
    auto load_rows = [&](int64_t tl, int xh, float4 (&pre)[NPRE]) {
        const bool ok = tl < n_tiles;
        const int bb = ok ? (int)(tl / tiles_per_clip) : 0;
        const int tt0 = ok ? (int)(tl % tiles_per_clip) * 32 : -(1 << 28);
        const float* cl = x + ((int64_t)bb * frames + (ok ? tt0 : 0)) * C + xh * G::XC;
#pragma unroll
        for (int it = 0; it < NPRE; ++it) {
            const int t = tt0 + pre_row[it];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < frames) v = *reinterpret_cast<const float4*>(cl + pre_src[it]);
            pre[it] = v;
        }
    };
    auto store_rows = [&](const float4 (&pre)[NPRE]) {
#pragma unroll
        for (int it = 0; it < NPRE; ++it)
            if (pre_dst[it] >= 0) *reinterpret_cast<float4*>(xs + pre_dst[it]) = pre[it];
    };
    const int64_t tile_stride = (int64_t)gridDim.x * WAVES;
    float4 pre[XH == 1 ? NPRE : 1];  // live across a tile only in the single-slice variants
    if constexpr (XH == 1) load_rows((int64_t)blockIdx.x * WAVES + wave, 0, pre);

    // every wave of the block runs the same number of iterations (the chunked variant has block barriers inside)
    for (int64_t base = (int64_t)blockIdx.x * WAVES; base < n_tiles; base += tile_stride) {
        const int64_t tile = base + wave;
        const bool tile_ok = tile < n_tiles;
        const int b = tile_ok ? (int)(tile / tiles_per_clip) : 0;
        const int t0 = tile_ok ? (int)(tile % tiles_per_clip) * 32 : 0;
        const float* clip = x + (int64_t)b * frames * C;

        // ---- depth-wise conv k7 for frame lj, channels k = 8q + 4 lh + r, from the wave's LDS copy of rows
        //      [t0 - 3, t0 + 35) of this clip (zeros outside the clip), staged one channel slice at a time ----
        float a[4 * G::KQ];
        float s1 = 0.f;
#pragma unroll
        for (int xh = 0; xh < XH; ++xh) {
            if constexpr (XH == 1) {
                store_rows(pre);
            } else {
                // register budget at C = 96: copy one 16-B chunk at a time
#pragma unroll 1
                for (int i = lane; i < G::ROWS * (G::XC / 4); i += 64) {
                    const int row = i / (G::XC / 4), ch = i % (G::XC / 4);
                    const int t = t0 - 3 + row;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (tile_ok && t >= 0 && t < frames)
                        v = *reinterpret_cast<const float4*>(clip + (int64_t)t * C + xh * G::XC + 4 * ch);
                    *reinterpret_cast<float4*>(xs + row * G::XS + 4 * ch) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (XH == 1) load_rows(tile + tile_stride, 0, pre);  // in flight until the next iteration
#pragma unroll
            for (int ql = 0; ql < G::KQ / XH; ++ql) {
                const int q = xh * (G::KQ / XH) + ql;
                const int k0 = 8 * q + 4 * lh;       // channel
                const int kl = 8 * ql + 4 * lh;      // channel within the staged slice
                float4 acc = *reinterpret_cast<const float4*>(DWs + 7 * C + k0);
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const float4 xv = *reinterpret_cast<const float4*>(xs + (lj + tap) * G::XS + kl);
                    const float4 wv = *reinterpret_cast<const float4*>(DWs + tap * C + k0);
                    acc.x = fmaf(xv.x, wv.x, acc.x);
                    acc.y = fmaf(xv.y, wv.y, acc.y);
                    acc.z = fmaf(xv.z, wv.z, acc.z);
                    acc.w = fmaf(xv.w, wv.w, acc.w);
                }
                a[4 * q] = acc.x; a[4 * q + 1] = acc.y; a[4 * q + 2] = acc.z; a[4 * q + 3] = acc.w;
                s1 += (acc.x + acc.y) + (acc.z + acc.w);
            }
            if (xh + 1 < XH) {  // the next slice overwrites the copy
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- LayerNorm over the C channels of frame lj (two lanes hold one frame) -----------------------
        s1 = halves_sum(s1);
        const float mean = s1 / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * G::KQ; ++i) {
            const float d = a[i] - mean;
            s2 = fmaf(d, d, s2);
        }
        s2 = halves_sum(s2);
        const float rstd = 1.0f / sqrtf(s2 / (float)C + 1e-8f);
        const bool frame_ok = tile_ok && t0 + lj < frames;
#pragma unroll
        for (int q = 0; q < G::KQ; ++q) {
            const int k0 = 8 * q + 4 * lh;
            const float4 g = *reinterpret_cast<const float4*>(DWs + 8 * C + k0);
            const float4 be = *reinterpret_cast<const float4*>(DWs + 9 * C + k0);
            a[4 * q] = frame_ok ? (a[4 * q] - mean) * rstd * g.x + be.x : 0.f;
            a[4 * q + 1] = frame_ok ? (a[4 * q + 1] - mean) * rstd * g.y + be.y : 0.f;
            a[4 * q + 2] = frame_ok ? (a[4 * q + 2] - mean) * rstd * g.z + be.z : 0.f;
            a[4 * q + 3] = frame_ok ? (a[4 * q + 3] - mean) * rstd * g.w + be.w : 0.f;
        }

        // ---- output accumulators start at the pw_conv2 bias ---------------------------------------------
        f32x16 yacc[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ct][r] = B2s[32 * ct + rowmap(r, lh)];

#pragma unroll 1
        for (int chunk = 0; chunk < G::NCH; ++chunk) {
            const float* W1s = smem + G::OFF_W1 + (RESIDENT ? 0 : (chunk & 1) * G::WBUF);  // NCH is even: parity is periodic
            const float* W2s = smem + G::OFF_W2 + (RESIDENT ? 0 : (chunk & 1) * G::WBUF);
            if (!RESIDENT) {
                // this chunk's buffer is complete (written during the previous step) and every wave is done reading the
                // other one, which the next chunk's rows (in flight from here on) will overwrite after the products
                __syncthreads();
                load_chunk(chunk + 1 < G::NCH ? chunk + 1 : 0);
            }
            const int n_base = chunk * HC;
            // X[n][m] = b1[n] + sum_k W1[n][k] a[m][k] for hidden tile ntl of this chunk
            auto first_product = [&](int ntl) -> f32x16 {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = B1s[n_base + 32 * ntl + rowmap(r, lh)];
#pragma unroll
                for (int q = 0; q < G::KQ; ++q) {
                    const float4 wf = *reinterpret_cast<const float4*>(W1s + (32 * ntl + lj) * G::W1S + 8 * q + 4 * lh);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, a[4 * q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, a[4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.z, a[4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.w, a[4 * q + 3], acc, 0, 0, 0);
                }
                return acc;
            };
            // software pipeline: tile ntl+1's first product (MFMA pipe) runs beside tile ntl's activation (VALU)
            f32x16 xnext = first_product(0);
#pragma unroll 1
            for (int ntl = 0; ntl < G::NTC; ++ntl) {
                f32x16 xacc = xnext;
                if (ntl + 1 < G::NTC) xnext = first_product(ntl + 1);
                // snake + GRN (normaliser == 1) on the accumulator registers (layers.py:29-33, :112-115)
                // registers (r, r+1), r even, are adjacent hidden channels: packed fp32 math on register pairs
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float* pp = Ps + n_base + 32 * ntl + rowmap(r, lh);
                    const f32x2 al = *reinterpret_cast<const f32x2*>(pp);
                    const f32x2 ia = *reinterpret_cast<const f32x2*>(pp + G::H4);
                    const f32x2 ga = *reinterpret_cast<const f32x2*>(pp + 2 * G::H4);
                    const f32x2 be = *reinterpret_cast<const f32x2*>(pp + 3 * G::H4);
                    f32x2 hv;
                    hv.x = xacc[r];
                    hv.y = xacc[r + 1];
                    const f32x2 s = snake_act2(hv, al, ia);
                    const f32x2 o = __builtin_elementwise_fma(ga, s, be) + s;
                    xacc[r] = o.x;
                    xacc[r + 1] = o.y;
                }
                // Y[c][m] += sum_n W2[c][n] X[n][m]: X's registers are the B operand, row order (r&3)+8(r>>2)+4h
#pragma unroll
                for (int ct = 0; ct < G::CT; ++ct) {
                    const int crow = (32 * ct + lj) < C ? 32 * ct + lj : C;  // padding channels read the zero row
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 wf = *reinterpret_cast<const float4*>(W2s + crow * G::W2S + 32 * ntl + 8 * g + 4 * lh);
                        yacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, xacc[4 * g], yacc[ct], 0, 0, 0);
                        yacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, xacc[4 * g + 1], yacc[ct], 0, 0, 0);
                        yacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.z, xacc[4 * g + 2], yacc[ct], 0, 0, 0);
                        yacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.w, xacc[4 * g + 3], yacc[ct], 0, 0, 0);
                    }
                }
            }
            if (!RESIDENT) store_chunk((chunk + 1) & 1);
        }

        // ---- residual + store: lane (frame lj, half lh) owns channels 32 ct + 8 g + 4 lh + {0..3} -------
        if (frame_ok) {
            float* dst = y + ((int64_t)b * frames + t0 + lj) * C;
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * ct + 8 * g + 4 * lh;
                    if (c0 < C) {
                        // residual: from the LDS copy when it holds all channels, else re-read (L2-served)
                        const float4 xr = XH == 1 ? *reinterpret_cast<const float4*>(xs + (lj + 3) * G::XS + c0)
                                                  : *reinterpret_cast<const float4*>(clip + (int64_t)(t0 + lj) * C + c0);
                        *reinterpret_cast<float4*>(dst + c0) =
                            make_float4(xr.x + yacc[ct][4 * g], xr.y + yacc[ct][4 * g + 1], xr.z + yacc[ct][4 * g + 2],
                                        xr.w + yacc[ct][4 * g + 3]);
                    }
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int C, int WAVES, int HC, int XH>
int launch_fused(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames, const char* name) {
    using G = Geo<C, HC, XH>;
    const size_t lds = (size_t)G::lds_floats(WAVES) * sizeof(float);
    static_assert(G::lds_floats(WAVES) * sizeof(float) <= 160 * 1024, "LDS budget exceeded");
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_fused_kernel<C, WAVES, HC, XH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured.done();
    }
    const int64_t tiles = (int64_t)batch * ((frames + 31) / 32);
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    int64_t blocks = ceil_div64(tiles, WAVES);
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    const double rows = (double)batch * frames;
    ProfScope prof(s, name, rows * (14.0 * C + 16.0 * C * C), rows * 8.0 * C);
    hipLaunchKernelGGL((conv_unit_fused_kernel<C, WAVES, HC, XH>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, s, w, x, y, batch,
                       frames);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

bool conv_unit_fused_supported(int c) { return c == 24 || c == 48 || c == 96; }

// In-place use is NOT allowed: a neighbouring wave's halo rows could already have been overwritten.
int launch_conv_unit_fused(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames, bool split, int ring) {
    L3AC_REQUIRE(x != y, "conv_unit_fused: in-place operation is not supported");
    if (split && w.ring_img && (ring == 2 ? conv_unit_ring_supported(w.c) : (ring == 1 && conv_unit_ring_preferred(w.c))))
        return launch_conv_unit_ring(s, w, x, y, batch, frames);
    if (split && w.w1_img && w.w2_img) return launch_conv_unit_split(s, w, x, y, batch, frames);
    switch (w.c) {
        case 24: return launch_fused<24, 8, 96, 1>(s, w, x, y, batch, frames, "conv_unit_fused_kernel<24>");
        case 48: return launch_fused<48, 8, 192, 1>(s, w, x, y, batch, frames, "conv_unit_fused_kernel<48>");
        case 96: return launch_fused<96, 8, 32, 2>(s, w, x, y, batch, frames, "conv_unit_fused_kernel<96>");
        default:
            l3ac_set_error("conv_unit_fused: C=%d not supported", w.c);
            return L3AC_EINVAL;
    }
}
