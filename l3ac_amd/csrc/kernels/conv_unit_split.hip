// Fused ConvUnit (C = 24 / 48; C = 96 until round 4: conv_unit_wide_kernel<96> took that width over) with both channel contractions on the bf16 matrix cores at fp32 accuracy
// ("bf16x3", split_bf16.hpp); reference l3ac/modules.py:10-41 + Residual (l3ac/xtract/nn/layers.py:59-62):
//
//     y = x + pw_conv2( GRN( snake( pw_conv1( LayerNorm( dw_conv7(x) ) ) ) ) )
//
// Same walk as conv_unit_fused.hip (one wave = 32 frames end to end, products evaluated transposed so the hidden tile
// stays in the accumulator registers and is the B operand of the second product), with these differences:
//   * the LayerNorm output a[frame][channel] — already held by the lane that feeds it to the MFMAs — is split ONCE per
//     tile into three bf16 planes in registers; lane half h holds channels 8q + 4h + {0..3}, which is exactly the k order
//     split_sigma() of an accumulator tile, so the same fragment order serves both products;
//   * every hidden tile X^T (32 hidden x 32 frames, fp32 accumulators) goes through snake / GRN and is split in
//     registers (split_acc_tile) before it multiplies W2;
//   * W1 / W2 come as fragment-ordered bf16x3 images built at model-load time (conv_unit_w{1,2}_image): a fragment is
//     one conflict-free 1-KB read, 3 reads per 6 MFMAs; the images of one 32-channel hidden tile are contiguous, so
//     weight chunks are copied to LDS verbatim.  C = 24: resident.  C = 48 / 96: streamed in chunks of 64 / 32 hidden
//     channels, double-buffered (chunk c+1 in flight in registers while chunk c is multiplied, one barrier per chunk);
//   * padding rows of a partial output tile (C = 24, 48) are not stored in the image: their lanes read the tile's last
//     real row and produce accumulator rows that are never written.
// MFMA cycles per 32 frames: 12 C/32 (ceil(C/16) + 2 ceil(C/32)) x 6 x 32  vs  (C/2 + 16 ceil(C/32)) (4C/32) x 64 for the
// fp32 kernel (C = 96: 27.6 k vs 73.7 k).
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "lane_sums.hpp"
#include "split_bf16.hpp"

#include <vector>

namespace {

template <int C, int HC, int XH>
struct SGeo {
    static constexpr int H4 = 4 * C;
    static constexpr int NCH = H4 / HC;        // weight chunks (1 = resident)
    static constexpr int NTC = HC / 32;        // hidden tiles per chunk
    static constexpr int CT = (C + 31) / 32;   // output-channel tiles
    static constexpr int KQ = C / 8;
    static constexpr int NS1 = (C + 15) / 16;  // k steps of the first product
    static constexpr int XC = C / XH;
    static constexpr int XS = XC + 4;
    static constexpr int ROWS = 38;
    // image geometry, bytes per hidden tile
    static constexpr int W1_TILE = NS1 * 3 * 1024;
    static constexpr int W2_TILE = 192 * C;    // sum over output tiles of 2 steps x 3 planes x 2 halves x rows x 16 B
    static constexpr int CHUNK1 = NTC * W1_TILE;
    static constexpr int CHUNK2 = NTC * W2_TILE;
    static constexpr int WBUF = CHUNK1 + CHUNK2;
    static constexpr int NBUF = NCH == 1 ? 1 : 2;
    // LDS carve: weight buffers (bytes), then floats
    static constexpr int OFF_P = NBUF * WBUF / 4;             // alpha[H4], 1/alpha[H4], gamma[H4], beta[H4]
    static constexpr int OFF_B1 = OFF_P + H4 * 4;
    static constexpr int OFF_B2 = OFF_B1 + H4;
    static constexpr int OFF_DW = OFF_B2 + 32 * CT;           // dw_w [7][C], dw_b, ln_w, ln_b
    static constexpr int OFF_XS = OFF_DW + 10 * C;
    static constexpr int xs_floats = ROWS * XS;
    static constexpr int lds_floats(int waves) { return OFF_XS + waves * xs_floats; }
    static_assert(H4 % HC == 0 && HC % 32 == 0 && C % (8 * XH) == 0 && WBUF % 16 == 0, "bad geometry");
};

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C, int WAVES, int HC, int XH>
__global__ __launch_bounds__(64 * WAVES, 2) void conv_unit_split_kernel(const ConvUnitW w, const float* __restrict__ x,
                                                                    float* __restrict__ y, int batch, int frames) {
    using G = SGeo<C, HC, XH>;
    constexpr bool RESIDENT = G::NCH == 1;
    constexpr int THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* Wb = reinterpret_cast<unsigned char*>(smem);
    float* Ps = smem + G::OFF_P;
    float* B1s = smem + G::OFF_B1;
    float* B2s = smem + G::OFF_B2;
    float* DWs = smem + G::OFF_DW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xs = smem + G::OFF_XS + wave * G::xs_floats;

    // weight chunk c = the images of hidden tiles [c NTC, (c+1) NTC): CHUNK1 bytes of w1_img then CHUNK2 bytes of w2_img
    constexpr int NW = (G::WBUF / 16 + THREADS - 1) / THREADS;
    u32x4 wpre[NW];
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int off = 16 * (tid + THREADS * k);
            wpre[k] = u32x4{0u, 0u, 0u, 0u};
            if (off < G::CHUNK1)
                wpre[k] = *reinterpret_cast<const u32x4*>(w.w1_img + (int64_t)chunk * G::CHUNK1 + off);
            else if (off < G::WBUF)
                wpre[k] = *reinterpret_cast<const u32x4*>(w.w2_img + (int64_t)chunk * G::CHUNK2 + (off - G::CHUNK1));
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int off = 16 * (tid + THREADS * k);
            if (off < G::WBUF) *reinterpret_cast<u32x4*>(Wb + buf * G::WBUF + off) = wpre[k];
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
    load_chunk(0);
    store_chunk(0);  // streamed variant: made visible by the barrier of the first chunk step
    __syncthreads();

    const int lj = lane & 31;  // frame within the tile (MFMA column) / weight row within a tile (A operand)
    const int lh = lane >> 5;
    const int tiles_per_clip = (frames + 31) / 32;
    const int64_t n_tiles = (int64_t)batch * tiles_per_clip;
    // per-lane offsets into the weight images: W1 rows are always 32 per block, W2 row blocks hold only the real rows
    const int w1_lane = (lh * 32 + lj) * 16;
    int w2_lane[G::CT], w2_rows[G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
        w2_rows[ct] = C - 32 * ct < 32 ? C - 32 * ct : 32;
        w2_lane[ct] = 6144 * ct + (lh * w2_rows[ct] + (lj < w2_rows[ct] ? lj : w2_rows[ct] - 1)) * 16;
    }

    constexpr int NPRE = (G::ROWS * (G::XC / 4) + 63) / 64;
    int pre_row[NPRE], pre_src[NPRE], pre_dst[NPRE];
#pragma unroll
    for (int it = 0; it < NPRE; ++it) {
        const int i = lane + 64 * it;
        const int row = i / (G::XC / 4), ch = i % (G::XC / 4);
        pre_row[it] = i < G::ROWS * (G::XC / 4) ? row - 3 : -(1 << 28);
        pre_src[it] = (row - 3) * C + 4 * ch;
        pre_dst[it] = i < G::ROWS * (G::XC / 4) ? row * G::XS + 4 * ch : -1;
    }
    // (clip, tile within the clip) of a tile index are carried incrementally from one pass to the next: the wave-uniform 64-bit
    // tile / tiles_per_clip and % of the first version were ~100 scalar instructions each, four per tile and wave
    auto load_rows = [&](bool ok, int bb_, int ti_, int xh, float4 (&pre)[NPRE]) {
        const int bb = ok ? bb_ : 0;
        const int tt0 = ok ? ti_ * 32 : -(1 << 28);
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
    const int stride_b = (int)((unsigned)tile_stride / (unsigned)tiles_per_clip), stride_t = (int)((unsigned)tile_stride % (unsigned)tiles_per_clip);
    const unsigned first_tile = (unsigned)blockIdx.x * WAVES + (unsigned)wave;  // (the launcher keeps n_tiles below 2^31)
    int cur_b = (int)(first_tile / (unsigned)tiles_per_clip), cur_t = (int)(first_tile % (unsigned)tiles_per_clip);
    float4 pre[XH == 1 ? NPRE : 1];
    if constexpr (XH == 1) load_rows((int64_t)first_tile < n_tiles, cur_b, cur_t, 0, pre);
    int gchunk = 0;  // chunk steps taken so far (streamed variant): its parity selects the LDS buffer

    // every wave of the block runs the same number of iterations (the streamed variant has block barriers inside)
    for (int64_t base = (int64_t)blockIdx.x * WAVES; base < n_tiles; base += tile_stride) {
        const int64_t tile = base + wave;
        const bool tile_ok = tile < n_tiles;
        const int b = tile_ok ? cur_b : 0;
        const int t0 = tile_ok ? cur_t * 32 : 0;
        const float* clip = x + (int64_t)b * frames * C;
        // the next pass's tile of this wave
        int nxt_b = cur_b + stride_b, nxt_t = cur_t + stride_t;
        if (nxt_t >= tiles_per_clip) {
            nxt_t -= tiles_per_clip;
            ++nxt_b;
        }

        // ---- depth-wise conv k7 + LayerNorm for frame lj, channels k = 8q + 4 lh + r (as conv_unit_fused.hip) ----
        float a[4 * G::KQ];
        float s1 = 0.f;
#pragma unroll
        for (int xh = 0; xh < XH; ++xh) {
            if constexpr (XH == 1) {
                store_rows(pre);
            } else {
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
            if constexpr (XH == 1) load_rows(tile + tile_stride < n_tiles, nxt_b, nxt_t, 0, pre);
#pragma unroll
            for (int ql = 0; ql < G::KQ / XH; ++ql) {
                const int q = xh * (G::KQ / XH) + ql;
                const int k0 = 8 * q + 4 * lh;
                const int kl = 8 * ql + 4 * lh;
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
            if (xh + 1 < XH) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
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
        // ---- LayerNorm affine, then split: k step s of lane half lh = a[8s .. 8s+7] = channels split_sigma(s, lh, j) ----
        bf16x8 ap[G::NS1][3];
#pragma unroll
        for (int s = 0; s < G::NS1; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = 8 * s + j;  // a[] index; beyond the channel count: zero padding (C = 24)
                v[j] = 0.f;
                if (idx < 4 * G::KQ) {
                    const int ch = 8 * (idx >> 2) + 4 * lh + (idx & 3);
                    v[j] = frame_ok ? (a[idx] - mean) * rstd * DWs[8 * C + ch] + DWs[9 * C + ch] : 0.f;
                }
            }
            unsigned p[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[s][pl] = __builtin_bit_cast(bf16x8, u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
        }

        // ---- output accumulators start at the pw_conv2 bias ---------------------------------------------
        f32x16_t yacc[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ct][r] = B2s[32 * ct + rowmap(r, lh)];

#pragma unroll 1
        for (int chunk = 0; chunk < G::NCH; ++chunk) {
            const unsigned char* wb = Wb + (RESIDENT ? 0 : (gchunk & 1) * G::WBUF);
            if (!RESIDENT) {
                // this chunk's buffer is complete (written during the previous step) and every wave is done reading the
                // other one, which the next chunk (in flight from here on) overwrites after the products
                __syncthreads();
                load_chunk(chunk + 1 < G::NCH ? chunk + 1 : 0);
            }
            const int n_base = chunk * HC;
            // X^T[n][m] = b1[n] + sum_k W1[n][k] a[m][k] for hidden tile ntl of this chunk
            auto first_product = [&](int ntl) __attribute__((always_inline)) -> f32x16_t {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = B1s[n_base + 32 * ntl + rowmap(r, lh)];
                const unsigned char* w1p = wb + ntl * G::W1_TILE + w1_lane;
#pragma unroll
                for (int s = 0; s < G::NS1; ++s) {
                    bf16x8 wf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(w1p + (s * 3 + pl) * 1024);
                    acc = mfma_split(wf, ap[s], acc);
                }
                return acc;
            };
            // software pipeline inside a chunk: tile ntl+1's first product (MFMA pipe) beside tile ntl's activation (VALU)
            f32x16_t xnext = first_product(0);
#pragma unroll 1
            for (int ntl = 0; ntl < G::NTC; ++ntl) {
                f32x16_t xacc = xnext;
                if (G::NTC > 1 && ntl + 1 < G::NTC) xnext = first_product(ntl + 1);
                // snake + GRN (normaliser == 1) on the accumulator registers (layers.py:29-33, :112-115)
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
                    const f32x2 sv = snake_act2(hv, al, ia);
                    const f32x2 o = __builtin_elementwise_fma(ga, sv, be) + sv;
                    xacc[r] = o.x;
                    xacc[r + 1] = o.y;
                }
                // Y^T[c][m] += sum_n W2[c][n] X^T[n][m]: the split accumulator tile is the B operand
                bf16x8 xb[2][3];
                split_acc_tile(xacc, xb);
                const unsigned char* w2p = wb + G::CHUNK1 + ntl * G::W2_TILE;
#pragma unroll
                for (int ct = 0; ct < G::CT; ++ct) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 wf[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[pl] = *reinterpret_cast<const bf16x8*>(w2p + w2_lane[ct] + (s * 3 + pl) * 2 * w2_rows[ct] * 16);
                        yacc[ct] = mfma_split(wf, xb[s], yacc[ct]);
                    }
                }
            }
            if (!RESIDENT) {
                store_chunk((gchunk + 1) & 1);
                ++gchunk;
            }
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
        cur_b = nxt_b;
        cur_t = nxt_t;
    }
}

template <int C, int WAVES, int HC, int XH>
int launch_split(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames, const char* name) {
    using G = SGeo<C, HC, XH>;
    const size_t lds = (size_t)G::lds_floats(WAVES) * sizeof(float);
    static_assert(G::lds_floats(WAVES) * sizeof(float) <= 160 * 1024, "LDS budget exceeded");
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_unit_split_kernel<C, WAVES, HC, XH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured.done();
    }
    const int64_t tiles = (int64_t)batch * ((frames + 31) / 32);
    L3AC_REQUIRE(tiles < ((int64_t)1 << 31) - 65536, "conv_unit_split: too many tiles");
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    int64_t blocks = ceil_div64(tiles, WAVES);
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    const double rows = (double)batch * frames;
    ProfScope prof(s, name, rows * (14.0 * C + 16.0 * C * C), rows * 8.0 * C);
    hipLaunchKernelGGL((conv_unit_split_kernel<C, WAVES, HC, XH>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, s, w, x, y, batch,
                       frames);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

// x must not alias y
int launch_conv_unit_split(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames) {
    L3AC_REQUIRE(x != y && w.w1_img && w.w2_img, "conv_unit_split: bad arguments");
    switch (w.c) {
        case 24: return launch_split<24, 8, 96, 1>(s, w, x, y, batch, frames, "conv_unit_split_kernel<24>");
        case 48: return launch_split<48, 8, 64, 1>(s, w, x, y, batch, frames, "conv_unit_split_kernel<48>");
        default:
            l3ac_set_error("conv_unit_split: C=%d not supported", w.c);
            return L3AC_EINVAL;
    }
}

// ---- host builders of the fragment-ordered weight images --------------------------------------------------------------
// w1 [4C][C]: per hidden tile nt, k step s, plane p, lane half h, row r (32): 8 bf16 = W1[32 nt + r][split_sigma(s, h, j)]
std::vector<unsigned char> conv_unit_w1_image(const float* w1, int c) {
    const int h4 = 4 * c, ns1 = (c + 15) / 16;
    std::vector<unsigned char> img((size_t)(h4 / 32) * ns1 * 3 * 1024, 0);
    for (int nt = 0; nt < h4 / 32; ++nt)
        for (int s = 0; s < ns1; ++s)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 32; ++r)
                    for (int j = 0; j < 8; ++j) {
                        const int ch = split_sigma(s, h, j);
                        if (ch >= c) continue;
                        uint16_t pl[3];
                        split3_host(w1[(size_t)(32 * nt + r) * c + ch], pl);
                        for (int p = 0; p < 3; ++p)
                            std::memcpy(img.data() + (size_t)nt * ns1 * 3072 + ((((size_t)s * 3 + p) * 2 + h) * 32 + r) * 16 + 2 * j,
                                        &pl[p], 2);
                    }
    return img;
}

// w2 [C][4C]: per hidden tile nt, output tile ct (rows = min(32, C - 32 ct)), k step s (2), plane, lane half, row:
// 8 bf16 = W2[32 ct + r][32 nt + split_sigma(s, h, j)]
std::vector<unsigned char> conv_unit_w2_image(const float* w2, int c) {
    const int h4 = 4 * c, ct_n = (c + 31) / 32;
    std::vector<unsigned char> img((size_t)(h4 / 32) * 192 * c, 0);
    for (int nt = 0; nt < h4 / 32; ++nt)
        for (int ct = 0; ct < ct_n; ++ct) {
            const int rows = c - 32 * ct < 32 ? c - 32 * ct : 32;
            for (int s = 0; s < 2; ++s)
                for (int h = 0; h < 2; ++h)
                    for (int r = 0; r < rows; ++r)
                        for (int j = 0; j < 8; ++j) {
                            uint16_t pl[3];
                            split3_host(w2[(size_t)(32 * ct + r) * h4 + 32 * nt + split_sigma(s, h, j)], pl);
                            for (int p = 0; p < 3; ++p)
                                std::memcpy(img.data() + (size_t)nt * 192 * c + (size_t)6144 * ct +
                                                ((((size_t)s * 3 + p) * 2 + h) * rows + r) * 16 + 2 * j,
                                            &pl[p], 2);
                        }
        }
    return img;
}
