// Encoder stem "FirstBlock" (reference l3ac/tconv/__init__.py:8-27, tconv/base.py:8-45):
//   5 trend branches  p_k = avg_pool(max_pool(|x|, k), k), k in {1 (identity, no abs), 5, 11, 21, 45}
//   -> weight-normed Conv1d(1 -> 4, k7, pad 3) each -> concat 20 ch -> 1x1 conv 20 -> 80 -> exact GELU
//   -> concat the raw sample (81 ch) -> 1x1 conv 81 -> d0.
// audio [batch][samples] -> y [batch][frames][d0]; frames >= samples, the tail is the zero right-padding of
// Codec.preprocess (l3ac/codec.py:79-84), folded into the load.
//
// One block = 256 consecutive frames of one clip.  The pooled signals are built in LDS (halo 3 + 2*(k/2) per
// side); pooling semantics follow ATen: max_pool pads -inf, avg_pool pads 0 and always divides by k
// (count_include_pad), its window summed left to right.  The 1x1 convs run on the VALU with wave-uniform
// (scalar-loaded) weights: ~3.7 kFMA per frame against 4 B read + 4*d0 B written — compute-bound on fp32 VALU.
#include "../kernels.hpp"
#include "device_math.hpp"
#include "ring_common.hpp"

namespace {

constexpr int TILE = 256;
constexpr int HALO = 47;  // 3 (conv) + 22 (avg 45) + 22 (max 45)


// UNR: unroll of the 80-channel loop.  4 when the grid leaves the SIMDs a single wave each (one clip): the loop then runs at the
// latency of its scalar weight loads, and four iterations' loads in flight are worth 76 -> 59 us; with 8 waves per SIMD (a batch)
// the loads are covered anyway and the extra registers cost a wave of occupancy.  Same fmaf order either way.
// the 20 branch channels of frame t0 + tid (tconv/__init__.py:25-27, tconv/base.py:8-45): shared by both forms of the kernel.
// xs [TILE + 2 HALO] holds the samples t0 - HALO .. (zero outside the clip); mbuf / pbuf are scratch.  All 256 threads take part.
__device__ __forceinline__ void first_block_branches(const FirstBlockW& w, const float* xs, float* mbuf, float* pbuf, int t0, int frames,
                                                     float (&h)[20]) {
    const int tid = threadIdx.x;
    // branch 0: identity pool (tconv/base.py:13), conv over the raw samples
    {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float acc = w.tb[o];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[o * 7 + j], xs[HALO + tid + j - 3], acc);
            h[o] = acc;
        }
    }
    const int pool_k[4] = {5, 11, 21, 45};
#pragma unroll
    for (int br = 0; br < 4; ++br) {
        const int k = pool_k[br];
        const int hk = k >> 1;
        // m[v] for v in [t0 - 3 - hk, t0 + TILE + 3 + hk): max over the in-clip part of the window; 0 outside the clip
        const int m_len = TILE + 6 + 2 * hk;
        for (int i = tid; i < m_len; i += TILE) {
            const int v = t0 - 3 - hk + i;
            float m = 0.f;
            if (v >= 0 && v < frames) {
                const int base = v - hk - (t0 - HALO);  // xs index of sample v - hk
                for (int s = 0; s < k; ++s) m = fmaxf(m, fabsf(xs[base + s]));
            }
            mbuf[i] = m;
        }
        __syncthreads();
        // p[u] for u in [t0 - 3, t0 + TILE + 3): left-to-right window sum / k; 0 outside the clip (conv zero pad)
        for (int i = tid; i < TILE + 6; i += TILE) {
            const int u = t0 - 3 + i;
            float pv = 0.f;
            if (u >= 0 && u < frames) {
                float sum = 0.f;
                for (int s = 0; s < k; ++s) {
                    const int v = u - hk + s;
                    if (v >= 0 && v < frames) sum += mbuf[i + s];
                }
                pv = sum / (float)k;
            }
            pbuf[i] = pv;
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int ch = (br + 1) * 4 + o;
            float acc = w.tb[ch];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[ch * 7 + j], pbuf[tid + j], acc);
            h[ch] = acc;
        }
        __syncthreads();
    }
}

// UNR: unroll of the 80-channel loop.  4 when the grid leaves the SIMDs a single wave each (one clip): the loop then runs at the
// latency of its scalar weight loads, and four iterations' loads in flight are worth 76 -> 59 us; with 8 waves per SIMD (a batch)
// the loads are covered anyway and the extra registers cost a wave of occupancy.  Same fmaf order either way.
template <int D0, int UNR>
__global__ __launch_bounds__(TILE) void first_block_kernel(const FirstBlockW w, const float* __restrict__ audio,
                                                          int64_t audio_stride, int samples, int frames,
                                                          float* __restrict__ y) {
    __shared__ float xs[TILE + 2 * HALO];
    __shared__ float mbuf[TILE + 6 + 44];
    __shared__ float pbuf[TILE + 6];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const float* clip = audio + (int64_t)b * audio_stride;

    for (int i = tid; i < TILE + 2 * HALO; i += TILE) {
        const int u = t0 - HALO + i;
        xs[i] = (u >= 0 && u < samples) ? clip[u] : 0.f;
    }
    __syncthreads();

    float h[20];
    first_block_branches(w, xs, mbuf, pbuf, t0, frames, h);

    const int t = t0 + tid;
    if (t >= frames) return;

    float out[D0];
#pragma unroll
    for (int d = 0; d < D0; ++d) out[d] = w.b2[d];
#pragma unroll UNR
    for (int o = 0; o < 80; ++o) {
        float s = w.b1[o];
#pragma unroll
        for (int i = 0; i < 20; ++i) s = fmaf(w.w1[o * 20 + i], h[i], s);
        const float g = gelu_erf(s);
#pragma unroll
        for (int d = 0; d < D0; ++d) out[d] = fmaf(w.w2[o * D0 + d], g, out[d]);
    }
    const float xv = xs[HALO + tid];
#pragma unroll
    for (int d = 0; d < D0; ++d) out[d] = fmaf(w.w2[80 * D0 + d], xv, out[d]);

    float* dst = y + ((int64_t)b * frames + t) * D0;
#pragma unroll
    for (int d = 0; d < D0; d += 4) *reinterpret_cast<float4*>(dst + d) = make_float4(out[d], out[d + 1], out[d + 2], out[d + 3]);
}

// ---- the two 1x1 convs on the bf16 matrix cores (bf16x3: fp32 accuracy, split_bf16.hpp) ---------------------------------------
// 3.5 of the stem's 3.7 kFMA per sample are its channel contractions 20 -> 80 and 81 -> d0: on the vector unit (the kernel above, kept
// for the exact-fp32 route) they run at 0.37 of the VALU peak.  Here they are "weights (A) x activations (B)" products on
// v_mfma_f32_16x16x32_bf16 as in the other fused kernels (ring_common.hpp): hidden^T [80][frames] = W1 . h^T with b1 as the
// accumulator's initial value; exact GELU on the accumulator registers; the five hidden tiles plus one tile that holds the raw sample
// in its first row are directly the B operand (k = 96) of out^T [d0][frames] = [W2 | w_x] . [gelu(hidden) ; x].
//   * Weight pieces (16 rows x 32 k, 3 planes) are built ONCE per workgroup in LDS from the fp32 tensors: 5 of W1 (k 20 -> 32 zero
//     padded), 3 per 16 output channels of conv_2 (k 81 -> 96).
//   * The branch channels are computed with the frame on the lane (as above) and change hands to the B-operand layout — lane (frame
//     ln, k group g) holds channels sigma(g, .) — through a per-wave LDS buffer, 32 frames at a time.
// A wave owns 64 frames: two halves of 32 (two column groups of 16 each) so that the hidden accumulators are 40 registers.
template <int D0>
struct FbGeo {
    static constexpr int NRT = (D0 + 15) / 16;           // 16-row tiles of output channels
    static constexpr int W1_PIECES = 5, W2_PIECES = 3 * NRT, PIECES = W1_PIECES + W2_PIECES;
    static constexpr int HS_STRIDE = 24;                  // floats per frame row of the hand-over buffer (20 channels + pad, 16-B rows)
    static constexpr int OFF_W = 0;                       // pieces
    static constexpr int OFF_B1 = OFF_W + PIECES * 3072;  // b1 [80] | b2 [32]
    static constexpr int OFF_HS = OFF_B1 + (80 + 32) * 4; // [4 waves][32 frames][HS_STRIDE]
    static constexpr int OFF_XS = OFF_HS + 4 * 32 * HS_STRIDE * 4;
    static constexpr int OFF_MB = OFF_XS + (TILE + 2 * HALO) * 4;
    static constexpr int OFF_PB = OFF_MB + (TILE + 6 + 44) * 4;
    static constexpr int LDS = OFF_PB + (TILE + 6) * 4 + 8;
};

template <int D0>
__global__ __launch_bounds__(TILE, 3) void first_block_mfma_kernel(const FirstBlockW w, const float* __restrict__ audio, int64_t audio_stride,
                                                                   int samples, int frames, float* __restrict__ y) {
    using G = FbGeo<D0>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fb[];
    unsigned char* const Wp = smem_fb + G::OFF_W;
    float* const Bs = reinterpret_cast<float*>(smem_fb + G::OFF_B1);
    float* const xs = reinterpret_cast<float*>(smem_fb + G::OFF_XS);
    float* const mbuf = reinterpret_cast<float*>(smem_fb + G::OFF_MB);
    float* const pbuf = reinterpret_cast<float*>(smem_fb + G::OFF_PB);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const float* clip = audio + (int64_t)b * audio_stride;
    for (int i = tid; i < TILE + 2 * HALO; i += TILE) {
        const int u = t0 - HALO + i;
        xs[i] = (u >= 0 && u < samples) ? clip[u] : 0.f;
    }
    // ---- weight pieces: word (piece q, lane slot ls = 16 g + m, pair jp) holds W[row0 + m][k0 + sigma(g, 2 jp)], [.. + 1] as three
    // bf16 pairs (ring_put_piece's layout: plane p at q * 3072 + p * 1024 + ls * 16 + 4 jp)
    for (int e = tid; e < G::PIECES * 256; e += TILE) {
        const int q = e >> 8, ls = (e >> 2) & 63, jp = e & 3;
        const int g = ls >> 4, m = ls & 15;
        const int j0 = 2 * jp;
        const int kk = (j0 < 4 ? 4 * g + j0 : 16 + 4 * g + j0 - 4);  // sigma(g, j0); j0 + 1 is the next k
        float v0 = 0.f, v1 = 0.f;
        if (q < G::W1_PIECES) {  // conv_1 [80][20], k step 0
            const int row = 16 * q + m;
            if (kk < 20) v0 = w.w1[row * 20 + kk];
            if (kk + 1 < 20) v1 = w.w1[row * 20 + kk + 1];
        } else {  // conv_2 given transposed [81][d0]: row = output channel, k = hidden channel (80: the raw sample)
            const int q2 = q - G::W1_PIECES, rt = q2 / 3, sstep = q2 % 3;
            const int row = 16 * rt + m, k = 32 * sstep + kk;
            if (row < D0 && k < 81) v0 = w.w2[k * D0 + row];
            if (row < D0 && k + 1 < 81) v1 = w.w2[(k + 1) * D0 + row];
        }
        unsigned p0, p1, p2;
        split2(v0, v1, p0, p1, p2);
        unsigned char* dst = Wp + q * 3072 + ls * 16 + 4 * jp;
        *reinterpret_cast<unsigned*>(dst) = p0;
        *reinterpret_cast<unsigned*>(dst + 1024) = p1;
        *reinterpret_cast<unsigned*>(dst + 2048) = p2;
    }
    if (tid < 80) Bs[tid] = w.b1[tid];
    if (tid >= 96 && tid < 128) Bs[80 + tid - 96] = tid - 96 < D0 ? w.b2[tid - 96] : 0.f;
    __syncthreads();

    float h[20];
    first_block_branches(w, xs, mbuf, pbuf, t0, frames, h);  // (ends with a barrier: the pieces are complete as well)

    float* const hs = reinterpret_cast<float*>(smem_fb + G::OFF_HS) + wave * (32 * G::HS_STRIDE);
    const unsigned char* const wl = Wp + 16 * lane;
    const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // ---- hand-over: the lanes of this half write their 20 channels (channel c at float c of the frame's row) -----------------
        if ((lane >> 5) == half) {
            float* row = hs + (lane & 31) * G::HS_STRIDE;
#pragma unroll
            for (int c = 0; c < 20; c += 4) *reinterpret_cast<f32x4_t*>(row + c) = f32x4_t{h[c], h[c + 1], h[c + 2], h[c + 3]};
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own writes (one wave = one instruction stream)
        __builtin_amdgcn_wave_barrier();
        bf16x8 hp[2][3];  // B operand of column group fh: lane (ln, g): channels 4 g .. 4 g + 3 and (g = 0 only) 16 .. 19
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
            const float* row = hs + (16 * fh + ln) * G::HS_STRIDE;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(row + 4 * lg);
            f32x4_t hi = *reinterpret_cast<const f32x4_t*>(row + 16);
            hi = lg == 0 ? hi : zero4;
            planes_of(lo, hi, hp[fh]);
        }
        __builtin_amdgcn_wave_barrier();  // (the other half's lanes overwrite the rows next iteration)
        // ---- hidden^T = W1 . h^T + b1, exact GELU ---------------------------------------------------------------------------------
        f32x4_t ha[6][2];
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(Bs + 16 * rt + 4 * lg);
            bf16x8 wf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wl + rt * 3072 + 1024 * pl);
#pragma unroll
            for (int fh = 0; fh < 2; ++fh) ha[rt][fh] = mfma6(wf, hp[fh], bv);
        }
#pragma unroll
        for (int rt = 0; rt < 5; ++rt)
#pragma unroll
            for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                for (int i = 0; i < 4; ++i) ha[rt][fh][i] = gelu_erf(ha[rt][fh][i]);
        // tile 5: row 80 = the raw sample of the lane's frame (lanes of k group 0, element 0), zeros below
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
            const float xv = xs[HALO + 64 * wave + 32 * half + 16 * fh + ln];
            ha[5][fh] = f32x4_t{lg == 0 ? xv : 0.f, 0.f, 0.f, 0.f};
        }
        // ---- out^T = [W2 | w_x] . [gelu(hidden) ; x] + b2 ------------------------------------------------------------------------
        f32x4_t oa[G::NRT][2];
#pragma unroll
        for (int rt = 0; rt < G::NRT; ++rt) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(Bs + 80 + 16 * rt + 4 * lg);
            oa[rt][0] = bv;
            oa[rt][1] = bv;
        }
#pragma unroll
        for (int sstep = 0; sstep < 3; ++sstep) {
            bf16x8 xp[2][3];
#pragma unroll
            for (int fh = 0; fh < 2; ++fh) planes_of(ha[2 * sstep][fh], ha[2 * sstep + 1][fh], xp[fh]);
#pragma unroll
            for (int rt = 0; rt < G::NRT; ++rt) {
                bf16x8 wf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wl + (G::W1_PIECES + 3 * rt + sstep) * 3072 + 1024 * pl);
#pragma unroll
                for (int fh = 0; fh < 2; ++fh) oa[rt][fh] = mfma6(wf, xp[fh], oa[rt][fh]);
            }
        }
        // ---- store: lane (frame ln of group fh, k group g) owns channels 16 rt + 4 g .. + 3 ----------------------------------------
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
            const int t = t0 + 64 * wave + 32 * half + 16 * fh + ln;
            if (t < frames) {
                float* dst = y + ((int64_t)b * frames + t) * D0;
#pragma unroll
                for (int rt = 0; rt < G::NRT; ++rt)
                    if (16 * rt + 4 * lg < D0) *reinterpret_cast<f32x4_t*>(dst + 16 * rt + 4 * lg) = oa[rt][fh];
            }
        }
    }
}

}  // namespace

// w.w2 is expected TRANSPOSED: [81][d0] (done once at weight upload).  `split`: the context's GEMM route — true: the two 1x1 convs on
// the bf16 matrix cores (bf16x3), false: everything on the fp32 vector unit (the exact route).
int launch_first_block(hipStream_t s, const FirstBlockW& w, const float* audio, int64_t audio_stride, int batch,
                       int samples, int frames, float* y, bool split) {
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && frames >= samples && samples > 0, "first_block: bad shape");
    const dim3 grid((unsigned)ceil_div64(frames, TILE), (unsigned)batch);
    ProfScope prof(s, split ? "first_block_mfma_kernel" : "first_block_kernel", 2.0 * (140.0 + 1600.0 + 81.0 * w.d0 + 164.0) * batch * frames,
                   4.0 * ((double)batch * samples + (double)batch * frames * w.d0));
    const bool few = (int64_t)grid.x * grid.y <= 512;  // at most two workgroups per CU
    static PerDeviceOnce configured;
    if (split && configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(first_block_mfma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, FbGeo<8>::LDS));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(first_block_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, FbGeo<16>::LDS));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(first_block_mfma_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, FbGeo<24>::LDS));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(first_block_mfma_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, FbGeo<32>::LDS));
        configured.done();
    }
    switch (w.d0) {
#define L3AC_FB_CASE(D)                                                                                                       \
    case D:                                                                                                                   \
        if (split) hipLaunchKernelGGL((first_block_mfma_kernel<D>), grid, dim3(TILE), FbGeo<D>::LDS, s, w, audio, audio_stride, samples, frames, y); \
        else if (few) hipLaunchKernelGGL((first_block_kernel<D, 4>), grid, dim3(TILE), 0, s, w, audio, audio_stride, samples, frames, y); \
        else hipLaunchKernelGGL((first_block_kernel<D, 1>), grid, dim3(TILE), 0, s, w, audio, audio_stride, samples, frames, y);    \
        break
        L3AC_FB_CASE(8);
        L3AC_FB_CASE(16);
        L3AC_FB_CASE(24);
        L3AC_FB_CASE(32);
#undef L3AC_FB_CASE
        default:
            l3ac_set_error("first_block: encoder_dims[0]=%d not in {8,16,24,32}", w.d0);
            return L3AC_EINVAL;
    }
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
