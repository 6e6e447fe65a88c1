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

namespace {

constexpr int TILE = 256;
constexpr int HALO = 47;  // 3 (conv) + 22 (avg 45) + 22 (max 45)


// The five trend branches of this thread's frame `f` of the workgroup's FT frames starting at t0: h[20] (tconv/base.py:8-45).  Every
// thread of the workgroup takes part (NT threads build the pooled signals in LDS); xs keeps the raw samples for the caller.
template <int FT, int NT>
__device__ __forceinline__ void first_block_trends(const FirstBlockW& w, const float* __restrict__ clip, const int t0, const int samples, const int frames,
                                                   const int tid, const int f, float* xs, float* mbuf, float* pbuf, float (&h)[20]) {
    for (int i = tid; i < FT + 2 * HALO; i += NT) {
        const int u = t0 - HALO + i;
        xs[i] = (u >= 0 && u < samples) ? clip[u] : 0.f;
    }
    __syncthreads();
    // branch 0: identity pool (tconv/base.py:13), conv over the raw samples
    {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float acc = w.tb[o];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[o * 7 + j], xs[HALO + f + j - 3], acc);
            h[o] = acc;
        }
    }
    const int pool_k[4] = {5, 11, 21, 45};
#pragma unroll
    for (int br = 0; br < 4; ++br) {
        const int k = pool_k[br];
        const int hk = k >> 1;
        // m[v] for v in [t0 - 3 - hk, t0 + FT + 3 + hk): max over the in-clip part of the window; 0 outside the clip
        const int m_len = FT + 6 + 2 * hk;
        for (int i = tid; i < m_len; i += NT) {
            const int v = t0 - 3 - hk + i;
            float m = 0.f;
            if (v >= 0 && v < frames) {
                const int base = v - hk - (t0 - HALO);  // xs index of sample v - hk
                for (int s = 0; s < k; ++s) m = fmaxf(m, fabsf(xs[base + s]));
            }
            mbuf[i] = m;
        }
        __syncthreads();
        // p[u] for u in [t0 - 3, t0 + FT + 3): left-to-right window sum / k; 0 outside the clip (conv zero pad)
        for (int i = tid; i < FT + 6; i += NT) {
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
            for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[ch * 7 + j], pbuf[f + j], acc);
            h[ch] = acc;
        }
        __syncthreads();
    }
}

// UNR: unroll of the 80-channel loop (1 for a batch: with 8 waves per SIMD the scalar weight loads are covered, and more registers
// cost a wave of occupancy).
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
    float h[20];
    first_block_trends<TILE, TILE>(w, audio + (int64_t)b * audio_stride, t0, samples, frames, tid, tid, xs, mbuf, pbuf, h);

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

// Few clips (a streaming chunk is ONE: 63 workgroups of the kernel above, one wave per SIMD on a quarter of the chip, each thread a
// serial chain of ~6.7 k vector instructions: 58 us): FOUR threads per frame, one per wave of a 64-frame workgroup.  Wave j computes
// the hidden channels 20 j .. 20 j + 19 of the workgroup's frames (the GELU is the expensive part) into LDS, then every wave
// accumulates ITS quarter of the output channels over all 80 hidden channels in the same order as the kernel above: per output the
// same fused multiply-adds in the same order, the same bits.  The trend branches are computed by all four waves (2 % of the work).
template <int D0>
__global__ __launch_bounds__(256) void first_block_split_kernel(const FirstBlockW w, const float* __restrict__ audio, int64_t audio_stride, int samples,
                                                                int frames, float* __restrict__ y) {
    constexpr int FT = 64, DP = D0 / 4;
    static_assert(D0 % 8 == 0, "two floats per store");
    __shared__ float xs[FT + 2 * HALO];
    __shared__ float mbuf[FT + 6 + 44];
    __shared__ float pbuf[FT + 6];
    __shared__ float gbuf[80][FT];
    const int tid = threadIdx.x;
    const int f = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FT;
    float h[20];
    first_block_trends<FT, 256>(w, audio + (int64_t)b * audio_stride, t0, samples, frames, tid, f, xs, mbuf, pbuf, h);
#pragma unroll 4
    for (int oo = 0; oo < 20; ++oo) {
        const int o = 20 * part + oo;
        float s = w.b1[o];
#pragma unroll
        for (int i = 0; i < 20; ++i) s = fmaf(w.w1[o * 20 + i], h[i], s);
        gbuf[o][f] = gelu_erf(s);
    }
    __syncthreads();
    const int t = t0 + f;
    if (t >= frames) return;
    float out[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) out[d] = w.b2[DP * part + d];
#pragma unroll 8
    for (int o = 0; o < 80; ++o) {
        const float g = gbuf[o][f];
#pragma unroll
        for (int d = 0; d < DP; ++d) out[d] = fmaf(w.w2[o * D0 + DP * part + d], g, out[d]);
    }
    const float xv = xs[HALO + f];
#pragma unroll
    for (int d = 0; d < DP; ++d) out[d] = fmaf(w.w2[80 * D0 + DP * part + d], xv, out[d]);
    float* dst = y + ((int64_t)b * frames + t) * D0 + DP * part;
#pragma unroll
    for (int d = 0; d < DP; d += 2) *reinterpret_cast<float2*>(dst + d) = make_float2(out[d], out[d + 1]);
}

}  // namespace

// w.w2 is expected TRANSPOSED: [81][d0] (done once at weight upload).
int launch_first_block(hipStream_t s, const FirstBlockW& w, const float* audio, int64_t audio_stride, int batch,
                       int samples, int frames, float* y) {
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && frames >= samples && samples > 0, "first_block: bad shape");
    const dim3 grid((unsigned)ceil_div64(frames, TILE), (unsigned)batch);
    ProfScope prof(s, "first_block_kernel", 2.0 * (140.0 + 1600.0 + 81.0 * w.d0 + 164.0) * batch * frames,
                   4.0 * ((double)batch * samples + (double)batch * frames * w.d0));
    // few clips: four threads per frame (first_block_split_kernel), up to four 64-frame workgroups per CU
    const dim3 grid_split((unsigned)ceil_div64(frames, 64), (unsigned)batch);
    const bool few = (int64_t)grid_split.x * grid_split.y <= 1024;
    switch (w.d0) {
#define L3AC_FB_CASE(D)                                                                                                       \
    case D:                                                                                                                   \
        if (few) hipLaunchKernelGGL((first_block_split_kernel<D>), grid_split, dim3(256), 0, s, w, audio, audio_stride, samples, frames, y); \
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
