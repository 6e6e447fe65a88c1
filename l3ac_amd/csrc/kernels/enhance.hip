// Decoder "EnhanceBlock" front end and the output head.
//
// EnhanceBlock (reference l3ac/tconv/__init__.py:30-44): from CHANNEL 0 of x only, four trend branches
//   p_k = avg_pool(max_pool(|x0|, k), k), k in {1 (identity), 3, 5, 9}
//   -> weight-normed Conv1d(1 -> 1, k7, dilation k//2 + 1 = {1, 2, 3, 5}, pad 3*dil)
//   -> InstanceNorm1d(4, affine) over the frames of each clip -> plain Conv1d(4 -> C, 1) -> x + y * x.
// Here: `enhance_branches` writes the four raw branch signals yi [batch][frames][4]; `enhance_stats` reduces
// them to mean / 1/sqrt(var + 1e-5) per (clip, branch); the normalise + merge + gate is the SRC_GATE row kernel.
//
// Output head (l3ac/modules.py:192-194): weight-normed Conv1d(c -> 1, k7, pad 3) -> tanh on the snake-activated
// last feature map.
#include "../kernels.hpp"
#include "lane_sums.hpp"

namespace {

constexpr int TILE = 256;
constexpr int XH = 23;  // 3*5 (conv reach at dilation 5) + 4 (avg 9) + 4 (max 9)

__global__ __launch_bounds__(TILE) void enhance_branches_kernel(const EnhanceW w, const float* __restrict__ x,
                                                               int frames, int c, float* __restrict__ yi) {
    __shared__ float xs[TILE + 2 * XH];
    __shared__ float mbuf[TILE + 30 + 8];
    __shared__ float pbuf[TILE + 30];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const float* clip = x + (int64_t)b * frames * c;
    for (int i = tid; i < TILE + 2 * XH; i += TILE) {
        const int u = t0 - XH + i;
        xs[i] = (u >= 0 && u < frames) ? clip[(int64_t)u * c] : 0.f;
    }
    __syncthreads();

    float out[4];
    {  // branch 0: identity pool, dilation 1
        float acc = w.tb[0];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[j], xs[XH + tid + j - 3], acc);
        out[0] = acc;
    }
    const int pool_k[3] = {3, 5, 9};
#pragma unroll
    for (int br = 0; br < 3; ++br) {
        const int k = pool_k[br];
        const int hk = k >> 1;
        const int dil = hk + 1;  // pool_kernel // dilation_rate(2) + 1 (tconv/base.py:34)
        const int reach = 3 * dil;
        const int m_len = TILE + 2 * reach + 2 * hk;
        for (int i = tid; i < m_len; i += TILE) {
            const int v = t0 - reach - hk + i;
            float m = 0.f;
            if (v >= 0 && v < frames) {
                const int base = v - hk - (t0 - XH);
                for (int s = 0; s < k; ++s) m = fmaxf(m, fabsf(xs[base + s]));
            }
            mbuf[i] = m;
        }
        __syncthreads();
        for (int i = tid; i < TILE + 2 * reach; i += TILE) {
            const int u = t0 - reach + i;
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
        float acc = w.tb[br + 1];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc = fmaf(w.tw[(br + 1) * 7 + j], pbuf[tid + j * dil], acc);
        out[br + 1] = acc;
        __syncthreads();
    }
    const int t = t0 + tid;
    if (t < frames)
        *reinterpret_cast<float4*>(yi + ((int64_t)b * frames + t) * 4) = make_float4(out[0], out[1], out[2], out[3]);
}

// one block per clip: two-pass mean / biased variance over frames for the 4 branch channels
__global__ __launch_bounds__(1024) void enhance_stats_kernel(const float* __restrict__ yi, int frames,
                                                            float* __restrict__ stats) {
    __shared__ float4 part[16];
    __shared__ float4 bc;
    const int b = blockIdx.x;
    const float4* src = reinterpret_cast<const float4*>(yi) + (int64_t)b * frames;
    auto block_sum = [&](float4 v) -> float4 {
        v = make_float4(wave_sum(v.x), wave_sum(v.y), wave_sum(v.z), wave_sum(v.w));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float4 s = part[0];
            for (int i = 1; i < 16; ++i) { s.x += part[i].x; s.y += part[i].y; s.z += part[i].z; s.w += part[i].w; }
            bc = s;
        }
        __syncthreads();
        return bc;
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = threadIdx.x; t < frames; t += 1024) {
        const float4 v = src[t];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 mean = block_sum(acc);
    const float inv_n = 1.0f / (float)frames;
    mean.x *= inv_n; mean.y *= inv_n; mean.z *= inv_n; mean.w *= inv_n;
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = threadIdx.x; t < frames; t += 1024) {
        const float4 v = src[t];
        const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
        acc.x += dx * dx; acc.y += dy * dy; acc.z += dz * dz; acc.w += dw * dw;
    }
    const float4 var = block_sum(acc);
    if (threadIdx.x == 0) {
        float* o = stats + (int64_t)b * 8;
        o[0] = mean.x; o[1] = mean.y; o[2] = mean.z; o[3] = mean.w;
        o[4] = 1.0f / sqrtf(var.x * inv_n + 1e-5f);
        o[5] = 1.0f / sqrtf(var.y * inv_n + 1e-5f);
        o[6] = 1.0f / sqrtf(var.z * inv_n + 1e-5f);
        o[7] = 1.0f / sqrtf(var.w * inv_n + 1e-5f);
    }
}

// head: x [batch][frames][c] (already snake-activated) -> audio [batch][frames]
__global__ __launch_bounds__(TILE) void head_kernel(const float* __restrict__ x, int frames, int c,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ audio, const int pretanh) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * TILE + threadIdx.x;
    if (t >= frames) return;
    const float* clip = x + (int64_t)b * frames * c;
    float acc = bias[0];
    for (int j = 0; j < 7; ++j) {
        const int ts = t + j - 3;
        if (ts < 0 || ts >= frames) continue;
        const float4* row = reinterpret_cast<const float4*>(clip + (int64_t)ts * c);
        const float4* wr = reinterpret_cast<const float4*>(w + j * c);
        for (int q = 0; q < (c >> 2); ++q) {
            const float4 xv = row[q];
            const float4 wv = wr[q];
            acc = fmaf(wv.x, xv.x, acc);
            acc = fmaf(wv.y, xv.y, acc);
            acc = fmaf(wv.z, xv.z, acc);
            acc = fmaf(wv.w, xv.w, acc);
        }
    }
    audio[(int64_t)b * frames + t] = pretanh ? acc : tanhf(acc);
}

}  // namespace

int launch_enhance_branches(hipStream_t s, const EnhanceW& w, const float* x, int batch, int frames, int c, float* yi) {
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && frames > 0, "enhance: bad shape");
    ProfScope prof(s, "enhance_branches_kernel", 2.0 * (28.0 + 34.0) * batch * frames, 4.0 * 5.0 * batch * frames);
    hipLaunchKernelGGL(enhance_branches_kernel, dim3((unsigned)ceil_div64(frames, TILE), (unsigned)batch), dim3(TILE), 0, s,
                       w, x, frames, c, yi);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_enhance_stats(hipStream_t s, const float* yi, int batch, int frames, float* stats) {
    ProfScope prof(s, "enhance_stats_kernel", 16.0 * batch * frames, 32.0 * batch * frames);
    hipLaunchKernelGGL(enhance_stats_kernel, dim3((unsigned)batch), dim3(1024), 0, s, yi, frames, stats);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_head(hipStream_t s, const float* x, int batch, int frames, int c, const float* w, const float* b,
                float* audio, bool pretanh) {
    L3AC_REQUIRE(c % 4 == 0 && batch <= 65535, "head: bad shape");
    ProfScope prof(s, "head_kernel", 14.0 * c * batch * frames, 4.0 * (c + 1.0) * batch * frames);
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)ceil_div64(frames, TILE), (unsigned)batch), dim3(TILE), 0, s, x, frames, c,
                       w, b, audio, pretanh ? 1 : 0);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
