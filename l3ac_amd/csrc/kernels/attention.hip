// Causal local attention with one window of look-back and an additive distance bias
// (PyPI local-attention==1.11.2 `LocalAttention.forward` as configured by reference l3ac/local_trans.py:34-38,
// with `DynamicPositionBias` from :30,:43; the package is an un-vendored dependency — see DESIGN.md).
//
// The package buckets the sequence into windows of W and lets window w attend to [w-1, w]; with
// exact_windowsize=False that is exactly:   query i sees key j  <=>  j <= i  and  floor(j/W) >= floor(i/W) - 1,
// score = (q_i * dh^-0.5) . k_j + table[h][i - j],  softmax over the visible keys, times v.
// For 1-second clips every shipped config has frames <= W: plain causal attention over <= 180 frames.
//
// One wave = 64 consecutive queries of one (clip, head); K/V chunks of 64 keys are staged in LDS and read as
// wave-wide broadcasts; scores are evaluated twice (max pass, then exp/sum pass) rather than rescaled online,
// to stay close to the reference's softmax.  <1 % of the path's FLOPs.
#include "../kernels.hpp"

#include <cmath>

namespace {

template <int DH>
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                      const float* __restrict__ table, int frames, int heads,
                                                      int window, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * DH];
    __shared__ __attribute__((aligned(16))) float Vs[64 * DH];
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 64;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int inner = heads * DH;
    const int ld = 3 * inner;
    const float* base = qkv + (int64_t)b * frames * ld + h * DH;
    const int i = i0 + lane;
    const bool active = i < frames;
    const int iq = active ? i : frames - 1;
    const int wi = iq / window;

    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)iq * ld + d);
        q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
    }
    const float* tab = table + (int64_t)h * 2 * window;
    const int w0 = i0 / window;
    const int jlo = w0 > 0 ? (w0 - 1) * window : 0;
    const int jhi = min(frames - 1, i0 + 63);

    auto stage = [&](int c0, bool with_v) {
        const int j = c0 + lane;
        if (j <= jhi) {
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                *reinterpret_cast<float4*>(Ks + lane * DH + d) =
                    *reinterpret_cast<const float4*>(base + (int64_t)j * ld + inner + d);
                if (with_v)
                    *reinterpret_cast<float4*>(Vs + lane * DH + d) =
                        *reinterpret_cast<const float4*>(base + (int64_t)j * ld + 2 * inner + d);
            }
        }
    };
    auto score = [&](int jj, int j, bool& visible) -> float {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(Ks + jj * DH + d);
            s = fmaf(q[d], kv.x, s);
            s = fmaf(q[d + 1], kv.y, s);
            s = fmaf(q[d + 2], kv.z, s);
            s = fmaf(q[d + 3], kv.w, s);
        }
        visible = active && j <= iq && (j / window) >= wi - 1;
        return s + tab[visible ? iq - j : 0];
    };

    // pass 1: row maximum
    float m = -INFINITY;
    for (int c0 = jlo; c0 <= jhi; c0 += 64) {
        __syncthreads();
        stage(c0, false);
        __syncthreads();
        const int cnt = min(64, jhi - c0 + 1);
        for (int jj = 0; jj < cnt; ++jj) {
            bool vis;
            const float s = score(jj, c0 + jj, vis);
            if (vis) m = fmaxf(m, s);
        }
    }
    // pass 2: exp, sum, weighted values
    float l = 0.f;
    float o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.f;
    for (int c0 = jlo; c0 <= jhi; c0 += 64) {
        __syncthreads();
        stage(c0, true);
        __syncthreads();
        const int cnt = min(64, jhi - c0 + 1);
        for (int jj = 0; jj < cnt; ++jj) {
            bool vis;
            const float s = score(jj, c0 + jj, vis);
            const float pz = vis ? expf(s - m) : 0.f;
            l += pz;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(Vs + jj * DH + d);
                o[d] = fmaf(pz, vv.x, o[d]);
                o[d + 1] = fmaf(pz, vv.y, o[d + 1]);
                o[d + 2] = fmaf(pz, vv.z, o[d + 2]);
                o[d + 3] = fmaf(pz, vv.w, o[d + 3]);
            }
        }
    }
    if (active) {
        const float inv = 1.0f / l;
        float* dst = out + ((int64_t)b * frames + i) * inner + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4)
            *reinterpret_cast<float4*>(dst + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    }
}

}  // namespace

int launch_attention(hipStream_t s, const float* qkv, float* out, const float* bias_table, int batch, int frames,
                     int heads, int dh, int window) {
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && frames > 0 && window > 0, "attention: bad shape");
    const dim3 grid((unsigned)ceil_div64(frames, 64), (unsigned)heads, (unsigned)batch);
    const float scale = (float)std::pow((double)dh, -0.5);
    const double pairs = frames <= window ? 0.5 * frames * (frames + 1.0) : 1.5 * (double)window * frames;
    ProfScope prof(s, "attention_kernel", 6.0 * dh * pairs * heads * batch, 4.0 * 4.0 * heads * dh * (double)batch * frames);
    switch (dh) {
#define L3AC_ATT_CASE(D) \
    case D: hipLaunchKernelGGL((attention_kernel<D>), grid, dim3(64), 0, s, qkv, out, bias_table, frames, heads, window, scale); break
        L3AC_ATT_CASE(4);
        L3AC_ATT_CASE(8);
        L3AC_ATT_CASE(16);
        L3AC_ATT_CASE(32);
        L3AC_ATT_CASE(64);
#undef L3AC_ATT_CASE
        default:
            l3ac_set_error("attention: dim_head=%d not in {4,8,16,32,64}", dh);
            return L3AC_EINVAL;
    }
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
