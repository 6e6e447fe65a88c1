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

#include <algorithm>
#include <cmath>

namespace {

template <int DH>
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                      const float* __restrict__ table, int frames, int heads,
                                                      int window, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * DH];
    __shared__ __attribute__((aligned(16))) float Vs[64 * DH];
    extern __shared__ float bias_s[];  // this head's distance table, entries [0, n_bias)
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
    const int w0 = i0 / window;
    const int jlo = w0 > 0 ? (w0 - 1) * window : 0;
    const int jhi = min(frames - 1, i0 + 63);
    // distances that can occur in this block: 0 .. (i0 + 63) - jlo  (< 2 * window); staged once, read per (query, key)
    const int n_bias = min(2 * window, jhi - jlo + 1);
    for (int d = lane; d < n_bias; d += 64) bias_s[d] = table[(int64_t)h * 2 * window + d];
    const float* tab = bias_s;

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

// ---------------------------------------------------------------------------------------------------------
// MFMA variant for dim_head == 32 (every shipped config): one wave = 32 queries of one (clip, head), a workgroup = 4
// consecutive query tiles sharing K / V^T chunks of 128 keys in LDS.
//   S^T[key][query] = K . Q^T   (v_mfma_f32_32x32x2_f32; keys on the accumulator's rows = registers, queries on lanes)
//   -> bias + causal/window mask + ONLINE softmax are lane-local (a query's keys sit in the 16 registers of its two
//      lanes; one cross-half shuffle per tile for the running max), and the probability tile is used directly as
//   the B operand of  O^T[d][query] += V^T[d][key] . P[key][query]  (reduction over the accumulator's row index).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
// KC: keys per LDS chunk = queries per workgroup (KC / 32 waves): 64 for clips of at most 64 frames (a 1 s chunk at 60 fps: half
// the staging of the 128-key form), 192 for 129-192 frames (one workgroup and ONE staging pass per (clip, head) instead of two
// workgroups and three), else 128.  Key tiles of 32 are visited in ascending order whatever KC is: the results do not depend on it.
constexpr int KS_STRIDE = 36;  // K rows: 32 + 4 floats (odd number of 16-B slots)

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int KC>
__global__ __launch_bounds__(2 * KC) void attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               const float* __restrict__ table, int frames, int heads,
                                                               int window, float scale) {
    constexpr int VT_STRIDE = KC + 4;
    constexpr int THREADS_A = 2 * KC;
    __shared__ __attribute__((aligned(16))) float Ks[KC * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Vt[32 * VT_STRIDE];
    extern __shared__ float bias_s[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * 32, ld = 3 * inner;
    const float* base = qkv + (int64_t)b * frames * ld + h * 32;
    const int i0 = blockIdx.x * KC;                  // first query of the workgroup
    const int q0 = i0 + 32 * wave;                   // first query of this wave
    const bool wave_active = q0 < frames;
    const int i = min(q0 + lj, frames - 1);          // this lane's query (clamped; inactive lanes never store)
    const bool q_ok = q0 + lj < frames;
    const int wi = i / window;
    const int jlo = (i0 / window) > 0 ? (i0 / window - 1) * window : 0;
    const int jhi = min(frames - 1, i0 + KC - 1);
    const int n_bias = min(2 * window, jhi - jlo + 1);
    for (int d = tid; d < n_bias; d += THREADS_A) bias_s[d] = table[(int64_t)h * 2 * window + d];

    float qv[16];  // Q[i][8q + 4 lh + r] * dh^-0.5
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)i * ld + 8 * q + 4 * lh);
        qv[4 * q] = v.x * scale; qv[4 * q + 1] = v.y * scale; qv[4 * q + 2] = v.z * scale; qv[4 * q + 3] = v.w * scale;
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    const int w_lo = (q0 / window) > 0 ? (q0 / window - 1) * window : 0;  // first key any query of this wave can see

    for (int c0 = jlo; c0 <= jhi; c0 += KC) {
        __syncthreads();
        for (int idx = tid; idx < KC * 8; idx += THREADS_A) {  // K rows and V^T of keys [c0, c0 + KC); zeros past jhi
            const int key = idx >> 3, d4 = (idx & 7) << 2;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (c0 + key <= jhi) {
                kv = *reinterpret_cast<const float4*>(base + (int64_t)(c0 + key) * ld + inner + d4);
                vv = *reinterpret_cast<const float4*>(base + (int64_t)(c0 + key) * ld + 2 * inner + d4);
            }
            *reinterpret_cast<float4*>(Ks + key * KS_STRIDE + d4) = kv;
            Vt[(d4 + 0) * VT_STRIDE + key] = vv.x;
            Vt[(d4 + 1) * VT_STRIDE + key] = vv.y;
            Vt[(d4 + 2) * VT_STRIDE + key] = vv.z;
            Vt[(d4 + 3) * VT_STRIDE + key] = vv.w;
        }
        __syncthreads();
        if (!wave_active) continue;
#pragma unroll 1
        for (int kt = 0; kt < KC / 32; ++kt) {
            const int key0 = c0 + 32 * kt;
            if (key0 > q0 + 31 || key0 > jhi || key0 + 31 < w_lo) continue;  // wave-uniform: nothing visible in this tile
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 kf = *reinterpret_cast<const float4*>(Ks + (32 * kt + lj) * KS_STRIDE + 8 * q + 4 * lh);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qv[4 * q], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qv[4 * q + 1], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qv[4 * q + 2], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qv[4 * q + 3], sacc, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + rowmap(r, lh);
                const bool vis = key <= i && key <= jhi && (key / window) >= wi - 1;
                const float sv = vis ? sacc[r] + bias_s[vis ? i - key : 0] : -INFINITY;
                sacc[r] = sv;
                mx = fmaxf(mx, sv);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const bool any = m_new > -INFINITY;
            const float alpha = any ? expf(m_run - m_new) : 1.f;  // exp(-inf) = 0 on the first visible tile
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = any ? expf(sacc[r] - m_new) : 0.f;  // masked entries: exp(-inf) = 0
                sacc[r] = pv;
                psum += pv;
                oacc[r] *= alpha;
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 vf = *reinterpret_cast<const float4*>(Vt + lj * VT_STRIDE + 32 * kt + 8 * g + 4 * lh);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, sacc[4 * g], oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, sacc[4 * g + 1], oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, sacc[4 * g + 2], oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, sacc[4 * g + 3], oacc, 0, 0, 0);
            }
        }
    }
    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q_ok) {
        const float inv = 1.0f / l_tot;
        float* dst = out + ((int64_t)b * frames + q0 + lj) * inner + h * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 8 * g + 4 * lh) =
                make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
    }
}

}  // namespace

int launch_attention(hipStream_t s, const float* qkv, float* out, const float* bias_table, int batch, int frames,
                     int heads, int dh, int window) {
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && frames > 0 && window > 0, "attention: bad shape");
    const dim3 grid((unsigned)ceil_div64(frames, 64), (unsigned)heads, (unsigned)batch);
    const float scale = (float)std::pow((double)dh, -0.5);
    const size_t bias_lds = (size_t)std::min(2 * window, frames) * sizeof(float);
    L3AC_REQUIRE(bias_lds <= 96 * 1024, "attention: window %d too large for the LDS bias table", window);
    const double pairs = frames <= window ? 0.5 * frames * (frames + 1.0) : 1.5 * (double)window * frames;
    ProfScope prof(s, "attention_kernel", 6.0 * dh * pairs * heads * batch, 4.0 * 4.0 * heads * dh * (double)batch * frames);
    if (dh == 32) {  // MFMA path
        const int kc = frames <= 64 ? 64 : (frames > 128 && frames <= 192 ? 192 : 128);
        const size_t lds = (size_t)std::min(2 * window, std::min(frames, kc + 2 * window)) * sizeof(float);
        const dim3 grid_m((unsigned)ceil_div64(frames, kc), (unsigned)heads, (unsigned)batch);
        if (kc == 64)
            hipLaunchKernelGGL(attention_mfma_kernel<64>, grid_m, dim3(128), lds, s, qkv, out, bias_table, frames, heads, window, scale);
        else if (kc == 192)
            hipLaunchKernelGGL(attention_mfma_kernel<192>, grid_m, dim3(384), lds, s, qkv, out, bias_table, frames, heads, window, scale);
        else
            hipLaunchKernelGGL(attention_mfma_kernel<128>, grid_m, dim3(256), lds, s, qkv, out, bias_table, frames, heads, window, scale);
        L3AC_LAUNCH_CHECK();
        return L3AC_OK;
    }
    switch (dh) {
#define L3AC_ATT_CASE(D) \
    case D: hipLaunchKernelGGL((attention_kernel<D>), grid, dim3(64), bias_lds, s, qkv, out, bias_table, frames, heads, window, scale); break
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
