// Row kernels: every op whose unit of work is one frame (all channels of one time step), HBM-bound.
//
//   source                      reference
//   SRC_PLAIN                   identity (feeds the norms below)
//   SRC_DWCONV7                 ConvUnit.dw_conv: depth-wise Conv1d k7 pad 3        l3ac/modules.py:19-20
//   SRC_LERP                    nn.Upsample(scale, 'linear', align_corners=False)  l3ac/modules.py:162, local_trans.py:121
//   SRC_GATE                    EnhanceBlock: x + merge(InstanceNorm(yi)) * x      l3ac/tconv/__init__.py:35-44
//   norm
//   NORM_LN                     F.layer_norm over channels                         l3ac/layers.py:79-80, local_attention LayerNorm
//   NORM_CN                     channel_norm channels_first (divide by sqrt)       l3ac/layers.py:50-56
//
// Layout: a group of LPR lanes (power of two, <= 64) owns one row; lane j holds the 16-byte chunks j, j + LPR
// (so c <= 8 * LPR), loads are 16 B per lane and contiguous across the group; mean / variance are two-pass in
// registers with a butterfly over the group.
#include "../kernels.hpp"
#include "device_math.hpp"
#include "lane_sums.hpp"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_CH = 2;  // chunks per lane

// sum over the lpr consecutive lanes of a row's group (pos = lane index inside the group).  Power-of-two groups use
// the xor butterfly; other sizes (6 / 12 lanes for 24 / 48 / 96 channels: five or ten rows per wave instead of padding
// every group to 8 / 16 lanes) a segmented shift-down reduction followed by a broadcast from the group's first lane.
__device__ __forceinline__ float group_sum(float v, int lpr, int pos) {
    if ((lpr & (lpr - 1)) == 0) return lanes_sum(v, lpr);  // (DPP / permlane adds instead of ds_bpermute round trips: lane_sums.hpp)
    for (int d = 1; d < lpr; d <<= 1) {
        const float o = __shfl_down(v, d, 64);
        v += pos + d < lpr ? o : 0.f;
    }
    return __shfl(v, (int)(threadIdx.x & 63) - pos, 64);
}

__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// IDX: the integer type of row numbers and element offsets.  The launcher picks 32 bits whenever every tensor has fewer than 2^31
// elements: with int64 the row / frames division and the address products are ~150 of the kernel's instructions per 16 bytes,
// and the kernel ran at the speed of its integer arithmetic (3.2 TB/s), not of HBM.
// LPR: the lane-group size as a compile-time constant for the sizes the network uses (0 = the run-time value): it folds the lane
// decomposition and the group reductions' loops.
// NCH: 16-byte chunks per lane (1 when c <= 4 lpr, else MAX_CH).
template <int SRC, int NORM, class IDX, int LPR, int NCH>
__global__ __launch_bounds__(THREADS) void row_kernel(const RowArgs p, const int lpr_rt) {
    const int lpr = LPR ? LPR : lpr_rt;
    // lane groups never straddle a wave: 64 / lpr groups per wave, the remaining lanes (4 of 64 for lpr = 6 / 12) idle
    const int gpw = 64 / lpr;
    const int lane = threadIdx.x & 63;
    const int grp = lane / lpr;
    const int j = lane - grp * lpr;
    const IDX rows = (IDX)(p.batch * p.frames_out);
    const IDX frames_out = (IDX)p.frames_out, frames_in = (IDX)p.frames_in, cc = (IDX)p.c;
    const int nchunk = p.c >> 2;
    const IDX wave_id = (IDX)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    const IDX row_step = (IDX)gridDim.x * (THREADS / 64) * gpw;
    // grid-stride over rows: a block handles many (a launch of one 96-B row per lane group is dispatch-bound); whole lane
    // groups enter and leave the loop together, so the group shuffles stay inside active lanes
    for (IDX row = grp < gpw ? wave_id * gpw + grp : rows; row < rows; row += row_step) {
    const bool row_ok = true;
    const IDX b = row / frames_out;
    const IDX t = row - b * frames_out;

    float4 v[NCH];
    bool ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int chunk = j + i * lpr;
        ok[i] = row_ok && chunk < nchunk;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!ok[i]) continue;
        const int c0 = chunk << 2;
        if (SRC == SRC_PLAIN) {
            v[i] = *reinterpret_cast<const float4*>(p.x + (size_t)((b * frames_in + t) * cc + (IDX)c0));
        } else if (SRC == SRC_DWCONV7) {
            float4 acc = *reinterpret_cast<const float4*>(p.dw_b + c0);
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const int64_t ts = (int64_t)t + tap - 3;
                if (ts >= 0 && ts < p.frames_in) {
                    const float4 xv = *reinterpret_cast<const float4*>(p.x + (size_t)((b * frames_in + (IDX)ts) * cc + (IDX)c0));
                    const float4 wv = *reinterpret_cast<const float4*>(p.dw_w + tap * p.c + c0);
                    acc = f4_fma(xv, wv, acc);
                }
            }
            v[i] = acc;
        } else if (SRC == SRC_LERP) {
            // ATen upsample_linear1d (align_corners=False): src = scale * (dst + 0.5) - 0.5 clamped at 0
            const float rscale = (float)(1.0 / (double)p.scale);
            float src = __fsub_rn(__fmul_rn(rscale, (float)t + 0.5f), 0.5f);
            src = src < 0.f ? 0.f : src;
            const IDX i0 = (IDX)src;
            const IDX i1 = i0 + (i0 + 1 < frames_in ? 1 : 0);
            float l1 = src - (float)i0;
            l1 = fminf(fmaxf(l1, 0.f), 1.f);
            const float l0 = 1.f - l1;
            const float4 x0 = *reinterpret_cast<const float4*>(p.x + (size_t)((b * frames_in + i0) * cc + (IDX)c0));
            const float4 x1 = *reinterpret_cast<const float4*>(p.x + (size_t)((b * frames_in + i1) * cc + (IDX)c0));
            v[i] = make_float4(__fadd_rn(__fmul_rn(l0, x0.x), __fmul_rn(l1, x1.x)),
                               __fadd_rn(__fmul_rn(l0, x0.y), __fmul_rn(l1, x1.y)),
                               __fadd_rn(__fmul_rn(l0, x0.z), __fmul_rn(l1, x1.z)),
                               __fadd_rn(__fmul_rn(l0, x0.w), __fmul_rn(l1, x1.w)));
        } else {  // SRC_GATE
            const float4 yraw = *reinterpret_cast<const float4*>(p.yi + (size_t)((b * frames_in + t) * 4));
            const float4 mean = *reinterpret_cast<const float4*>(p.stats + (size_t)(b * 8));
            const float4 istd = *reinterpret_cast<const float4*>(p.stats + (size_t)(b * 8 + 4));
            const float4 iw = *reinterpret_cast<const float4*>(p.in_w);
            const float4 ib = *reinterpret_cast<const float4*>(p.in_b);
            const float y0 = (yraw.x - mean.x) * istd.x * iw.x + ib.x;
            const float y1 = (yraw.y - mean.y) * istd.y * iw.y + ib.y;
            const float y2 = (yraw.z - mean.z) * istd.z * iw.z + ib.z;
            const float y3 = (yraw.w - mean.w) * istd.w * iw.w + ib.w;
            const float4 xv = *reinterpret_cast<const float4*>(p.x + (size_t)((b * frames_in + t) * cc + (IDX)c0));
            const float4 gb = *reinterpret_cast<const float4*>(p.gate_b + c0);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            const float gbs[4] = {gb.x, gb.y, gb.z, gb.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 gw = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)(c0 + e) * 4);
                const float g = gbs[e] + gw.x * y0 + gw.y * y1 + gw.z * y2 + gw.w * y3;
                o[e] = xs[e] + g * xs[e];
            }
            v[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }

    if (NORM != NORM_NONE) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);  // masked chunks hold zeros
        const float mean = group_sum(s, lpr, j) / (float)p.c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (ok[i]) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float var = group_sum(q, lpr, j) / (float)p.c;
        // LN: (x - mu) * rstd (F.layer_norm).  CN: the reference divides by sqrt(var + eps) (layers.py:54); one reciprocal
        // per row + a multiply per element differs from the per-element division by at most ~1 ulp
        const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (!ok[i]) continue;
            const int c0 = (j + i * lpr) << 2;
            const float4 w = *reinterpret_cast<const float4*>(p.nw + c0);
            const float4 bb = *reinterpret_cast<const float4*>(p.nb + c0);
            if (NORM == NORM_LN) {
                v[i] = make_float4((v[i].x - mean) * rstd * w.x + bb.x, (v[i].y - mean) * rstd * w.y + bb.y,
                                   (v[i].z - mean) * rstd * w.z + bb.z, (v[i].w - mean) * rstd * w.w + bb.w);
            } else {
                v[i] = make_float4(w.x * ((v[i].x - mean) * rstd) + bb.x, w.y * ((v[i].y - mean) * rstd) + bb.y,
                                   w.z * ((v[i].z - mean) * rstd) + bb.z, w.w * ((v[i].w - mean) * rstd) + bb.w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (ok[i]) *reinterpret_cast<float4*>(p.y + (size_t)(row * cc + (IDX)((j + i * lpr) << 2))) = v[i];
    }
    }  // row loop
}

// ---- depth-wise conv k7 + LayerNorm with a rolling register window ------------------------------------------
// The generic row kernel re-reads each input row seven times (L1/L2-served).  Here a lane group walks SEG consecutive
// frames of one clip: the 7-row window of its channels, the 7 tap weights and the LayerNorm affine stay in registers,
// each step loads ONE new row (16 B per lane, contiguous across the group).  Window slot of relative row r is r mod 7,
// so inside a 7-way unrolled step every index is a compile-time constant.
// SEG (a multiple of 7, chosen at launch) trades halo re-reads (6 / SEG, L2-served) against the number of independent
// lane groups: a group's steps are serial, so short clips need short segments to keep every SIMD supplied with waves.

template <int CH>
__global__ __launch_bounds__(THREADS) void dwconv_ln_kernel(const RowArgs p, const int lpr, const int DW_SEG) {
    const int groups_per_block = THREADS / lpr;
    const int64_t seg_per_clip = (p.frames_out + DW_SEG - 1) / DW_SEG;
    const int64_t seg = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / lpr;
    const int j = threadIdx.x % lpr;
    const bool seg_ok = seg < p.batch * seg_per_clip;
    const int64_t sg = seg_ok ? seg : 0;
    const int64_t b = sg / seg_per_clip;
    const int t0 = (int)(sg % seg_per_clip) * DW_SEG;
    const int frames = (int)p.frames_in;
    const int nchunk = p.c >> 2;
    const float* clip = p.x + b * p.frames_in * p.c;
    float* dst = p.y + b * p.frames_in * p.c;

    bool ok[CH];
    int c0[CH];
    float4 wt[7][CH], bias[CH], lw[CH], lb[CH], win[7][CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int chunk = j + i * lpr;
        ok[i] = seg_ok && chunk < nchunk;
        c0[i] = (ok[i] ? chunk : 0) << 2;
        bias[i] = *reinterpret_cast<const float4*>(p.dw_b + c0[i]);
        lw[i] = *reinterpret_cast<const float4*>(p.nw + c0[i]);
        lb[i] = *reinterpret_cast<const float4*>(p.nb + c0[i]);
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) wt[tap][i] = *reinterpret_cast<const float4*>(p.dw_w + tap * p.c + c0[i]);
    }
    auto load_row = [&](int t, int i) -> float4 {
        if (ok[i] && t >= 0 && t < frames) return *reinterpret_cast<const float4*>(clip + (int64_t)t * p.c + c0[i]);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // rows t0-3 .. t0+2 -> slots 4,5,6,0,1,2
#pragma unroll
    for (int d = -3; d <= 2; ++d)
#pragma unroll
        for (int i = 0; i < CH; ++i) win[(d + 7) % 7][i] = load_row(t0 + d, i);

    for (int base = 0; base < DW_SEG; base += 7) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int t = t0 + base + u;
#pragma unroll
            for (int i = 0; i < CH; ++i) win[(u + 3) % 7][i] = load_row(t + 3, i);  // overwrites row t-4
            float4 v[CH];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                float4 acc = bias[i];
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) acc = f4_fma(win[(u + tap + 4) % 7][i], wt[tap][i], acc);  // row t+tap-3
                v[i] = ok[i] ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
            const float mean = group_sum(s, lpr, j) / (float)p.c;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (ok[i]) {
                    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
            const float rstd = 1.0f / sqrtf(group_sum(q, lpr, j) / (float)p.c + p.eps);
            if (t < frames) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (!ok[i]) continue;
                    *reinterpret_cast<float4*>(dst + (int64_t)t * p.c + c0[i]) =
                        make_float4((v[i].x - mean) * rstd * lw[i].x + lb[i].x, (v[i].y - mean) * rstd * lw[i].y + lb[i].y,
                                    (v[i].z - mean) * rstd * lw[i].z + lb[i].z, (v[i].w - mean) * rstd * lw[i].w + lb[i].w);
                }
            }
        }
    }
}

int launch_dwconv_ln(hipStream_t s, const RowArgs& r, int lpr) {
    // aim at ~6 waves per SIMD worth of lane groups
    const int64_t rows_per_wave_group = (r.batch * r.frames_out * lpr / 64) / (256 * 4 * 6);
    int DW_SEG = (int)(rows_per_wave_group / 7) * 7;
    DW_SEG = DW_SEG < 7 ? 7 : (DW_SEG > 56 ? 56 : DW_SEG);
    const int64_t segs = r.batch * ((r.frames_out + DW_SEG - 1) / DW_SEG);
    const int64_t blocks = ceil_div64(segs, THREADS / lpr);
    if (blocks <= 0) return L3AC_OK;
    L3AC_REQUIRE(blocks < (int64_t)1 << 31, "rows: grid too large");
    const double elems = (double)r.batch * r.frames_out * r.c;
    ProfScope prof(s, "dwconv_ln_kernel", 22.0 * elems, 8.0 * elems);
    if (r.c <= 4 * lpr)
        hipLaunchKernelGGL((dwconv_ln_kernel<1>), dim3((unsigned)blocks), dim3(THREADS), 0, s, r, lpr, DW_SEG);
    else
        hipLaunchKernelGGL((dwconv_ln_kernel<2>), dim3((unsigned)blocks), dim3(THREADS), 0, s, r, lpr, DW_SEG);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

template <int SRC, int NORM>
int launch_rows_t(hipStream_t s, const RowArgs& r, int lpr) {
    const int64_t rows = r.batch * r.frames_out;
    int64_t blocks = ceil_div64(rows, (THREADS / 64) * (64 / lpr));
    if (blocks <= 0) return L3AC_OK;
    // the kernel strides over rows; measured: the upsample (many short output rows per input row, 2.3 -> 2.7 TB/s) gains from
    // fewer, longer-lived blocks, the 1:1 variants do not
    if (SRC == SRC_LERP && blocks > 256 * 32) blocks = 256 * 32;
    static const char* const names[4][3] = {{"row_kernel<PLAIN,NONE>", "row_kernel<PLAIN,LN>", "row_kernel<PLAIN,CN>"},
                                            {"row_kernel<DWCONV7,NONE>", "row_kernel<DWCONV7,LN>", "row_kernel<DWCONV7,CN>"},
                                            {"row_kernel<LERP,NONE>", "row_kernel<LERP,LN>", "row_kernel<LERP,CN>"},
                                            {"row_kernel<GATE,NONE>", "row_kernel<GATE,LN>", "row_kernel<GATE,CN>"}};
    const double in_elems = (double)r.batch * r.frames_in * r.c, out_elems = (double)rows * r.c;
    ProfScope prof(s, names[SRC][NORM], (SRC == SRC_DWCONV7 ? 14.0 : 2.0) * out_elems + (NORM != NORM_NONE ? 8.0 * out_elems : 0.0),
                   4.0 * (in_elems + out_elems));
    const bool small = (double)r.batch * r.frames_in * r.c < 2147483648.0 && out_elems < 2147483648.0 && rows + blocks * 256 < ((int64_t)1 << 31);
#define L3AC_ROWS_LAUNCH(IDX_, LPR_)                                                                                             \
    do {                                                                                                                         \
        if (r.c <= 4 * lpr)                                                                                                      \
            hipLaunchKernelGGL((row_kernel<SRC, NORM, IDX_, LPR_, 1>), dim3((unsigned)blocks), dim3(THREADS), 0, s, r, lpr);     \
        else                                                                                                                     \
            hipLaunchKernelGGL((row_kernel<SRC, NORM, IDX_, LPR_, MAX_CH>), dim3((unsigned)blocks), dim3(THREADS), 0, s, r, lpr); \
    } while (0)
    if (!small) {
        L3AC_ROWS_LAUNCH(int64_t, 0);
    } else {
        switch (lpr) {
            case 6: L3AC_ROWS_LAUNCH(unsigned, 6); break;     // 24 channels
            case 12: L3AC_ROWS_LAUNCH(unsigned, 12); break;   // 48 / 96
            case 32: L3AC_ROWS_LAUNCH(unsigned, 32); break;   // 128
            case 48: L3AC_ROWS_LAUNCH(unsigned, 48); break;   // 192
            case 64: L3AC_ROWS_LAUNCH(unsigned, 64); break;   // 256 / 512
            default: L3AC_ROWS_LAUNCH(unsigned, 0); break;
        }
    }
#undef L3AC_ROWS_LAUNCH
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

// ---- elementwise -------------------------------------------------------------------------------
// mode bit 0: evaluate with the two-elements-per-lane form (snake_act2 / sin_squared2) the GEMM epilogues and the fused
// units use; bit 1: y = sin(x)^2 alone (alpha unused) instead of snake(x); mode 4: y = gelu(x), the kernels' exact-GELU (gelu_erf)
__global__ __launch_bounds__(THREADS) void snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int64_t n4, int c4, const float* __restrict__ alpha,
                                                       const float* __restrict__ inv_alpha, const int mode) {
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * THREADS) {
        const int c0 = (int)(i % c4) << 2;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        if (mode == 4) {
            o = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
        } else if (mode & 2) {
            if (mode & 1) {
                const f32x2 lo = sin_squared2(f32x2{v.x, v.y}), hi = sin_squared2(f32x2{v.z, v.w});
                o = make_float4(lo.x, lo.y, hi.x, hi.y);
            } else {
                o = make_float4(sin_squared(v.x), sin_squared(v.y), sin_squared(v.z), sin_squared(v.w));
            }
        } else {
            const float4 a = *reinterpret_cast<const float4*>(alpha + c0);
            const float4 ia = *reinterpret_cast<const float4*>(inv_alpha + c0);
            if (mode & 1) {
                const f32x2 lo = snake_act2(f32x2{v.x, v.y}, f32x2{a.x, a.y}, f32x2{ia.x, ia.y});
                const f32x2 hi = snake_act2(f32x2{v.z, v.w}, f32x2{a.z, a.w}, f32x2{ia.z, ia.w});
                o = make_float4(lo.x, lo.y, hi.x, hi.y);
            } else {
                o = make_float4(snake_act(v.x, a.x, ia.x), snake_act(v.y, a.y, ia.y), snake_act(v.z, a.z, ia.z),
                                snake_act(v.w, a.w, ia.w));
            }
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

__global__ __launch_bounds__(THREADS) void geglu_kernel(const float* __restrict__ h, int64_t ldh, float* __restrict__ y,
                                                       int64_t ldy, int64_t rows, int inner) {
    const int64_t total = rows * ldy;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t r = i / ldy;
        const int j = (int)(i % ldy);
        float o = 0.f;
        if (j < inner) o = h[r * ldh + j] * gelu_erf(h[r * ldh + inner + j]);
        y[i] = o;
    }
}

// per-clip sum of squares (GRN exact mode, layers.py:113): one block per (clip, slice), atomics into sumsq[batch]
__global__ __launch_bounds__(THREADS) void grn_sumsq_kernel(const float* __restrict__ h, int64_t per_clip4,
                                                           float* __restrict__ sumsq) {
    const int64_t b = blockIdx.y;
    const float4* base = reinterpret_cast<const float4*>(h) + b * per_clip4;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < per_clip4; i += (int64_t)gridDim.x * THREADS) {
        const float4 v = base[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int mask = 32; mask > 0; mask >>= 1) acc += __shfl_xor(acc, mask, 64);
    __shared__ float part[THREADS / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sumsq + b, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ __launch_bounds__(THREADS) void grn_apply_kernel(float* __restrict__ h, int64_t per_clip4, int c4,
                                                           const float* __restrict__ sumsq,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int* __restrict__ min_track) {
    const int64_t b = blockIdx.y;
    // guard of the fast path (which takes the normaliser as exactly 1.0f, true for ||x|| >= 0.25): the smallest sum of squares
    // seen by this context (non-negative floats order like their bit patterns)
    if (min_track && blockIdx.x == 0 && threadIdx.x == 0) atomicMin(min_track, __float_as_int(sumsq[b]));
    const float g = sqrtf(sumsq[b]);
    const float nx = g / (g + 1e-8f);  // layers.py:114 (the mean over a size-1 dim is the value itself)
    float4* base = reinterpret_cast<float4*>(h) + b * per_clip4;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < per_clip4; i += (int64_t)gridDim.x * THREADS) {
        const int c0 = (int)(i % c4) << 2;
        const float4 v = base[i];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
        const float4 be = *reinterpret_cast<const float4*>(beta + c0);
        base[i] = make_float4((ga.x * (v.x * nx) + be.x) + v.x, (ga.y * (v.y * nx) + be.y) + v.y,
                              (ga.z * (v.z * nx) + be.z) + v.z, (ga.w * (v.w * nx) + be.w) + v.w);
    }
}

// EnhanceBlock gate as a flat elementwise pass (no row reduction is needed): thread i owns 16 B of the tensor, so every lane
// is busy whatever the channel count (the row kernel idles 25 % of its lanes at c = 48 / 96); same arithmetic as SRC_GATE.
__global__ __launch_bounds__(THREADS) void gate_flat_kernel(const RowArgs p, const unsigned n4, const unsigned nchunk) {
    const float4 iw = *reinterpret_cast<const float4*>(p.in_w);
    const float4 ib = *reinterpret_cast<const float4*>(p.in_b);
    for (unsigned i = blockIdx.x * THREADS + threadIdx.x; i < n4; i += gridDim.x * THREADS) {
        const unsigned row = i / nchunk;
        const int c0 = (int)(i - row * nchunk) << 2;
        const unsigned b = row / (unsigned)p.frames_in;
        const float4 yraw = *reinterpret_cast<const float4*>(p.yi + (int64_t)row * 4);
        const float4 mean = *reinterpret_cast<const float4*>(p.stats + b * 8);
        const float4 istd = *reinterpret_cast<const float4*>(p.stats + b * 8 + 4);
        const float y0 = (yraw.x - mean.x) * istd.x * iw.x + ib.x;
        const float y1 = (yraw.y - mean.y) * istd.y * iw.y + ib.y;
        const float y2 = (yraw.z - mean.z) * istd.z * iw.z + ib.z;
        const float y3 = (yraw.w - mean.w) * istd.w * iw.w + ib.w;
        const float4 xv = reinterpret_cast<const float4*>(p.x)[i];
        const float4 gb = *reinterpret_cast<const float4*>(p.gate_b + c0);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float gbs[4] = {gb.x, gb.y, gb.z, gb.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 gw = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)(c0 + e) * 4);
            const float g = gbs[e] + gw.x * y0 + gw.y * y1 + gw.z * y2 + gw.w * y3;
            o[e] = xs[e] + g * xs[e];
        }
        reinterpret_cast<float4*>(p.y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

inline unsigned stream_grid(int64_t work_items) {
    const int64_t blocks = ceil_div64(work_items, THREADS);
    return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks));
}

}  // namespace

int launch_rows(hipStream_t s, const RowArgs& r) {
    L3AC_REQUIRE(r.x && r.y && r.c > 0 && r.c % 4 == 0, "rows: bad arguments (c=%d)", r.c);
    // lanes per row: the 16-B chunks of a row, two per lane once that keeps more lanes of a wave busy (24 chunks: 12 lanes
    // x 5 rows instead of 24 x 2) or when the row is wider than a wave
    const int nchunk = r.c / 4;
    int lpr = nchunk <= 64 ? nchunk : 64;
    if (nchunk > 16 && nchunk % 2 == 0 && (64 / (nchunk / 2)) * (nchunk / 2) > (64 / lpr) * lpr) lpr = nchunk / 2;
    while (lpr * MAX_CH < nchunk) ++lpr;  // (very wide rows)
    L3AC_REQUIRE(lpr <= 64 && r.c <= 4 * lpr * MAX_CH, "rows: c=%d too wide for the row kernel", r.c);
    const int lpr_pow2 = [&] { int v = 1; while (v * 4 < r.c && v < 64) v <<= 1; return v; }();  // dwconv_ln_kernel's grouping
    if (r.norm != NORM_NONE) L3AC_REQUIRE(r.nw && r.nb, "rows: norm without affine parameters");
#define L3AC_ROWS_CASE(S, N) \
    if (r.src == S && r.norm == N) return launch_rows_t<S, N>(s, r, lpr)
    L3AC_ROWS_CASE(SRC_PLAIN, NORM_LN);
    L3AC_ROWS_CASE(SRC_PLAIN, NORM_CN);
    if (r.src == SRC_DWCONV7 && r.norm == NORM_LN && r.frames_in == r.frames_out) return launch_dwconv_ln(s, r, lpr_pow2);
    L3AC_ROWS_CASE(SRC_DWCONV7, NORM_LN);
    L3AC_ROWS_CASE(SRC_LERP, NORM_NONE);
    L3AC_ROWS_CASE(SRC_LERP, NORM_CN);
    if (r.src == SRC_GATE && r.norm == NORM_NONE && r.frames_in == r.frames_out && r.batch * r.frames_out * (r.c / 4) < (1ll << 31)) {
        const int64_t n4 = r.batch * r.frames_out * (r.c / 4);
        if (n4 == 0) return L3AC_OK;
        ProfScope prof(s, "gate_flat_kernel", 10.0 * n4 * 4, 32.0 * n4);
        const int64_t blocks = ceil_div64(n4, THREADS);
        hipLaunchKernelGGL(gate_flat_kernel, dim3((unsigned)(blocks > 256 * 64 ? 256 * 64 : blocks)), dim3(THREADS), 0, s, r, (unsigned)n4,
                           (unsigned)(r.c / 4));
        L3AC_LAUNCH_CHECK();
        return L3AC_OK;
    }
    L3AC_ROWS_CASE(SRC_GATE, NORM_NONE);
#undef L3AC_ROWS_CASE
    l3ac_set_error("rows: unsupported (src=%d, norm=%d)", r.src, r.norm);
    return L3AC_EINVAL;
}

int launch_snake(hipStream_t s, const float* x, float* y, int64_t rows, int c, const float* alpha,
                 const float* inv_alpha, int mode) {
    L3AC_REQUIRE(c % 4 == 0, "snake: c=%d must be a multiple of 4", c);
    const int64_t n4 = rows * (c / 4);
    if (n4 == 0) return L3AC_OK;
    ProfScope prof(s, "snake_kernel", 16.0 * n4, 32.0 * n4);
    hipLaunchKernelGGL(snake_kernel, dim3(stream_grid(n4)), dim3(THREADS), 0, s, x, y, n4, c / 4, alpha, inv_alpha, mode);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_geglu(hipStream_t s, const float* h, int64_t ldh, float* y, int64_t ldy, int64_t rows, int inner) {
    if (rows == 0) return L3AC_OK;
    ProfScope prof(s, "geglu_kernel", 10.0 * rows * inner, 12.0 * rows * inner);
    hipLaunchKernelGGL(geglu_kernel, dim3(stream_grid(rows * ldy)), dim3(THREADS), 0, s, h, ldh, y, ldy, rows, inner);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_grn_sumsq(hipStream_t s, const float* h, int64_t batch, int64_t per_clip, float* sumsq) {
    L3AC_REQUIRE(per_clip % 4 == 0, "grn: per-clip size must be a multiple of 4");
    L3AC_HIP_CHECK(hipMemsetAsync(sumsq, 0, (size_t)batch * sizeof(float), s));
    const unsigned gx = (unsigned)(ceil_div64(per_clip / 4, THREADS * 8) < 1 ? 1 : ceil_div64(per_clip / 4, THREADS * 8));
    ProfScope prof(s, "grn_sumsq_kernel", 2.0 * batch * per_clip, 4.0 * batch * per_clip);
    hipLaunchKernelGGL(grn_sumsq_kernel, dim3(gx, (unsigned)batch), dim3(THREADS), 0, s, h, per_clip / 4, sumsq);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_grn_apply(hipStream_t s, float* h, int64_t batch, int64_t frames, int c, const float* sumsq,
                     const float* gamma, const float* beta, float* min_track) {
    const int64_t per_clip4 = frames * c / 4;
    const unsigned gx = (unsigned)(ceil_div64(per_clip4, THREADS * 8) < 1 ? 1 : ceil_div64(per_clip4, THREADS * 8));
    ProfScope prof(s, "grn_apply_kernel", 4.0 * batch * frames * c, 8.0 * batch * frames * c);
    hipLaunchKernelGGL(grn_apply_kernel, dim3(gx, (unsigned)batch), dim3(THREADS), 0, s, h, per_clip4, c / 4, sumsq, gamma,
                       beta, reinterpret_cast<int*>(min_track));
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
