// fp32 MFMA GEMM for every channel contraction of the L3AC conv stacks and transformer blocks
// (reference: nn.Linear / nn.Conv1d calls in l3ac/modules.py:19-36,96-99,110,150,161 and local_attention's
// to_qkv / to_out / FeedForward linears).
//
//   c[m][n] = epilogue( sum_k A(m, k) * w[n][k] )
//
// gfx950 design
//   * v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD = the fp32 peak.
//   * block = 4 waves, tile 128 (m) x 32*NT (n) x 32 (k); wave w owns rows [32w, 32w+32) across all NT column
//     tiles, so a whole output row lives in one wave (row-wise epilogues need no cross-wave traffic).
//   * both operands are k-contiguous in HBM; tiles are staged global -> registers -> LDS (16 B per lane,
//     full 128-B lines per 8 lanes) and double-buffered, the next tile's global loads in flight during the MFMAs.
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row >> 1) & 7 so that the ds_read_b128
//     fragment reads (one row per lane, same k chunk) are bank-conflict free.
//   * one ds_read_b128 per operand feeds FOUR MFMAs: lane (i, h) holds k = 8q + 4h + {0..3}; MFMA r consumes
//     element r from A and B alike, i.e. the k order inside a group of 8 is permuted identically on both sides.
//   * A can be an implicit 1-D convolution (taps > 1): row m = (clip b, frame t) gathers frames
//     t + (tap - taps/2) * dil of the same clip, zero outside it — no im2col buffer.
#include "../kernels.hpp"
#include "device_math.hpp"
#include "gemm_epilogue.hpp"

#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BM = 128;
constexpr int THREADS = 256;

// float index of 16-B chunk `chunk` of row `row` in a [rows][BK] tile; the XOR spreads the 16 rows a
// ds_read_b128 lane group touches over all 16 slots of the 256-B bank row
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return BK == 32 ? row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2) : row * 16 + ((chunk ^ ((row >> 2) & 3)) << 2);
}

// FULLK: k is a multiple of BK and A is plain rows -> the staging loads are unconditional (rows / columns past the
// edge are clamped to row 0: they only feed accumulators that are never stored), no exec-mask branches in the loop.
// GATED: the EnhanceBlock gate is applied to A on its way to LDS (GemmArgs::gate_*), saving that tensor's own pass.
template <int NT, bool CONV, int BK, bool FULLK, bool GATED = false>
__global__ __launch_bounds__(THREADS, BK == 16 ? 3 : 2) void gemm_f32_kernel(const GemmArgs p) {
    static_assert(!GATED || (FULLK && !CONV), "the gated A operand is implemented for plain full-k tiles");
    constexpr int BN = 32 * NT;
    constexpr int CPR = BK / 4;            // 16-B chunks per tile row
    constexpr int RP = THREADS / CPR;      // tile rows staged per pass
    constexpr int AP = BM / RP;            // passes for the A tile
    constexpr int WP = (BN + RP - 1) / RP; // passes for the W tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM * BK]
    float* Ws = smem + 2 * BM * BK;   // [2][BN * BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD), each with
    // its own L2.  Within a group of 8 row panels, panel = blockIdx % 8 and the column tile advances every 8 blocks,
    // so all column tiles of one A row panel run on ONE XCD and the panel is fetched from HBM once, not 8 times.
    const int n_blocks = (p.n + BN - 1) / BN;
    const int64_t m_panels = (p.m + BM - 1) / BM;
    // (32-bit on purpose: the 64-bit forms of these four wave-uniform divisions are ~100 scalar instructions each, a visible part of
    // a short-K block's life; the launcher keeps the grid, hence every quotient, below 2^31)
    const unsigned group = blockIdx.x / (unsigned)(8 * n_blocks);
    const unsigned in_group = blockIdx.x % (unsigned)(8 * n_blocks);
    const unsigned panels_here = ((int64_t)group * 8 + 8 <= m_panels) ? (unsigned)8 : (unsigned)(m_panels - (int64_t)group * 8);
    const int64_t m0 = ((int64_t)group * 8 + in_group % panels_here) * BM;
    const int n0 = (int)(in_group / panels_here) * BN;

    // ---- staging roles: thread -> (chunk column cc, rows r0 + RP i) ---------------------------------
    const int cc = tid % CPR;
    const int r0 = tid / CPR;
    const float* a_row[AP];
    int a_t[AP];
    bool a_ok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int64_t m = m0 + r0 + RP * i;
        a_ok[i] = m < p.m;
        const int64_t mm = a_ok[i] ? m : 0;
        a_t[i] = CONV ? (int)((unsigned)mm % (unsigned)p.frames) : 0;  // (m < 2^31: checked by the launcher)
        a_row[i] = p.a + mm * p.lda;
    }
    const float* w_row[WP];
    bool w_ok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int nl = r0 + RP * i;
        const int n = n0 + nl;
        w_ok[i] = nl < BN && n < p.n;
        w_row[i] = p.w + (int64_t)(w_ok[i] ? n : 0) * p.ldw;
    }
    // conv bookkeeping for this thread's chunk: k = k_tile + 4 cc = tap * cin + c
    int tap = 0, ch = 4 * cc;
    if (CONV) {
        while (ch >= p.cin) { ch -= p.cin; ++tap; }
    }
    const int half = p.taps >> 1;

    float4 yn[GATED ? AP : 1];  // InstanceNorm'ed branch signals of this thread's rows
    if constexpr (GATED) {
        const float4 iw = *reinterpret_cast<const float4*>(p.gate_in_w);
        const float4 ib = *reinterpret_cast<const float4*>(p.gate_in_b);
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int64_t m = m0 + r0 + RP * i;
            const int64_t mm = m < p.m ? m : 0;
            const float4 yraw = *reinterpret_cast<const float4*>(p.gate_yi + mm * 4);
            const float* st = p.gate_stats + (size_t)((unsigned)mm / (unsigned)p.gate_frames) * 8;
            const float4 mean = *reinterpret_cast<const float4*>(st);
            const float4 istd = *reinterpret_cast<const float4*>(st + 4);
            yn[i] = make_float4((yraw.x - mean.x) * istd.x * iw.x + ib.x, (yraw.y - mean.y) * istd.y * iw.y + ib.y,
                                (yraw.z - mean.z) * istd.z * iw.z + ib.z, (yraw.w - mean.w) * istd.w * iw.w + ib.w);
        }
    }
    float4 a_reg[AP], w_reg[WP];
    auto load_tile = [&](int k_tile) {
        const int k = k_tile + 4 * cc;
        const bool k_ok = k < p.k;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CONV) {
                const int ts = a_t[i] + (tap - half) * p.dil;
                if (a_ok[i] && k_ok && ts >= 0 && ts < p.frames)
                    v = *reinterpret_cast<const float4*>(a_row[i] + (int64_t)(tap - half) * p.dil * p.lda + ch);
            } else if (FULLK) {
                v = *reinterpret_cast<const float4*>(a_row[i] + k);
            } else {
                if (a_ok[i] && k_ok) v = *reinterpret_cast<const float4*>(a_row[i] + k);
            }
            a_reg[i] = v;
        }
        if constexpr (GATED) {
            const float4 gb = *reinterpret_cast<const float4*>(p.gate_b + k);
            const float4 g0 = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)k * 4);
            const float4 g1 = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)k * 4 + 4);
            const float4 g2 = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)k * 4 + 8);
            const float4 g3 = *reinterpret_cast<const float4*>(p.gate_w + (int64_t)k * 4 + 12);
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const float4 y = yn[i];
                const float e0 = gb.x + g0.x * y.x + g0.y * y.y + g0.z * y.z + g0.w * y.w;
                const float e1 = gb.y + g1.x * y.x + g1.y * y.y + g1.z * y.z + g1.w * y.w;
                const float e2 = gb.z + g2.x * y.x + g2.y * y.y + g2.z * y.z + g2.w * y.w;
                const float e3 = gb.w + g3.x * y.x + g3.y * y.y + g3.z * y.z + g3.w * y.w;
                a_reg[i] = make_float4(a_reg[i].x + e0 * a_reg[i].x, a_reg[i].y + e1 * a_reg[i].y, a_reg[i].z + e2 * a_reg[i].z,
                                       a_reg[i].w + e3 * a_reg[i].w);
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULLK) {
                v = *reinterpret_cast<const float4*>(w_row[i] + k);
            } else if (w_ok[i] && k_ok) {
                v = *reinterpret_cast<const float4*>(w_row[i] + k);
            }
            w_reg[i] = v;
        }
        if (CONV) {  // advance (tap, ch) by BK for the next tile
            ch += BK;
            while (ch >= p.cin) { ch -= p.cin; ++tap; }
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * BK;
        float* ws = Ws + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(as + lds_off<BK>(r0 + RP * i, cc)) = a_reg[i];
#pragma unroll
        for (int i = 0; i < WP; ++i)
            if (r0 + RP * i < BN) *reinterpret_cast<float4*>(ws + lds_off<BK>(r0 + RP * i, cc)) = w_reg[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int li = lane & 31;
    const int lh = lane >> 5;
    const int n_tiles = (p.k + BK - 1) / BK;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < n_tiles;
        if (more) load_tile((kt + 1) * BK);
        const float* as = As + buf * BM * BK;
        const float* ws = Ws + buf * BN * BK;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int chunk = 2 * q + lh;
            const float4 af = *reinterpret_cast<const float4*>(as + lds_off<BK>(32 * wave + li, chunk));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bf = *reinterpret_cast<const float4*>(ws + lds_off<BK>(32 * nt + li, chunk));
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[nt], 0, 0, 0);
            }
            // the next tile goes to the OTHER buffer (last read before the previous barrier): store it from the
            // middle of the MFMA sequence so that neither the global-load wait nor the LDS write sits in front of
            // the barrier
            if (q == BK / 16 - 1 && more) store_tile(buf ^ 1);
        }
        __syncthreads();
    }

    gemm_epilogue<NT>(p, acc, m0, n0, wave, li, lh);
}

// ---- a single clip's products (a streaming chunk: 2-43 blocks of the kernel above) --------------------------------------------------
// v_mfma_f32_32x32x2_f32 retires two k per 64 cycles into ONE accumulator: a wave's chain over K = 576 is 288 dependent instructions =
// 18 k cycles, 9 us, whatever feeds it (180 x 128 x 576 on 8 workgroups: 23 us).  v_mfma_f32_16x16x4_f32 is the same arithmetic — per
// output element the k-ordered chain of fused multiply-adds, its four k in lane-group order (tools/probes/mfma_f32_order_probe.hip:
// 0 of 204 800 outputs differ from the fmaf chain, for either instruction) — at four k per 32 cycles.  Here: 64 x (32 NT) blocks, wave w =
// rows 16 w .. 16 w + 15, two 16-column accumulators per 32 columns (independent chains: back-to-back issue), the k tiles FOUR ahead
// through four register sets (nothing else covers a fetch with one wave per SIMD), the same LDS tiles and swizzle as above.  The k
// sequence of the kernel above is kept — inside a group of 8: 0, 4, 1, 5, 2, 6, 3, 7 — by giving lane group g of instruction j the
// element k = 8 q + 4 (g & 1) + (g >> 1) + 2 j of the 16 B it reads: same bits (tested against the large-grid form).
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int NT, bool CONV, bool FULLK>
__global__ __launch_bounds__(THREADS, 1) void gemm_f32_small_kernel(const GemmArgs p) {
    constexpr int BK = 32, BMS = 64, BN = 32 * NT, D = 4;
    constexpr int CPR = BK / 4, RP = THREADS / CPR, AP = BMS / RP, WP = (BN + RP - 1) / RP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BMS * BK]
    float* Ws = smem + 2 * BMS * BK;   // [2][BN * BK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blocks = (p.n + BN - 1) / BN;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)n_blocks) * BMS;
    const int n0 = (int)(blockIdx.x % (unsigned)n_blocks) * BN;
    const int cc = tid % CPR;
    const int r0 = tid / CPR;
    const float* a_row[AP];
    int a_t[AP];
    bool a_ok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int64_t m = m0 + r0 + RP * i;
        a_ok[i] = m < p.m;
        const int64_t mm = a_ok[i] ? m : 0;
        a_t[i] = CONV ? (int)((unsigned)mm % (unsigned)p.frames) : 0;
        a_row[i] = p.a + mm * p.lda;
    }
    const float* w_row[WP];
    bool w_ok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int nl = r0 + RP * i;
        const int n = n0 + nl;
        w_ok[i] = nl < BN && n < p.n;
        w_row[i] = p.w + (int64_t)(w_ok[i] ? n : 0) * p.ldw;
    }
    int tap = 0, ch = 4 * cc;
    if (CONV) {
        while (ch >= p.cin) { ch -= p.cin; ++tap; }
    }
    const int half = p.taps >> 1;
    float4 a_regs[D][AP], w_regs[D][WP];
    auto load_tile = [&](int k_tile, auto set_) __attribute__((always_inline)) {  // (requested in order: the conv's (tap, channel) bookkeeping advances)
        float4 (&a_reg)[AP] = a_regs[decltype(set_)::value];
        float4 (&w_reg)[WP] = w_regs[decltype(set_)::value];
        const int k = k_tile + 4 * cc;
        const bool k_ok = k < p.k;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CONV) {
                const int ts = a_t[i] + (tap - half) * p.dil;
                if (a_ok[i] && k_ok && ts >= 0 && ts < p.frames)
                    v = *reinterpret_cast<const float4*>(a_row[i] + (int64_t)(tap - half) * p.dil * p.lda + ch);
            } else if (FULLK) {
                v = *reinterpret_cast<const float4*>(a_row[i] + k);
            } else {
                if (a_ok[i] && k_ok) v = *reinterpret_cast<const float4*>(a_row[i] + k);
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULLK) {
                v = *reinterpret_cast<const float4*>(w_row[i] + k);
            } else if (w_ok[i] && k_ok) {
                v = *reinterpret_cast<const float4*>(w_row[i] + k);
            }
            w_reg[i] = v;
        }
        if (CONV) {
            ch += BK;
            while (ch >= p.cin) { ch -= p.cin; ++tap; }
        }
    };
    auto store_tile = [&](int buf, auto set_) __attribute__((always_inline)) {
        const float4 (&a_reg)[AP] = a_regs[decltype(set_)::value];
        const float4 (&w_reg)[WP] = w_regs[decltype(set_)::value];
        float* as = As + buf * BMS * BK;
        float* ws = Ws + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(as + lds_off<BK>(r0 + RP * i, cc)) = a_reg[i];
#pragma unroll
        for (int i = 0; i < WP; ++i)
            if (r0 + RP * i < BN) *reinterpret_cast<float4*>(ws + lds_off<BK>(r0 + RP * i, cc)) = w_reg[i];
    };
    f32x4acc acc[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) acc[i] = f32x4acc{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    const int lh = lg & 1;         // which 16-B half of a group of 8 this lane reads
    const bool odd = (lg >> 1) != 0;  // elements 1, 3 of it (else 0, 2)
    const int n_tiles = (p.k + BK - 1) / BK;
    auto step = [&](int kt, auto set_) __attribute__((always_inline)) {
        constexpr int set = decltype(set_)::value;
        using Next = std::integral_constant<int, (set + 1) % D>;
        const int buf = kt & 1;
        const bool more = kt + 1 < n_tiles;
        if (kt + D < n_tiles) load_tile((kt + D) * BK, set_);
        const float* as = As + buf * BMS * BK;
        const float* ws = Ws + buf * BN * BK;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int chunk = 2 * q + lh;
            const float4 af = *reinterpret_cast<const float4*>(as + lds_off<BK>(16 * wave + li, chunk));
            const float a0 = odd ? af.y : af.x, a1 = odd ? af.w : af.z;
#pragma unroll
            for (int ct = 0; ct < 2 * NT; ++ct) {
                const float4 bf = *reinterpret_cast<const float4*>(ws + lds_off<BK>(16 * ct + li, chunk));
                const float b0 = odd ? bf.y : bf.x, b1 = odd ? bf.w : bf.z;
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[ct], 0, 0, 0);
            }
            if (q == BK / 16 - 1 && more) store_tile(buf ^ 1, Next{});
        }
        __syncthreads();
    };
    using S0 = std::integral_constant<int, 0>;
    load_tile(0, S0{});
    store_tile(0, S0{});
    if (1 < n_tiles) load_tile(1 * BK, std::integral_constant<int, 1>{});
    if (2 < n_tiles) load_tile(2 * BK, std::integral_constant<int, 2>{});
    if (3 < n_tiles) load_tile(3 * BK, std::integral_constant<int, 3>{});
    __syncthreads();
    for (int kt = 0; kt < n_tiles; kt += 4) {
        step(kt, S0{});
        if (kt + 1 < n_tiles) step(kt + 1, std::integral_constant<int, 1>{});
        if (kt + 2 < n_tiles) step(kt + 2, std::integral_constant<int, 2>{});
        if (kt + 3 < n_tiles) step(kt + 3, std::integral_constant<int, 3>{});
    }
    // acc[ct][r] = c[m0 + 16 wave + 4 lg + r][n0 + 16 ct + li]; bias / residual epilogues (the others stay with the kernel above)
#pragma unroll
    for (int ct = 0; ct < 2 * NT; ++ct) {
        const int n = n0 + 16 * ct + li;
        if (n >= p.n) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = m0 + 16 * wave + 4 * lg + r;
            if (m >= p.m) continue;
            float v = acc[ct][r] + bias;
            if (p.epi == EPI_BIAS_RES) v = p.res[m * p.ldres + n] + v;
            p.c[m * p.ldc + n] = v;
        }
    }
}

// (An LDS-DMA staged variant — global_load_lds, three LDS stages — was measured equal to the register-staged kernel in round 2 and
// retired in round 5: git show 782323b:l3ac_amd/csrc/kernels/gemm_f32.hip.)
template <int NT, bool CONV, int BK>
int launch_one(hipStream_t s, const GemmArgs& g) {
    const bool fullk = !CONV && g.k % BK == 0;
    constexpr int BN = 32 * NT;
    const int64_t blocks = ceil_div64(g.m, BM) * ceil_div64(g.n, BN);
    if (blocks <= 0) return L3AC_OK;
    L3AC_REQUIRE(blocks < (int64_t)1 << 31, "gemm: grid too large (m=%lld n=%d)", (long long)g.m, g.n);
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    char name[64];  // instantiation + shape + epilogue: the profile aggregates launches of identical work
    std::snprintf(name, sizeof(name), "gemm_f32_kernel<%d,%s,%d,%s%s> %lldx%dx%d e%d", NT, CONV ? "true" : "false", BK,
                      fullk ? "true" : "false", g.gate_yi ? ",gated" : "", (long long)g.m, g.n, g.k, g.epi);
    const double a_elems = CONV ? (double)g.m * g.cin : (double)g.m * g.k;
    const double c_cols = g.epi == EPI_GEGLU ? (double)g.ldc : (double)g.n;
    ProfScope prof(s, name, 2.0 * (double)g.m * g.n * g.k,
                   4.0 * (a_elems + (double)g.n * g.k + (double)g.m * c_cols * (g.epi == EPI_BIAS_RES ? 2.0 : 1.0)));
    if constexpr (!CONV && BK == 16) {
        if (g.gate_yi) {
            L3AC_REQUIRE(fullk, "gemm: the gated A operand needs k %% 16 == 0 (k=%d)", g.k);
            hipLaunchKernelGGL((gemm_f32_kernel<NT, false, 16, true, true>), dim3((unsigned)blocks), dim3(THREADS), lds, s, g);
            L3AC_LAUNCH_CHECK();
            return L3AC_OK;
        }
    }
    if constexpr (BK == 32) {
        // a single clip (at most a block per four CUs) with a bias / residual epilogue: the 16x16x4 form (same bits)
        if ((g.epi == EPI_BIAS || g.epi == EPI_BIAS_RES) && 4 * blocks <= l3ac_device_cu_count()) {
            const unsigned grid = (unsigned)(ceil_div64(g.m, 64) * ceil_div64(g.n, BN));
            const size_t lds_s = (size_t)2 * (64 + BN) * 32 * sizeof(float);
            if (fullk) hipLaunchKernelGGL((gemm_f32_small_kernel<NT, false, true>), dim3(grid), dim3(THREADS), lds_s, s, g);
            else hipLaunchKernelGGL((gemm_f32_small_kernel<NT, CONV, false>), dim3(grid), dim3(THREADS), lds_s, s, g);
            L3AC_LAUNCH_CHECK();
            return L3AC_OK;
        }
    }
    if (fullk)
        hipLaunchKernelGGL((gemm_f32_kernel<NT, false, BK, true>), dim3((unsigned)blocks), dim3(THREADS), lds, s, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<NT, CONV, BK, false>), dim3((unsigned)blocks), dim3(THREADS), lds, s, g);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

}  // namespace

int launch_gemm(hipStream_t s, const GemmArgs& g) {
    L3AC_REQUIRE(g.a && g.w && g.c, "gemm: null operand");
    L3AC_REQUIRE(g.k > 0 && g.k % 4 == 0, "gemm: k=%d must be a positive multiple of 4", g.k);
    L3AC_REQUIRE(g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm: lda=%lld ldw=%lld must be multiples of 4", (long long)g.lda,
                 (long long)g.ldw);
    L3AC_REQUIRE(((uintptr_t)g.a & 15) == 0 && ((uintptr_t)g.w & 15) == 0, "gemm: operands must be 16-byte aligned");
    const bool conv = g.taps > 1;
    if (conv || g.gate_yi) L3AC_REQUIRE(g.m < ((int64_t)1 << 31), "gemm: implicit-conv / gated A needs m < 2^31 (32-bit row arithmetic)");
    // bf16x3 split route (gemm_split.hip).  The choice depends on the WEIGHT's shape only, never on m: a clip must give
    // bit-identical tokens and samples whether it is coded alone or inside a batch (tests: test_full_batch_properties).
    if (g.gate_yi) {
        L3AC_REQUIRE(!conv && g.gate_stats && g.gate_in_w && g.gate_in_b && g.gate_w && g.gate_b && g.gate_frames > 0 &&
                         g.m % g.gate_frames == 0,
                     "gemm: incomplete gate arguments");
    } else if (g.w_img && gemm_split_eligible(g.n, g.k) && (!conv || gemm_split_conv_ok(g))) {
        return launch_gemm_split(s, g);
    }
    if (conv) {
        L3AC_REQUIRE(g.cin > 0 && g.cin % 4 == 0 && g.k == g.taps * g.cin && g.frames > 0 && g.m % g.frames == 0,
                     "gemm: bad implicit-conv geometry (taps=%d cin=%d k=%d frames=%lld m=%lld)", g.taps, g.cin, g.k,
                     (long long)g.frames, (long long)g.m);
    }
    // K tile: 32 floats (64 KB of LDS per block, 2 blocks per CU) or 16 (32 KB, 3 blocks per CU: more waves to cover
    // the prologue / epilogue of short-K products).
    int bk = 16;  // (the gated kernel exists for 16 only)
    // A grid that leaves CUs idle (a single clip) runs at one block per CU: nothing covers a k tile's fetch -> LDS -> barrier chain,
    // so take half as many of them (same k order inside every group of 8: same bits)
    if (!g.gate_yi && ceil_div64(g.m, BM) * ceil_div64(g.n, 32) <= 256) bk = 32;
#define L3AC_GEMM_LAUNCH(NT_, CONV_) (bk == 16 ? launch_one<NT_, CONV_, 16>(s, g) : launch_one<NT_, CONV_, 32>(s, g))
    if (g.epi == EPI_GEGLU) {
        L3AC_REQUIRE(g.n % 64 == 0 && !conv, "gemm: GEGLU epilogue needs interleaved 64-column tiles");
        return L3AC_GEMM_LAUNCH(4, false);
    }
    if (g.epi == EPI_BIAS_RES) L3AC_REQUIRE(g.res, "gemm: residual epilogue without residual");
    if (g.epi == EPI_SNAKE || g.epi == EPI_SNAKE_GRN) L3AC_REQUIRE(g.alpha && g.inv_alpha, "gemm: snake without alpha");
    if (g.epi == EPI_SNAKE_GRN) L3AC_REQUIRE(g.gamma && g.beta, "gemm: GRN without gamma/beta");
    // widest column tile that does not over-pad n ...
    int nt = g.n <= 32 ? 1 : (g.n <= 64 ? 2 : (g.n <= 96 ? 3 : 4));
    // ... unless that leaves most SIMDs with a single wave (small m): narrower tiles give more workgroups, which is
    // what hides the operand-fetch latency of these short launches (A is re-read once more per column tile, L2-served)
    const int64_t m_panels = ceil_div64(g.m, BM);
    while (nt > 1 && (g.n % (32 * nt) == 0 || nt == 4) && m_panels * ceil_div64(g.n, 32 * nt) < 768 && g.n % (16 * nt) == 0) nt >>= 1;
    if (conv) {
        switch (nt) {
            case 1: return L3AC_GEMM_LAUNCH(1, true);
            case 2: return L3AC_GEMM_LAUNCH(2, true);
            case 3: return L3AC_GEMM_LAUNCH(3, true);
            default: return L3AC_GEMM_LAUNCH(4, true);
        }
    }
    switch (nt) {
        case 1: return L3AC_GEMM_LAUNCH(1, false);
        case 2: return L3AC_GEMM_LAUNCH(2, false);
        case 3: return L3AC_GEMM_LAUNCH(3, false);
        default: return L3AC_GEMM_LAUNCH(4, false);
    }
#undef L3AC_GEMM_LAUNCH
}
