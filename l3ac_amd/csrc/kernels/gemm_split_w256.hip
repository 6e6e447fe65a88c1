// The bf16x3 GEMM of gemm_split.hip for a BATCH's rows and weights of 256-column blocks (the C = 512 stage: 2048 x 512 and 512 x 2048),
// at ONE wave per SIMD with a hand-placed instruction stream.  Built with -fno-slp-vectorize (l3ac_amd/build.py): the split's per-element
// subtractions must stay single-issue instructions beside the MFMAs (packed fp32 costs several times its issue slot there).
#include "gemm_split_common.hpp"
#define L3AC_DIAG_UNIT_GEMM_W256
#include "diag.hpp"

namespace {

// ---- round 6: the C = 512 stage's products at ONE wave per SIMD -----------------------------------------------------------------------
// gemm_split_kernel runs three workgroups per CU, each wave 32 rows x 128 columns: per k tile and wave 96 MFMAs against 88 split
// instructions, 24 fragment reads and 12 copies of W, and A is re-read and re-split by each of the n / 128 column blocks (16 x at N = 2048).
// Counters say no single resource is saturated (MFMA 0.53 busy, vector issue ~50 %): three waves per SIMD cover each other's waits
// only on average.  Here a workgroup is four waves, one per SIMD, and owns 64 RG rows x 256 columns (RG = 3: 192 x 256); a wave owns
// 16 RG rows across all 256 columns — 4 RG x 16 accumulator tiles in AGPRs.  Per k tile and wave: 96 RG MFMAs, 44 RG split instructions
// (A split once per 256 columns), 48 fragment reads, 12 copies of W — 0.9 non-MFMA instructions per MFMA instead of 2.3 — and the
// stream is placed by hand (the pattern of conv_unit_wide_kernel: program order = issue order, a sched_barrier wall every three MFMAs,
// inside it one MFMA then at most two fillers; MI355X_MICROARCH.md: a 16-cycle MFMA holds the issue port for 8, two 4-cycle fillers
// per gap are free):
//   * A(kt + 1) is split (five stages per value pair, split_stage) beside the MFMAs of k tile kt, into the second fragment set;
//     its registers then receive A(kt + 2): one k tile of 6 RG MFMA-cycles x 16 = 4.6 k cycles ahead of use;
//   * W(kt + 1) — two 24-KB image tiles, the column blocks 2 cb and 2 cb + 1 — is requested in two phases of six 16-B loads per lane
//     and stored to the other LDS buffer 700+ cycles later, one ds_write_b128 per window;
//   * the step's only barrier sits in front of its LAST column tile: behind it the first fragments of k tile kt + 1 are read while
//     the last 6 RG MFMAs of k tile kt run, so no step starts with an exposed LDS round trip.
// Same image, same k order, the same six plane products per accumulator in the same order as every other form: same bits.
// Grid (XCD-aware): block b -> XCD b % 8 (round-robin dispatch), j = b / 8: row panel (j / n_cb) * 8 + b % 8, column block j % n_cb —
// the n_cb column blocks of a row panel run side by side on ONE XCD, so the panel's A rows cross the fabric once.
struct SplitPair { float x0, x1, r0, r1; unsigned p0, p1; };
// split2 (split_bf16.hpp) operation for operation, cut into stages of at most four single-issue vector instructions
template <int ST>
__device__ __forceinline__ void split_stage(SplitPair& s, unsigned& o0, unsigned& o1, unsigned& o2) {
    if constexpr (ST == 0) {
        const f32x2_t v = {s.x0, s.x1};
        s.p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        o0 = s.p0;
    } else if constexpr (ST == 1) {
        s.r0 = s.x0 - __builtin_bit_cast(float, s.p0 << 16);
        s.r1 = s.x1 - __builtin_bit_cast(float, s.p0 & 0xffff0000u);
    } else if constexpr (ST == 2) {
        const f32x2_t v = {s.r0, s.r1};
        s.p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        o1 = s.p1;
    } else if constexpr (ST == 3) {
        s.x0 = s.r0 - __builtin_bit_cast(float, s.p1 << 16);
        s.x1 = s.r1 - __builtin_bit_cast(float, s.p1 & 0xffff0000u);
    } else {
        static_assert(ST == 4, "five stages");
        const f32x2_t v = {s.x0, s.x1};
        o2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    }
}

template <int RG>
__global__ __launch_bounds__(THREADS, 1) void gemm_split_kernel_w256(const GemmArgs p) {
    constexpr int WM = 16 * RG, BMW = 4 * WM, NT = 16;
    constexpr int TG = 6 * RG;          // MFMAs (gaps) of one column tile
    constexpr int GAPS = TG * NT;       // ... of a k tile
    constexpr int NST = 20 * RG;        // split stages of a k tile: RG row groups x 4 value pairs x 5 stages
    constexpr int BAR_GAP = TG * (NT - 1);  // the step's barrier: in front of the last column tile
    static_assert(3 * NST + 1 < GAPS, "the split does not fit the k tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int n_cb = p.n / 256;
    const int64_t panels = (p.m + BMW - 1) / BMW;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const int64_t panel = (int64_t)(j / (unsigned)n_cb) * 8 + xcd;
    if (panel >= panels) return;  // (whole workgroups, before any barrier)
    const int cb = (int)(j % (unsigned)n_cb);
    const int64_t m0 = panel * BMW;
    const int n0 = cb * 256;
    const int n_tiles = p.k / BK;  // (whole k tiles, at least two: the launcher)
    const int last = n_tiles - 1;

    W256_STAMP(0);
    const float* a_row[RG];
#pragma unroll
    for (int h = 0; h < RG; ++h) {
        const int64_t row = m0 + WM * wave + 16 * h + ln;
        a_row[h] = p.a + (row < p.m ? row : 0) * p.lda + 8 * lg;  // rows past the edge: row 0, never stored
    }
    // column blocks 2 cb (phase 0) and 2 cb + 1 (phase 1) of the image
    const unsigned char* const w_src = p.w_img + (int64_t)(2 * cb) * n_tiles * W_TILE + 16 * tid;
    const int64_t w_phase = (int64_t)n_tiles * W_TILE;
    // LDS: two buffers of 2 x W_TILE; fragment reads as immediates off two per-lane bases kept opaque (hipcc would otherwise keep a
    // register per base + constant combination)
    auto opaque = [](int v) __attribute__((always_inline)) -> int {
        asm volatile("" : "+v"(v));
        return v;
    };
    const int frag_lane = tile_off(ln, lg);
    const int frag_base = opaque(frag_lane);
    const int store_base = opaque(16 * tid);

    float4 raw[RG][2];
    u32x4 af[2][RG][3];
    u32x4 w_reg[W_LOADS];
    bf16x8 bq[2][3];
    f32x4a acc[2][RG][8];  // [column half][row group][column tile of the half]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < RG; ++h)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[c][h][t] = f32x4a{0.f, 0.f, 0.f, 0.f};

    auto load_a = [&](int kt, int h) __attribute__((always_inline)) {
        raw[h][0] = *reinterpret_cast<const float4*>(a_row[h] + kt * BK);
        raw[h][1] = *reinterpret_cast<const float4*>(a_row[h] + kt * BK + 4);
    };
    auto load_w = [&](int kt, int phase) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i)
            w_reg[i] = *reinterpret_cast<const u32x4*>(w_src + phase * w_phase + (int64_t)kt * W_TILE + 16 * THREADS * i);
    };
    // (fragment planes are read in the order 2, 1, 0: a column tile's first MFMA takes plane 0 of the weights, the YOUNGEST read — LDS returns in order)
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: A(0) split, A(1) requested, W(0) in buffer 0, fragments of column tile 0 -----------------------------
#pragma unroll
    for (int h = 0; h < RG; ++h) load_a(0, h);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        load_w(0, ph);
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i) *reinterpret_cast<u32x4*>(smem_split + store_base + ph * W_TILE + 16 * THREADS * i) = w_reg[i];
    }
#pragma unroll
    for (int h = 0; h < RG; ++h) {
        unsigned pl[3][4];
        split2(raw[h][0].x, raw[h][0].y, pl[0][0], pl[1][0], pl[2][0]);
        split2(raw[h][0].z, raw[h][0].w, pl[0][1], pl[1][1], pl[2][1]);
        split2(raw[h][1].x, raw[h][1].y, pl[0][2], pl[1][2], pl[2][2]);
        split2(raw[h][1].z, raw[h][1].w, pl[0][3], pl[1][3], pl[2][3]);
#pragma unroll
        for (int q = 0; q < 3; ++q) af[0][h][q] = u32x4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]};
        load_a(1 < last ? 1 : last, h);
    }
    lds_barrier();
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) bq[0][pl] = *reinterpret_cast<const bf16x8*>(smem_split + frag_base + pl * W_PLANE);

    // One k tile: MFMAs of tile kt from af[0] and LDS buffer kt & 1; af[1] <- split A(kt + 1), copied to af[0] at the end (one loop body:
    // unrolled over the two parities, hipcc permuted the accumulators through VGPRs at the back edge); raw <- A(kt + 2); W(kt + 1) -> the
    // other buffer.
    // Timetable.  Window w = MFMA gaps 3 w .. 3 w + 2 (96 windows per k tile at RG = 3).  Measured with bounding builds (profiles/r06/
    // gemm_w256.md): a vector-memory instruction issued by all four waves in the same gap costs each of them ~50 cycles — the CU's one
    // address path takes them one after the other, 16 cycles each — and the step's 18 of them cost 900 of its 5 900 cycles.  So every wave
    // has its OWN timetable (PH = wave): vector-memory slot s sits in window 4 s + PH, one instruction per window and CU, and the LDS
    // stores of W likewise (ds_write_b128 holds the store path for 13-26 cycles).  Slots: 0-5 W phase 0 | 6-9 A(kt + 2) of row groups 0, 1
    // | 10-15 W phase 1 | 16-17 A(kt + 2) of row group 2; a W load is stored 20 windows (~1 k cycles) later.
    auto k_loop = [&](auto ph_) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_)::value;
        static_assert(RG == 3, "the timetable below is written for 96 windows");
#pragma unroll 1
        for (int kt = 0; kt < n_tiles; ++kt) {
            const int kt1 = kt + 1 < last ? kt + 1 : last, kt2 = kt + 2 < last ? kt + 2 : last;
            const int buf_off = (kt & 1) * (2 * W_TILE), nxt_off = 2 * W_TILE - buf_off;  // (wave-uniform)
            const int fb_cur = frag_base + buf_off, fb_nxt = frag_base + nxt_off, sb_nxt = store_base + nxt_off;
            auto read_b = [&](int base, int t, int pl, bf16x8 (&b)[3]) __attribute__((always_inline)) {
                b[pl] = *reinterpret_cast<const bf16x8*>(smem_split + base + (t >> 3) * W_TILE + pl * W_PLANE + 1024 * (t & 7));
            };
            SplitPair sp[4];
            unsigned planes[3][4];
            tail_for<GAPS>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value, t = g / TG, r = g % TG, q = r / RG, h = r % RG, win = g / 3;
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
                if constexpr (g == BAR_GAP) lds_barrier();  // every wave's stores of W(kt + 1) are in LDS; every wave has read the last fragments of W(kt)
                acc[t >> 3][h][t & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[0][h][PA[q]]), bq[t & 1][PB[q]],
                                                                                  acc[t >> 3][h][t & 7], 0, 0, 0);
                // fragments of the next column tile (the last tile: column tile 0 of k tile kt + 1, behind the barrier)
                if constexpr (r % RG == 0 && r / RG < 3) {
                    constexpr int pl = 2 - r / RG;
                    if constexpr (t + 1 < NT) read_b(fb_cur, t + 1, pl, bq[(t + 1) & 1]);
                    else read_b(fb_nxt, 0, pl, bq[0]);
                }
                // split of A(kt + 1): stage j = (row group, stage, pair), the four pairs of a row group interleaved
                if constexpr (g % 3 == 1 && win < NST) {
                    constexpr int hh = win / 20, st = (win % 20) / 4, pr = (win % 20) % 4;
                    if constexpr (st == 0) {
                        const float4 v = raw[hh][pr >> 1];
                        sp[pr].x0 = (pr & 1) ? v.z : v.x;
                        sp[pr].x1 = (pr & 1) ? v.w : v.y;
                    }
                    split_stage<st>(sp[pr], planes[0][pr], planes[1][pr], planes[2][pr]);
                    if constexpr (st == 4 && pr == 3) {
#pragma unroll
                        for (int u = 0; u < 3; ++u) af[1][hh][u] = u32x4{planes[u][0], planes[u][1], planes[u][2], planes[u][3]};
                    }
                }
                // this wave's vector-memory slot / LDS store of the window
                if constexpr (g % 3 == 2) {
                    if constexpr (win >= PH && (win - PH) % 4 == 0 && (win - PH) / 4 < 18) {
                        constexpr int slot = (win - PH) / 4;
                        if constexpr (slot < 6) w_reg[slot] = *reinterpret_cast<const u32x4*>(w_src + (int64_t)kt1 * W_TILE + 16 * THREADS * slot);
                        if constexpr (slot >= 10 && slot < 16)
                            w_reg[slot - 10] = *reinterpret_cast<const u32x4*>(w_src + w_phase + (int64_t)kt1 * W_TILE + 16 * THREADS * (slot - 10));
                        // (a row group's registers are free once its pairs have passed split stage 1: window 20 hh + 7)
                        if constexpr (slot >= 6 && slot < 10) raw[(slot - 6) >> 1][slot & 1] = *reinterpret_cast<const float4*>(a_row[(slot - 6) >> 1] + kt2 * BK + 4 * (slot & 1));
                        if constexpr (slot >= 16) raw[2][slot & 1] = *reinterpret_cast<const float4*>(a_row[2] + kt2 * BK + 4 * (slot & 1));
                    }
                    if constexpr (win >= 20 + PH && (win - 20 - PH) % 4 == 0 && (win - 20 - PH) / 4 < 6)
                        *reinterpret_cast<u32x4*>(smem_split + sb_nxt + 16 * THREADS * ((win - 20 - PH) / 4)) = w_reg[(win - 20 - PH) / 4];
                    if constexpr (win >= 60 + PH && (win - 60 - PH) % 4 == 0 && (win - 60 - PH) / 4 < 6)
                        *reinterpret_cast<u32x4*>(smem_split + sb_nxt + W_TILE + 16 * THREADS * ((win - 60 - PH) / 4)) = w_reg[(win - 60 - PH) / 4];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x126, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#pragma unroll
            for (int h = 0; h < RG; ++h)
#pragma unroll
                for (int u = 0; u < 3; ++u) af[0][h][u] = af[1][h][u];
        }
    };
    W256_STAMP(1);
    switch (wave) {  // (wave-uniform)
        case 0: k_loop(std::integral_constant<int, 0>{}); break;
        case 1: k_loop(std::integral_constant<int, 1>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        default: k_loop(std::integral_constant<int, 3>{}); break;
    }
    W256_STAMP(2);
    epilogue_rows<RG, 4>(p, [&](int sq, int h, int tt) __attribute__((always_inline)) -> f32x4a& { return acc[sq >> 1][h][4 * (sq & 1) + tt]; }, m0, n0, wave, lane,
                         reinterpret_cast<float*>(smem_split + 2 * W_TILE + 12288 * wave));
    W256_STAMP(3);
}


}  // namespace

// shapes and alignments the 256-column form takes (the dispatcher in gemm_split.hip asks; anything else stays on gemm_split_kernel)
bool gemm_split_w256_ok(const GemmArgs& g) {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return g.taps <= 1 && g.k % BK == 0 && g.k / BK >= 2 && g.n % 256 == 0 && g.epi != EPI_GEGLU && g.ldc % 4 == 0 && al16(g.c) && al16(g.bias) &&
           (g.epi != EPI_BIAS_RES || (g.ldres % 4 == 0 && al16(g.res))) &&
           ((g.epi != EPI_SNAKE && g.epi != EPI_SNAKE_GRN) || (al16(g.alpha) && al16(g.inv_alpha))) && (g.epi != EPI_SNAKE_GRN || (al16(g.gamma) && al16(g.beta)));
}

int launch_gemm_split_w256(hipStream_t s, const GemmArgs& g) {
    L3AC_REQUIRE(gemm_split_w256_ok(g), "split gemm (256-column form): unsupported shape or alignment n=%d k=%d", g.n, g.k);
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel_w256<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * W_TILE));
        configured.done();
    }
    const int64_t panels = ceil_div64(g.m, 192);
    const int64_t grid = ceil_div64(panels, 8) * 8 * (g.n / 256);
    L3AC_REQUIRE(grid < (int64_t)1 << 31, "split gemm: grid too large (m=%lld n=%d)", (long long)g.m, g.n);
    hipLaunchKernelGGL((gemm_split_kernel_w256<3>), dim3((unsigned)grid), dim3(THREADS), 4 * W_TILE, s, g);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
