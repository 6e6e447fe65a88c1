// One LocalTrans STACK (all layers of one reference `LocalTrans`, l3ac/local_trans.py:7-53) per launch, one workgroup per clip.
// Arithmetic of PyPI local-attention==1.11.2 as configured by local_trans.py:23-53 (see attention.hip / DESIGN.md §4): per layer
//
//     x = x + to_out( causal_attention( to_qkv( LayerNorm(x) ) ) + distance bias )          (LocalMHA, prenorm, heads 6 x 32)
//     x = x + W2( GEGLU( W1( LayerNorm(x) ) ) )                                                (FeedForward, inner 341)
//
// What it replaces: 7 launches per layer (LayerNorm, qkv GEMM, attention, out GEMM, LayerNorm, FF-in GEMM, FF-out GEMM — 56
// launches and ~1.75 ms of the 256-clip step for 6 % of its FLOPs, ~45 % of a streaming chunk), each a round trip of the
// [rows][128..704] tensors through L2 / HBM with a prologue and an epilogue around a K = 128..344 product.
//
// Here a clip's rows never leave the CU.  Clips are independent and, for the clip lengths this kernel takes (frames <= the
// attention window and <= 192), every layer is a single window: ONE workgroup owns a clip for the whole stack.
//   * wave w owns frames 16 w .. 16 w + 15 (the columns of every MFMA tile) from the first LayerNorm to the last residual; the
//     residual stream x lives in its registers in accumulator layout (channel 16 t + 4 g + i in register i of tile t, lane
//     (frame f, group g)), which is at once the LayerNorm's input (channel sums = 32 registers + two cross-lane steps) and,
//     split into bf16 planes, the B operand of the next product in the k order sigma(g, j) = (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
//     (guide: 'An accumulator tile as the next MFMA's operand'; conv_unit_wide.hip uses the same order);
//   * every product is weights (A) x activations (B) on v_mfma_f32_16x16x32_bf16 with both operands as exact bf16x3 splits
//     (split_bf16.hpp: 6 plane products per fp32 MAC, error <= the fp32 instruction's), so each result tile is again in
//     accumulator layout: q^T and k^T as computed; V with the operands SWAPPED (activations as A, weights as B: the same weight
//     bytes) so that its tile is V[frame][d] — what P.V needs as its A operand; S^T = K.Q^T has keys on its rows, so the
//     probabilities are directly the B operand of O^T = V^T.P^T; O^T that of the out projection; GEGLU's tile that of W2;
//   * K and V^T of the current head are the only activations that cross waves: written to LDS as operand fragments by the
//     producing lane itself (the lane that computed element (d, frame) is the lane that will read it as (key, d));
//   * the weights of the stack are ONE stream of fragment-ordered bf16x3 "pieces" (16 rows x 32 k, 3 planes x 1 KB) in
//     consumption order (trans_stack_image), L2-resident (1.37 MB per layer, every CU walks it in step), staged by LDS-DMA into a
//     ring of 4 slots of 4 pieces behind a counted s_waitcnt vmcnt and one raw s_barrier per slot.
// Online softmax over key steps of 32 (as attention_mfma_kernel) with ts_exp_neg (1-2 ulp); erff is the OCML one.
// Results differ from the unfused route's in rounding only (other summation orders); clip i of a batch is bit-identical to
// clip i alone (nothing depends on the batch).
//
// Few clips (a streaming chunk is ONE): a clip on one workgroup is a clip on one of 256 CUs.  The COOPERATIVE form (KS = 6) gives a
// clip six co-resident workgroups: workgroup j runs head j of every attention sub-layer and hidden chunks 2 j, 2 j + 1 of every
// FeedForward for ALL frames of the clip, streaming only those weights (20 of a layer's 114 ring slots).  A sub-layer's output is
// then a sum of six PARTIAL tiles (one per head / per chunk pair), each accumulated from zero; the partials cross workgroups through
// write-through slabs in global memory behind an arrival counter (guide: 'inter-workgroup visibility', the hand-off form measured
// with sc1 loads in place of the acquire — producer: every byte by 16-B sc1 stores -> every wave's vmcnt(0) -> workgroup barrier ->
// ONE lane's agent-scope add; consumer: that lane polls with sc1 loads -> workgroup barrier -> every load of the bytes a 16-B sc1
// buffer load to registers; no plain load ever touches a slab, so no L1 line of it can be stale) and EVERY workgroup adds them in the
// same fixed order ((((p0 + p1) + p2) + p3) + p4) + p5 — which is also how the one-workgroup form (KS = 1) now sums its heads and
// chunk pairs, so the two forms return the same bits and batch invariance holds across them.
#include "../kernels.hpp"
#include "../network.hpp"
#include "device_math.hpp"
#include "split_bf16.hpp"
#include "ring_common.hpp"
#define L3AC_DIAG_UNIT_TRANS_STACK
#include "diag.hpp"

#include <cmath>
#include <vector>

namespace {

constexpr int TS_DIM = 128, TS_HEADS = 6, TS_DH = 32, TS_INNER = TS_HEADS * TS_DH;
constexpr int TS_FFI = 341;                    // int(128 * 4 * 2 / 3)
constexpr int TS_FF_CHUNKS = (TS_FFI + 31) / 32;  // hidden units in chunks of 32 (zero padded)
constexpr int TS_PIECE = 3072, TS_SLOT_PIECES = 4, TS_SLOT = TS_SLOT_PIECES * TS_PIECE;
constexpr int TS_PIECES_PER_HEAD = 32, TS_PIECES_PER_CHUNK = 24;
constexpr int TS_PIECES_PER_LAYER = TS_HEADS * TS_PIECES_PER_HEAD + TS_FF_CHUNKS * TS_PIECES_PER_CHUNK;  // 456
constexpr int TS_SLOTS_PER_LAYER = TS_PIECES_PER_LAYER / TS_SLOT_PIECES;                                // 114
static_assert(TS_PIECES_PER_LAYER % TS_SLOT_PIECES == 0, "a layer is a whole number of ring slots");
constexpr int TS_MAX_FRAMES = 192, TS_MAX_LAYERS = 8;
static_assert(TS_MAX_FRAMES == 192, "slab geometry");
// LDS (bytes) of the instantiation for at most MAXW waves: the K / V^T fragments need MAXW key tiles, what that leaves goes to the
// weight ring — 4 slots (3 in flight) at 12 waves, 8 (7 in flight, 84 KB) at 4 waves: with ONE wave per SIMD a slot is consumed
// in ~0.2 us, and a single clip (the streaming chunk) pulls its weights from beyond L2: the stream has to run that far ahead
template <int MAXW>
struct TsLds {
    // MAXW <= 4 (clips of at most 64 frames: one computing wave per SIMD, nothing hides a wave's own overheads, and 60 frames of
    // work per 1.37 MB of weights make the stream itself the bound: ~25 GB/s per issuing wave): four dedicated LOADER waves, one per
    // SIMD, issue the LDS-DMAs (each costs its issuing wave ~60-180 cycles); slots are synchronised in PAIRS (one barrier and one
    // exposed first-fragment read per 48 MFMAs instead of 24); the ring is 8 slots deep (a single clip — the streaming chunk —
    // pulls its weights from beyond L2).  Otherwise six of the computing waves copy two 1-KB blocks of every slot each.
    static constexpr bool LOADER = MAXW <= 4;
    static constexpr int RING_SLOTS = LOADER ? 8 : 4;     // (a power of two)
    static constexpr int SG = LOADER ? 2 : 1;              // slots per synchronisation group
    static constexpr int NDW = LOADER ? 4 : 6;             // waves that issue the copies
    static constexpr int BPW = 12 / NDW;                   // 1-KB blocks per such wave and slot
    static constexpr int GROUPS = RING_SLOTS / SG, PFG = GROUPS - 1;  // groups in the ring / in flight
    static constexpr int WAIT = BPW * SG * (PFG - 1);      // a DMA wave's copies that may stay outstanding at a group's end
    static constexpr int THREADS = 64 * (MAXW + (LOADER ? NDW : 0));
    static constexpr int OFF_RING = 0;
    static constexpr int OFF_K = OFF_RING + RING_SLOTS * TS_SLOT;        // K fragments: [key tile MAXW][plane][lane] 16 B
    static constexpr int OFF_V = OFF_K + MAXW * TS_PIECE;                // V^T fragments: [key step MAXW / 2][d tile 2][plane][lane] 16 B
    static constexpr int OFF_BIAS = OFF_V + MAXW * TS_PIECE;             // distance bias [heads][192]
    static constexpr int OFF_LN = OFF_BIAS + TS_HEADS * TS_MAX_FRAMES * 4;  // LayerNorm parameters [layers][4][128]
    static constexpr int bytes(int layers) { return OFF_LN + layers * 4 * TS_DIM * 4; }
    static_assert(bytes(TS_MAX_LAYERS) <= 160 * 1024 && WAIT <= 63 && TS_SLOTS_PER_LAYER % SG == 0, "LDS budget / vmcnt field / slot pairing");
};

struct TransStackArgs {
    float* x;                  // [batch][frames][128], updated in place
    int frames, n_layers;
    const unsigned char* img;  // n_layers * TS_SLOTS_PER_LAYER slots
    const float* ln;           // [n_layers][4][128]: norm.weight, norm.bias (LocalMHA), ff.0.weight, ff.0.bias
    const float* bias_table;   // [heads][table_stride], entry = bias at distance i - j
    int table_stride;
    float scale;               // dim_head^-0.5 (applied to q before the scores, as the package does)
    // cooperative form only (KS > 1): per clip slot, TS_COOP_SLAB_FLOATS of partial slabs [parity 2][part KS][frame 192][128],
    // KS private copies of the residual stream [192][128], and two counters (arrivals of the current launch, workgroups done)
    float* coop_slab;
    float* coop_x;
    unsigned* coop_cnt;
    int batch;
    int coop_slots;            // clip slots of this launch: the batch rounded up to a multiple of 8 (grid = TS_KS * coop_slots)
    // An arrival poll that is not answered within `coop_timeout_ticks` (s_memrealtime, 100 MHz) is COUNTED in *coop_fail (host-visible,
    // system-scope add) and the workgroup stops waiting for the rest of the launch: the launch ends promptly with invalid numbers
    // and the host hears about it (l3ac_coop_timeout_count / the next call on the context).  coop_fault_part >= 0 (test hook):
    // that workgroup of every clip withholds its first arrival.
    unsigned* coop_fail;
    unsigned coop_timeout_ticks;
    int coop_fault_part;
};
constexpr int TS_KS = 6;                                    // workgroups per clip of the cooperative form (= heads)
constexpr int TS_COOP_CLIPS = 32;                           // clip slots (a multiple of 8: blocks b and b + 8 are dealt to the same XCD — speed only)
constexpr int64_t TS_COOP_SLAB_FLOATS = 2LL * TS_KS * 192 * TS_DIM;
constexpr int64_t TS_COOP_X_FLOATS = (int64_t)TS_KS * 192 * TS_DIM;

// exp(x) for x <= 0 (softmax terms; x = -inf for masked entries): the hardware's 2^t on t = x log2(e) carried in two parts — the
// rounded product and its residual (exact by fma) plus the constant's low part — with the residual applied to first order:
// 2^(t + r) = 2^t (1 + r ln 2), |r| < 2^-22.  6 instructions and 1-2 ulp, against ~20 for the OCML expf of the same accuracy;
// the exponentials were a quarter of the vector work of an attention step.
__device__ __forceinline__ float ts_exp_neg(float x) {
    x = fmaxf(x, -1.0e4f);  // (-inf -> a value whose exponential underflows to 0 as well; keeps the residual finite)
    const float t = x * 1.44269502162933349609375f;
    const float r = fmaf(x, 1.44269502162933349609375f, -t) + x * 1.92596299112661746e-8f;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147182464599609375f, e);
}

// MAXW: most waves a workgroup of this instantiation is launched with (register budget 512 / ceil(MAXW / 4) per lane).  Where the
// budget allows (MAXW <= 8: clips of at most 128 frames, e.g. the 60-token stages) the next weight fragment is fetched from LDS
// while the current one multiplies; with one wave per SIMD nothing else covers that latency.

// (amdgpu_waves_per_eu: one workgroup per CU, so THREADS / 256 waves per SIMD is all the occupancy there will ever be — told so, the
// scheduler spends the registers on keeping fragments in flight instead of re-reading them next to their use behind a full wait)
template <int MAXW, int KS>
__global__ __launch_bounds__(TsLds<MAXW>::THREADS) __attribute__((amdgpu_waves_per_eu(TsLds<MAXW>::THREADS / 256, TsLds<MAXW>::THREADS / 256)))
void trans_stack_kernel(const TransStackArgs p) {
    constexpr bool PREF = MAXW <= 8;
    constexpr bool COOP = KS > 1;
    static_assert(KS == 1 || KS == TS_HEADS, "one workgroup per clip, or one per head");
    using L = TsLds<MAXW>;
    // cooperative form: block = part * coop_slots + clip slot, coop_slots a multiple of 8 (the KS workgroups of a clip are then a
    // multiple of 8 blocks apart: one XCD under the observed round-robin placement); slots beyond the batch have nothing to do
    const int clip = COOP ? (int)(blockIdx.x % (unsigned)p.coop_slots) : (int)blockIdx.x;
    const int part = COOP ? (int)(blockIdx.x / (unsigned)p.coop_slots) : 0;
    if (COOP && clip >= p.batch) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ts[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, lg = lane >> 4;
    const int frames = p.frames;
    TS_STAMP_INIT();
    const int frame = 16 * wave + fl;            // this lane's frame (column of every tile)
    const bool frame_ok = frame < frames;
    float* const xclip = p.x + (int64_t)clip * frames * TS_DIM;
    float* const xtrue = xclip + (int64_t)(frame_ok ? frame : 0) * TS_DIM + 4 * lg;  // + 16 t: this lane's 4 channels of tile t
    // where this workgroup keeps the residual stream between sub-layers: the tensor itself, or (cooperative form: six workgroups
    // hold the same stream and none may see another's update early) a private copy
    // (the private copy is laid out [tile 8][frame 192][16]: a wave instruction covers 1 KB of whole lines, as for the slabs below)
    constexpr int XT = COOP ? 192 * 16 : 16;  // floats from tile t to tile t + 1 of this lane's frame
    float* const xlane = COOP ? p.coop_x + ((int64_t)clip * TS_KS + part) * (192 * TS_DIM) + (int64_t)(frame_ok ? frame : 0) * 16 + 4 * lg : xtrue;
    float* const bias_s = reinterpret_cast<float*>(smem_ts + L::OFF_BIAS);
    float* const ln_s = reinterpret_cast<float*>(smem_ts + L::OFF_LN);

    // ---- the residual stream of this wave's 16 frames: xr[t][i] = x[frame][16 t + 4 g + i].  It is in registers only around a
    // LayerNorm: the products of a sub-layer accumulate from ZERO (yacc) and are added to x once, at the sub-layer's end —
    // accumulated directly onto x, every one of the ~100 MFMA steps of a layer rounded at the magnitude of x (measured: 5 x the
    // error of the unfused route against fp64).  In between x waits in its own rows of the global tensor (L2).
    f32x4_t xr[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        xr[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (frame_ok) xr[t] = *reinterpret_cast<const f32x4_t*>(xtrue + 16 * t);
        if (COOP && frame_ok) *reinterpret_cast<f32x4_t*>(xlane + XT * t) = xr[t];
    }
    // distance bias (distances 0 .. frames - 1 of every head) and the stack's LayerNorm parameters
    for (int i = tid; i < TS_HEADS * TS_MAX_FRAMES; i += blockDim.x) {
        const int h = i / TS_MAX_FRAMES, d = i % TS_MAX_FRAMES;
        bias_s[i] = d < frames ? p.bias_table[(int64_t)h * p.table_stride + d] : 0.f;
    }
    for (int i = tid; i < p.n_layers * 4 * TS_DIM; i += blockDim.x) ln_s[i] = p.ln[i];
    __syncthreads();  // (every plain load above is drained here, before the first hand-counted LDS-DMA is issued)

    // ---- the weight stream -------------------------------------------------------------------------------------------
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem_ts + L::OFF_RING);
    const unsigned lane_off = 16u * (unsigned)lane;
    // slots this workgroup consumes per layer, in its own order: all 114, or (cooperative) head `part` (8 slots) then hidden chunks
    // 2 part, 2 part + 1 (6 slots each; the last workgroup has chunk 10 alone)
    const int my_chunks = COOP ? (2 * part + 1 < TS_FF_CHUNKS ? 2 : 1) : TS_FF_CHUNKS;
    const int slots_per_layer = COOP ? 8 + 6 * my_chunks : TS_SLOTS_PER_LAYER;
    const int total_slots = p.n_layers * slots_per_layer;
    auto image_slot = [&](int n) -> int {  // n-th slot of this workgroup's sequence -> slot of the image (wave-uniform)
        if (!COOP) return n;
        const int layer = n / slots_per_layer, r = n - layer * slots_per_layer;
        return layer * TS_SLOTS_PER_LAYER + (r < 8 ? 8 * part + r : 8 * TS_HEADS + 12 * part + (r - 8));
    };
    int dma_slot = 0;  // next slot of the stream to fetch (wave-uniform); past the end it wraps around (never consumed)
    const int n_compute = (int)(blockDim.x >> 6) - (L::LOADER ? L::NDW : 0);
    const int dma_wave = L::LOADER ? (wave >= n_compute ? wave - n_compute : -1) : (wave < L::NDW ? wave : -1);  // this wave's share of the copies, -1: none
    auto issue_group = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < L::SG; ++g) {
            if (dma_wave >= 0) {
                const int src = dma_slot < total_slots ? dma_slot : dma_slot - total_slots;
                const unsigned char* sb = p.img + (int64_t)image_slot(src) * TS_SLOT + 1024 * dma_wave;
                const unsigned db = ring_lds + (unsigned)((dma_slot & (L::RING_SLOTS - 1)) * TS_SLOT + 1024 * dma_wave);
#pragma unroll
                for (int i = 0; i < L::BPW; ++i) ring_dma_1k(sb + 1024 * L::NDW * i, lane_off, db + 1024u * (unsigned)(L::NDW * i));
            }
            ++dma_slot;
        }
    };
    // end of a group of slots: this wave's copies of the NEXT group have landed (all but the youngest PFG - 1 groups'; plain loads /
    // stores of x in the queue only make the wait stronger), its own LDS reads and writes are done, then everybody's are; the
    // group just consumed is free for the next step's DMA
    auto step_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(L::WAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int j = 0; j < L::PFG; ++j) issue_group();
    step_sync();  // group 0 has landed
    TS_STAMP(0);  // prologue
    if (L::LOADER && wave >= n_compute) {  // a loader: one refill and one barrier per group, in step with the computing waves
        const int groups_per_layer = slots_per_layer / L::SG;
        for (int g = 0; g < total_slots / L::SG; ++g) {
            issue_group();
            step_sync();
            if (COOP) {  // the computing waves' exchange of partials has two workgroup barriers: after the head's last group and
                         // after the layer's last group (coop_combine below) — joined here, or the group barriers would pair up wrongly
                const int r = g % groups_per_layer;
                if (r == 8 / L::SG - 1 || r == groups_per_layer - 1) {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    int slot_no = 0;  // slots consumed so far (wave-uniform)
    const unsigned char* const ring_lane = smem_ts + L::OFF_RING + 16 * lane;
    auto load_frag = [&](bf16x8 (&f)[3], const unsigned char* a) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(a + 1024 * pl);
    };
    // One slot step: the slot's 4 weight pieces in order, body(j, fragment) for piece j (compile-time j); `last` runs after the
    // last piece's products and before the step's barrier (LDS writes other waves read after it).
    auto slot_step = [&](auto&& body, auto&& last) __attribute__((always_inline)) {
        const bool group_begins = (slot_no & (L::SG - 1)) == 0, group_ends = (slot_no & (L::SG - 1)) == L::SG - 1;
        if (!L::LOADER && group_begins) issue_group();  // refill the group consumed one step ago
        const unsigned char* a = ring_lane + (slot_no & (L::RING_SLOTS - 1)) * TS_SLOT;
        bf16x8 f[PREF ? 2 : 1][3];
        load_frag(f[0], a);
        ring_static_for<TS_SLOT_PIECES>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (PREF) {
                if constexpr (j + 1 < TS_SLOT_PIECES) load_frag(f[(j + 1) & 1], a + (j + 1) * TS_PIECE);
                body(j_, f[j & 1]);
            } else {
                body(j_, f[0]);
                if constexpr (j + 1 < TS_SLOT_PIECES) load_frag(f[0], a + (j + 1) * TS_PIECE);
            }
        });
        last();
        ++slot_no;
        if (group_ends) step_sync();
    };
    // Two consecutive slot steps (every phase is a whole number of pairs and starts on an even slot).  Where slots are synchronised in
    // pairs (LOADER) the second slot's fragments are fetched while the first slot's products run: one exposed LDS round trip per
    // 48 MFMAs.  body(half, j, fragment), last(half).
    auto slot_pair = [&](auto&& body, auto&& last) __attribute__((always_inline)) {
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        if constexpr (L::SG == 2) {
            const unsigned char* a = ring_lane + (slot_no & (L::RING_SLOTS - 1)) * TS_SLOT;  // (slot_no even: the pair is contiguous in the ring)
            bf16x8 fa[TS_SLOT_PIECES][3], fb[TS_SLOT_PIECES][3];
#pragma unroll
            for (int j = 0; j < TS_SLOT_PIECES; ++j) load_frag(fa[j], a + j * TS_PIECE);
            ring_static_for<TS_SLOT_PIECES>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                load_frag(fb[j], a + TS_SLOT + j * TS_PIECE);
                body(H0{}, j_, fa[j]);
            });
            last(H0{});
            ring_static_for<TS_SLOT_PIECES>([&](auto j_) { body(H1{}, j_, fb[decltype(j_)::value]); });
            // the issue order, spelled out: left alone hipcc sinks every fragment read next to its use behind a full s_waitcnt (105
            // such waits for 360 MFMAs, ~450 cycles per slot of exposed LDS latency with one wave per SIMD)
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * TS_SLOT_PIECES, 0);  // the first slot's fragments
#pragma unroll
            for (int j = 0; j < TS_SLOT_PIECES; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // one fragment of the second slot ...
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);  // ... in front of one piece's products of the first
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * TS_SLOT_PIECES, 0);
            last(H1{});
            slot_no += 2;
            step_sync();
        } else {
            slot_step([&](auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) { body(H0{}, j_, f); }, [&]() __attribute__((always_inline)) { last(H0{}); });
            slot_step([&](auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) { body(H1{}, j_, f); }, [&]() __attribute__((always_inline)) { last(H1{}); });
        }
    };
    auto nothing = [](auto) __attribute__((always_inline)) {};

    // LayerNorm over the 128 channels of this lane's frame (F.layer_norm, eps 1e-5: two-pass, rstd = 1 / sqrt(var + eps) as
    // rows.hip), then the bf16x3 planes of the result: ap[s] = k step s (channels 32 s .. 32 s + 31)
    bf16x8 ap[4][3];
    auto layer_norm_planes = [&](const float* w, const float* b) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += (xr[t][0] + xr[t][1]) + (xr[t][2] + xr[t][3]);
        s = rows_sum(s);
        const float mean = s / (float)TS_DIM;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = xr[t][i] - mean;
                q = fmaf(d, d, q);
            }
        q = rows_sum(q);
        const float rstd = 1.0f / sqrtf(q / (float)TS_DIM + 1e-5f);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            f32x4_t a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s2 + u;
                const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(w + 16 * t + 4 * lg);
                const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(b + 16 * t + 4 * lg);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[u][i] = (xr[t][i] - mean) * rstd * wv[i] + bv[i];
            }
            planes_of(a[0], a[1], ap[s2]);
        }
    };
    // A sub-layer's output is the sum of TS_HEADS partial tiles, each accumulated from zero in `tacc` (one per head; one per pair of
    // hidden chunks), added in index order: yacc = ((((p0 + p1) + p2) + p3) + p4) + p5.  One workgroup per clip: it computes them one
    // after the other.  Cooperative form: workgroup `part` computes partial `part`, all six exchange them through the slabs.
    f32x4_t yacc[8], tacc[8];
    int coop_phase = 0;  // sub-layers finished so far (wave-uniform)
    bool coop_dead = false;  // (the polling lane's) an arrival poll of this launch has expired: counted, no further waiting
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(COOP ? p.coop_slab + (int64_t)clip * TS_COOP_SLAB_FLOATS : p.x), 0, (int)(TS_COOP_SLAB_FLOATS * 4), 0x00020000);
    auto add_partial = [&](int index) __attribute__((always_inline)) {  // KS == 1: partial `index` is complete in tacc
#pragma unroll
        for (int t = 0; t < 8; ++t) yacc[t] = index == 0 ? tacc[t] : yacc[t] + tacc[t];
    };
    constexpr bool XPRE = COOP && MAXW <= 8;  // (at 12 waves — 168 registers — the eight rows spill)
    f32x4_t xpre[XPRE ? 8 : 1];  // cooperative form: the residual rows, requested before the exchange and added after it
    auto coop_combine = [&]() __attribute__((always_inline)) {  // KS > 1: tacc (this workgroup's partial) -> yacc (the sum of all six)
        if constexpr (XPRE) {
#pragma unroll
            for (int t = 0; t < 8; ++t) xpre[t] = frame_ok ? *reinterpret_cast<const f32x4_t*>(xlane + XT * t) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        // byte offset of this lane's 16 B of tile t in partial k of the phase's slab: + 192 * 64 t + k * 192 * 512.  A partial is laid out
        // [tile 8][frame 192][64 B]: one wave instruction (16 frames x 4 groups x 16 B) then covers 1 KB of whole 128-B lines; in the
        // tensor's own [frame][128] order it touched 16 half lines, and the exchange is bound by the line requests of these
        // L1-bypassing accesses, not by where the lines are
        const unsigned slab_off = (unsigned)(coop_phase & 1) * (unsigned)(TS_KS * 192 * TS_DIM * 4) + (unsigned)(frame_ok ? frame : 0) * 64u + 16u * (unsigned)lg;
        if (frame_ok) {
#pragma unroll
            for (int t = 0; t < 8; ++t)  // write-through (sc1) 16-byte stores: the bytes are in memory when the wait below returns
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tacc[t]), slab_rsrc, slab_off + (unsigned)part * (192 * TS_DIM * 4) + (192 * 64) * t, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also drains this wave's LDS-DMAs: the ring refills behind the exchange)
        __builtin_amdgcn_s_barrier();                      // every wave's stores have left
        TS_STAMP(10);  // partial stored
        unsigned* const cnt = p.coop_cnt + 2 * clip;
        if (tid == 0) {
            if (!(coop_phase == 0 && part == p.coop_fault_part))  // (test hook: this workgroup's first arrival is withheld)
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)TS_KS * (unsigned)(coop_phase + 1);
            // Bounded in TIME: were the six workgroups of a clip ever not co-resident (more cooperative launches at once than the
            // chip has CUs — launch_trans_stack's admission rules that out inside one process —, another process holding the CUs, a
            // CU mask, a debugger) the poll expires after coop_timeout_ticks; the expiry is counted where the host sees it and this
            // workgroup stops waiting, so the launch ends within one timeout — its numbers are invalid and the host is told.
            if (!coop_dead && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    __builtin_amdgcn_s_sleep(2);
                    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)p.coop_timeout_ticks) {
                        coop_dead = true;
                        __hip_atomic_fetch_add(p.coop_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        __builtin_amdgcn_s_barrier();                      // the poll has matched: all six partials of this phase are in memory
        TS_STAMP(11);  // arrival + wait for the other five
#pragma unroll
        for (int t = 0; t < 8; ++t) yacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (frame_ok) {
            // sc1 loads (never served from this CU's L1), TB tiles x six partials requested before the first add (the 12-wave form
            // has 168 registers): left to itself hipcc issued them one by one, each behind a full wait (40 round trips per exchange)
            constexpr int TB = MAXW > 8 ? 2 : 4;  // (8 at <= 8 waves spilled 16 registers and was 10 % slower)
#pragma unroll
            for (int tb = 0; tb < 8; tb += TB) {
                u32x4 v[TS_KS][TB];
#pragma unroll
                for (int k = 0; k < TS_KS; ++k)
#pragma unroll
                    for (int tt = 0; tt < TB; ++tt)
                        v[k][tt] = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, slab_off + (unsigned)k * (192 * TS_DIM * 4) + (192 * 64) * (tb + tt), 0, 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    f32x4_t sum = __builtin_bit_cast(f32x4_t, v[0][tt]);
#pragma unroll
                    for (int k = 1; k < TS_KS; ++k) sum = sum + __builtin_bit_cast(f32x4_t, v[k][tt]);  // ((((p0 + p1) + p2) + p3) + p4) + p5
                    yacc[tb + tt] = sum;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TS_STAMP_DRAIN();
        TS_STAMP(12);  // six partials read and added
        ++coop_phase;
    };
    // end of a sub-layer: x += yacc, back to its rows (the next sub-layer's end re-reads them), and into xr for the LayerNorm
    auto add_residual = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4_t xv = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (XPRE) {
                xv = xpre[t];
            } else {
                if (frame_ok) xv = *reinterpret_cast<const f32x4_t*>(xlane + XT * t);
            }
            xr[t] = xv + yacc[t];
            if (frame_ok) *reinterpret_cast<f32x4_t*>(xlane + XT * t) = xr[t];
        }
    };

    unsigned char* const k_mine = smem_ts + L::OFF_K + wave * TS_PIECE + 16 * lane;                               // + 1024 plane
    unsigned char* const v_mine = smem_ts + L::OFF_V + (wave >> 1) * 2 * TS_PIECE + 16 * lane + 8 * (wave & 1);  // + 3072 dt + 1024 plane
    const unsigned char* const k_lane = smem_ts + L::OFF_K + 16 * lane;
    const unsigned char* const v_lane = smem_ts + L::OFF_V + 16 * lane;
    const int key_steps = (wave >> 1) + 1;  // causal: keys 0 .. 16 wave + 15 in steps of 32
    const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int layer = 0; layer < p.n_layers; ++layer) {
        const float* const lnp = ln_s + layer * 4 * TS_DIM;
        // ================= LocalMHA ===================================================================================
        layer_norm_planes(lnp, lnp + TS_DIM);
        TS_STAMP(1);  // LayerNorm 1
#pragma unroll 1
        for (int h = COOP ? part : 0; h < (COOP ? part + 1 : TS_HEADS); ++h) {
#pragma unroll
            for (int t = 0; t < 8; ++t) tacc[t] = zero4;
            // ---- q^T, k^T (weights x activations) and V (activations x weights) of head h for this wave's frames ------
            f32x4_t qa[2] = {zero4, zero4}, ka[2] = {zero4, zero4}, va[2] = {zero4, zero4};
            slot_pair([&](auto dt_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                constexpr int dt = decltype(dt_)::value;
                qa[dt] = mfma6(f, ap[decltype(j_)::value], qa[dt]);
            }, nothing);
            bf16x8 qp[3];
            {
                f32x4_t q0 = qa[0] * p.scale, q1 = qa[1] * p.scale;  // q pre-scaled (local_attention: q = q * scale)
                planes_of(q0, q1, qp);
            }
            slot_pair([&](auto dt_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                constexpr int dt = decltype(dt_)::value;
                ka[dt] = mfma6(f, ap[decltype(j_)::value], ka[dt]);
            }, [&](auto dt_) __attribute__((always_inline)) {
                if constexpr (decltype(dt_)::value == 1) {  // K of this wave's 16 keys as S^T's A-operand fragment
                    bf16x8 kp[3];
                    planes_of(ka[0], ka[1], kp);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(k_mine + 1024 * pl) = kp[pl];
                }
            });
            // operands swapped: va[dt][i] = V[frame 16 wave + 4 g + i][d = 16 dt + fl]
            slot_pair([&](auto dt_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                constexpr int dt = decltype(dt_)::value;
                va[dt] = mfma6(ap[decltype(j_)::value], f, va[dt]);
            }, [&](auto dt_) __attribute__((always_inline)) {  // half (keys of this wave) of the V^T fragment of key step wave / 2, d tile dt
                constexpr int dt = decltype(dt_)::value;
                unsigned w0[3], w1[3];
                split2(va[dt][0], va[dt][1], w0[0], w0[1], w0[2]);
                split2(va[dt][2], va[dt][3], w1[0], w1[1], w1[2]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(v_mine + TS_PIECE * dt + 1024 * pl) = make_uint2(w0[pl], w1[pl]);
            });  // (the pair's last barrier publishes K and V of every wave)
            TS_STAMP(2);  // q, k, v products (6 slots)
            // ---- causal attention of this wave's 16 queries over keys 0 .. 16 wave + 15 ------------------------------
            float m_run = -INFINITY, l_run = 0.f;
            f32x4_t oa[2] = {zero4, zero4};
            const float* const bias_h = bias_s + h * TS_MAX_FRAMES;
#pragma unroll 1
            for (int ks = 0; ks < key_steps; ++ks) {
                f32x4_t st[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 kf[3];
                    load_frag(kf, k_lane + (2 * ks + u) * TS_PIECE);
                    st[u] = mfma6(kf, qp, zero4);  // S^T[key 32 ks + 16 u + 4 g + i][query fl]
                }
                float mx = -INFINITY;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = 32 * ks + 16 * u + 4 * lg + i;
                        const bool vis = key <= frame;
                        const float sv = vis ? st[u][i] + bias_h[vis ? frame - key : 0] : -INFINITY;
                        st[u][i] = sv;
                        mx = fmaxf(mx, sv);
                    }
                mx = rows_max(mx);
                const float m_new = fmaxf(m_run, mx);  // finite from the first step on: key 0 is visible to every query
                const float alpha = ts_exp_neg(m_run - m_new);
                float psum = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float pv = ts_exp_neg(st[u][i] - m_new);  // masked entries: exp(-inf) = 0
                        st[u][i] = pv;
                        psum += pv;
                    }
                l_run = l_run * alpha + psum;
                m_run = m_new;
                bf16x8 pp[3];
                planes_of(st[0], st[1], pp);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x8 vf[3];
                    load_frag(vf, v_lane + (2 * ks + dt) * TS_PIECE);
                    oa[dt] = mfma6(vf, pp, oa[dt] * alpha);  // O^T[d = 16 dt + 4 g + i][query fl]
                }
            }
            const float l_tot = rows_sum(l_run);
            const float inv = 1.0f / l_tot;
            bf16x8 op[3];
            {
                f32x4_t o0 = oa[0] * inv, o1 = oa[1] * inv;
                planes_of(o0, o1, op);
            }
            TS_STAMP(3);  // attention
            // ---- out projection of this head's 32 columns: partial h of the sub-layer's output ------------------------------
            slot_pair([&](auto half_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                constexpr int rt = 4 * decltype(half_)::value + decltype(j_)::value;
                tacc[rt] = mfma6(f, op, tacc[rt]);
            }, nothing);
            if (!COOP) add_partial(h);
            TS_STAMP(4);  // out projection (2 slots)
        }
        if (COOP) coop_combine();
        add_residual();
        // ================= FeedForward (GEGLU) ========================================================================
        layer_norm_planes(lnp + 2 * TS_DIM, lnp + 3 * TS_DIM);
        TS_STAMP(5);  // residual + LayerNorm 2
#pragma unroll 1
        for (int c = COOP ? 2 * part : 0; c < (COOP ? 2 * part + my_chunks : TS_FF_CHUNKS); ++c) {
            if ((c & 1) == 0) {  // a partial = a pair of hidden chunks (2 i, 2 i + 1), accumulated from zero
#pragma unroll
                for (int t = 0; t < 8; ++t) tacc[t] = zero4;
            }
            f32x4_t vg[4] = {zero4, zero4, zero4, zero4};  // value tiles 0, 1 then gate tiles 0, 1 of hidden units 32 c .. 32 c + 31
            ring_static_for<2>([&](auto w_) {  // value tiles 0, 1, then gate tiles 0, 1
                slot_pair([&](auto t_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                    constexpr int u = 2 * decltype(w_)::value + decltype(t_)::value;
                    vg[u] = mfma6(f, ap[decltype(j_)::value], vg[u]);
                }, nothing);
            });
            TS_STAMP(6);  // FF-in products (4 slots)
            bf16x8 hp[3];
            {
                f32x4_t h0, h1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h0[i] = vg[0][i] * gelu_erf(vg[2][i]);  // GEGLU: value * gelu(gate), the first half is the value
                    h1[i] = vg[1][i] * gelu_erf(vg[3][i]);
                }
                planes_of(h0, h1, hp);
            }
            TS_STAMP(7);  // GEGLU
            slot_pair([&](auto half_, auto j_, const bf16x8 (&f)[3]) __attribute__((always_inline)) {
                constexpr int rt = 4 * decltype(half_)::value + decltype(j_)::value;
                tacc[rt] = mfma6(f, hp, tacc[rt]);
            }, nothing);
            if (!COOP && ((c & 1) || c == TS_FF_CHUNKS - 1)) add_partial(c >> 1);
            TS_STAMP(8);  // FF-out products (2 slots)
        }
        if (COOP) coop_combine();
        add_residual();  // (also the next layer's LayerNorm input, already in xr)
        TS_STAMP(9);  // residual
    }
    if (COOP) {
        // every workgroup holds the same final stream; workgroup 0 writes it to the tensor.  The last workgroup to get here zeroes
        // the clip's counters for the next launch (the others have all passed their last poll: they counted themselves done after it)
        if (part == 0 && frame_ok) {
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4_t*>(xtrue + 16 * t) = xr[t];
        }
        if (tid == 0) {
            unsigned* const cnt = p.coop_cnt + 2 * clip;
            if (__hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)TS_KS - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // leave no LDS-DMA in flight behind the workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace


bool trans_stack_supported(int dim, int dim_head, int heads, int ff_inner, int frames, int window, int n_layers) {
    return dim == TS_DIM && dim_head == TS_DH && heads == TS_HEADS && ff_inner == TS_FFI && frames >= 1 && frames <= TS_MAX_FRAMES &&
           frames <= window && n_layers >= 1 && n_layers <= TS_MAX_LAYERS;
}

// The weight stream of one layer in consumption order (host, at model-build time).  wqkv [576][128], wout [128][192] as the
// reference stores them; wff1 / wff2 in the re-laid forms of network.hip (value / gate 32-row tiles interleaved, [ff_n][128];
// [128][ff_pad] zero padded along k).
void trans_stack_layer_image(std::vector<unsigned char>& img, const float* wqkv, const float* wout, const float* wff1, int ff_n,
                             const float* wff2, int ff_pad) {
    for (int h = 0; h < TS_HEADS; ++h) {
        for (int part = 0; part < 3; ++part)
            for (int dt = 0; dt < 2; ++dt)
                for (int s = 0; s < 4; ++s) ring_put_piece(img, wqkv, TS_DIM, 3 * TS_INNER, TS_DIM, part * TS_INNER + h * TS_DH + 16 * dt, 32 * s);
        for (int rt = 0; rt < 8; ++rt) ring_put_piece(img, wout, TS_INNER, TS_DIM, TS_INNER, 16 * rt, 32 * h);
    }
    for (int c = 0; c < TS_FF_CHUNKS; ++c) {
        for (int half = 0; half < 2; ++half)  // value rows 64 c + r, gate rows 64 c + 32 + r of the interleaved image
            for (int t = 0; t < 2; ++t)
                for (int s = 0; s < 4; ++s) ring_put_piece(img, wff1, TS_DIM, ff_n, TS_DIM, 64 * c + 32 * half + 16 * t, 32 * s);
        for (int rt = 0; rt < 8; ++rt) ring_put_piece(img, wff2, ff_pad, TS_DIM, ff_pad, 16 * rt, 32 * c);
    }
}
int64_t trans_stack_layer_image_bytes() { return (int64_t)TS_SLOTS_PER_LAYER * TS_SLOT; }

size_t trans_stack_coop_bytes() {  // scratch of the cooperative form: slabs + private streams of TS_COOP_CLIPS clip slots, then the counters
    return (size_t)TS_COOP_CLIPS * (size_t)(TS_COOP_SLAB_FLOATS + TS_COOP_X_FLOATS) * sizeof(float) + 2 * TS_COOP_CLIPS * sizeof(unsigned);
}
int trans_stack_coop_max_batch() { return TS_COOP_CLIPS; }
size_t trans_stack_coop_counter_offset() { return (size_t)TS_COOP_CLIPS * (size_t)(TS_COOP_SLAB_FLOATS + TS_COOP_X_FLOATS) * sizeof(float); }
size_t trans_stack_coop_counter_bytes() { return 2 * TS_COOP_CLIPS * sizeof(unsigned); }

// ---- admission of cooperative launches ---------------------------------------------------------------------------------------
// The six workgroups of a clip wait for each other, so every workgroup of every cooperative launch that can be on the device at
// one time must be able to hold a CU of its own.  Launches of ONE context are ordered (one stream, or the workspace event), but two
// contexts on two streams — or two graphs — run side by side, and nothing at enqueue time says which.  So each context CLAIMS CUs
// in a process-wide per-device registry: a launch of `batch` clips needs TS_KS * batch of them (clip slots beyond the batch return
// at once); a context's claim only ever grows (a graph captured from it may replay the launch at any later time) and is returned when
// the context is destroyed.  A launch whose claim does not fit the device's CU count any more runs in the one-workgroup form — same
// bits.  Other PROCESSES on the device are outside this registry: they need trans_coop = 0 (include/l3ac_hip.h), and the kernel's
// timed poll reports it if they do not.
static std::atomic<int> g_coop_claimed[L3AC_MAX_DEVICES] = {};

bool trans_coop_admit(TransCoopState& st, int batch) {
    const int slot = l3ac_device_slot();
    if (slot < 0) return false;
    const int need = TS_KS * batch;
    if (need <= st.claim) return true;
    const int more = need - st.claim, cus = l3ac_device_cu_count();
    int cur = g_coop_claimed[slot].load(std::memory_order_relaxed);
    do {
        if (cur + more > cus) return false;
    } while (!g_coop_claimed[slot].compare_exchange_weak(cur, cur + more, std::memory_order_relaxed));
    st.claim = need;
    st.claim_slot = slot;
    return true;
}
void trans_coop_release(TransCoopState& st) {
    if (st.claim > 0 && st.claim_slot >= 0) g_coop_claimed[st.claim_slot].fetch_sub(st.claim, std::memory_order_relaxed);
    st.claim = 0;
}
int trans_coop_claimed_on_device(int device) {
    return device < 0 || device >= L3AC_MAX_DEVICES ? -1 : g_coop_claimed[device].load(std::memory_order_relaxed);
}

template <int MAXW>
static void launch_ts(hipStream_t s, bool coop, int batch, int waves, int n_layers, const TransStackArgs& a) {
    const unsigned threads = 64 * (waves + (TsLds<MAXW>::LOADER ? TsLds<MAXW>::NDW : 0));  // (+ the loader waves)
    if (coop)
        hipLaunchKernelGGL((trans_stack_kernel<MAXW, TS_KS>), dim3(TS_KS * a.coop_slots), dim3(threads), TsLds<MAXW>::bytes(n_layers), s, a);
    else
        hipLaunchKernelGGL((trans_stack_kernel<MAXW, 1>), dim3((unsigned)batch), dim3(threads), TsLds<MAXW>::bytes(n_layers), s, a);
}

// `coop`: the context's cooperative-form state (scratch of trans_stack_coop_bytes() bytes whose counters — the last 2 * TS_COOP_CLIPS
// words — were zeroed when it was allocated, the host-visible failure word, the claim), or null.  With it, batches of at most
// TS_COOP_CLIPS clips that are admitted (above) run in the cooperative form (six workgroups per clip).
int launch_trans_stack(hipStream_t s, const LocalTransW& w, float* x, int batch, int frames, float scale, TransCoopState* coop) {
    const int n_layers = (int)w.layers.size();
    L3AC_REQUIRE(w.stack_img && w.stack_ln && batch > 0 && frames >= 1 && frames <= TS_MAX_FRAMES && frames <= w.window &&
                     n_layers >= 1 && n_layers <= TS_MAX_LAYERS,
                 "trans_stack: bad arguments (frames=%d window=%d layers=%d)", frames, w.window, n_layers);
    static PerDeviceOnce configured;
    if (configured.first()) {
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<4>::bytes(TS_MAX_LAYERS)));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<8>::bytes(TS_MAX_LAYERS)));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<12, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<12>::bytes(TS_MAX_LAYERS)));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<4, TS_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<4>::bytes(TS_MAX_LAYERS)));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<8, TS_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<8>::bytes(TS_MAX_LAYERS)));
        L3AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trans_stack_kernel<12, TS_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, TsLds<12>::bytes(TS_MAX_LAYERS)));
        configured.done();
    }
    int waves = 2 * (int)ceil_div64(frames, 32);  // even: every key tile a wave reads in pairs has been written by some wave
    if (waves < 4) waves = 4;                      // (the 4-wave instantiation adds its loader wave at the launch)
    // the cooperative form needs its TS_KS workgroups per clip co-resident (they wait for each other), one workgroup per CU
    const bool use_coop = coop != nullptr && coop->enabled && coop->scratch != nullptr && coop->fail_dev != nullptr &&
                          batch <= TS_COOP_CLIPS && trans_coop_admit(*coop, batch);
    TransStackArgs a{};
    a.x = x; a.frames = frames; a.n_layers = n_layers; a.img = w.stack_img; a.ln = w.stack_ln; a.bias_table = w.bias_table;
    a.table_stride = 2 * w.window; a.scale = scale; a.batch = batch; a.coop_slots = (batch + 7) / 8 * 8;
    a.coop_fault_part = -1;
    if (use_coop) {
        a.coop_slab = reinterpret_cast<float*>(coop->scratch);
        a.coop_x = a.coop_slab + (int64_t)TS_COOP_CLIPS * TS_COOP_SLAB_FLOATS;
        a.coop_cnt = reinterpret_cast<unsigned*>(a.coop_x + (int64_t)TS_COOP_CLIPS * TS_COOP_X_FLOATS);
        a.coop_fail = coop->fail_dev;
        const long long ticks = (long long)(coop->timeout_ms > 0 ? coop->timeout_ms : 1) * 100000LL;  // s_memrealtime: 100 MHz
        a.coop_timeout_ticks = (unsigned)(ticks > 0xffffffffLL ? 0xffffffffLL : ticks);
        a.coop_fault_part = coop->fault_part;
    }
    const double rows = (double)batch * frames;
    const double lin = 2.0 * (3.0 * TS_INNER * TS_DIM + TS_DIM * TS_INNER + 3.0 * TS_FFI * TS_DIM);
    const double att = 2.0 * 2.0 * TS_INNER * 0.5 * (frames + 1.0);
    char name[64];
    std::snprintf(name, sizeof(name), "trans_stack_kernel%s %dx%d L%d", use_coop ? "<coop>" : "", batch, frames, n_layers);
    ProfScope prof(s, name, n_layers * rows * (lin + att), 2.0 * rows * TS_DIM * 4.0);
    // (the instantiation only changes the register budget and whether weight fragments are fetched one piece ahead: same bits)
    if (waves <= 4) launch_ts<4>(s, use_coop, batch, waves, n_layers, a);
    else if (waves <= 8) launch_ts<8>(s, use_coop, batch, waves, n_layers, a);
    else launch_ts<12>(s, use_coop, batch, waves, n_layers, a);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
