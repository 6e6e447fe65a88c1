// Host-side model of the L3AC network: device-resident weights laid out for the kernels, the workspace,
// and the encode / decode pipelines (reference call stacks: SURVEY.md §3.2, §3.3).
#pragma once

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.hpp"

struct ConvUnitW {  // modules.py:10-41
    int c = 0;
    const float *dw_w, *dw_b, *ln_w, *ln_b, *w1, *b1, *alpha, *inv_alpha, *gamma, *beta, *w2, *b2;
    // fragment-ordered bf16x3 images of w1 / w2 for conv_unit_split_kernel (narrow stages), null when not built
    const unsigned char *w1_img = nullptr, *w2_img = nullptr;
    // W1 / W2 as ONE fragment-ordered bf16x3 stream in consumption order for conv_unit_wide_kernel (wide stages)
    const unsigned char* wide_img = nullptr;
    // the same for conv_unit_ring_kernel (narrow stages, 16 frames per wave): pieces of ring_common.hpp per hidden pair
    const unsigned char* ring_img = nullptr;
};
struct DownW {  // modules.py:96-99 and local_trans.py:136: Conv1d(k = stride) [+ ChannelNorm]
    int cin = 0, cout = 0, stride = 1;
    const float *w, *b, *nw = nullptr, *nb = nullptr;
    // the conv's weight [cout][stride * cin] as bf16x3 pieces for the DOWN form of up_fused_kernel, null when the geometry is not the kernel's
    const unsigned char* fused_img = nullptr;
    // ... and as fp32 in fragment order for down_exact_kernel (the default one-kernel form: the unfused route's bits)
    const unsigned char* exact_img = nullptr;
};
struct ConvK3W {  // modules.py:110, :150
    int cin = 0, cout = 0;
    const float *w, *b;
};
struct EnhW {  // tconv/__init__.py:30-44
    int c = 0;
    EnhanceW t;
    const float *in_w, *in_b, *gate_w, *gate_b;
};
struct UpW {  // modules.py:160-164
    int cin = 0, cout = 0, scale = 1;
    const float *w, *b, *nw, *nb;
    // the 1x1 conv's weight as bf16x3 pieces for up_fused_kernel (kernels/up_fused.hip), null when the geometry is not the kernel's
    const unsigned char* fused_img = nullptr;
};
struct LegacyW {  // modules.py:47-64
    int c = 0, dil = 1;
    const float *a0, *ia0, *w1, *b1, *a1, *ia1, *w2, *b2;
    // bf16x3 fragment images of w1 / w2 for the split kernel (kernels/last_block.hip), null when not built
    const unsigned char *w1_img = nullptr, *w2_img = nullptr;
};
struct HeadW {  // modules.py:192-194
    int c = 0;
    const float *alpha, *inv_alpha, *w, *b;
};
struct TransLayerW {  // local_attention LocalMHA + FeedForward
    const float *ln1w, *ln1b, *wqkv, *wout, *ln2w, *ln2b, *wff1, *wff2;
};
struct LocalTransW {  // local_trans.py:7-53
    int window = 0;
    std::vector<TransLayerW> layers;
    const float* bias_table = nullptr;  // [heads][2 * window]
    // the whole stack for trans_stack_kernel (kernels/trans_stack.hip): the layers' weights as one fragment-ordered bf16x3 stream
    // in consumption order, and their LayerNorm parameters [layers][4][dim]; null when the geometry is not the kernel's
    const unsigned char* stack_img = nullptr;
    const float* stack_ln = nullptr;
};

// Cooperative form of trans_stack_kernel (kernels/trans_stack.hip): per-context state.
struct TransCoopState {
    void* scratch = nullptr;        // device: partial slabs, private residual streams, arrival counters (trans_stack_coop_bytes())
    unsigned* fail_host = nullptr;  // pinned, device-visible host word: arrival polls that expired (cumulative); the kernel adds to it
    unsigned* fail_dev = nullptr;   // its device address
    int enabled = 1;                // option "trans_coop"; cleared for good once a timeout has been seen
    int claim = 0, claim_slot = -1; // CUs of the device this context's cooperative launches may occupy (process-wide registry)
    int timeout_ms = 250;           // option "coop_timeout_ms": how long an arrival poll waits
    int fault_part = -1;            // option "coop_test_fault" (test hook): workgroup that withholds its first arrival, -1 = none
    unsigned seen = 0;              // value of *fail_host the host has already acted on (fallback + report)
    unsigned count_base = 0;        // value of *fail_host at the last l3ac_coop_timeout_count(reset = 1)
    unsigned calls_since_check = 0; // workspace-using calls entered since the last SYNCHRONISING check that found nothing (bounds which
                                    // calls an L3AC_ECOOP report can refer to: the failure word is read at enqueue time, without a sync)
};

struct Workspace {
    float *x0 = nullptr, *x1 = nullptr, *a = nullptr, *h = nullptr, *yi = nullptr, *stats = nullptr, *sumsq = nullptr;
    size_t x_cap = 0, a_cap = 0, h_cap = 0, yi_cap = 0, b_cap = 0;  // capacities in floats
    size_t bytes() const { return (2 * x_cap + a_cap + h_cap + yi_cap + 9 * b_cap) * sizeof(float); }
};

struct l3ac_ctx {
    l3ac_config cfg{};
    int device = 0;
    int hop = 0, enc_rate = 1, dim_head = 0, inner = 0, ff_inner = 0, ff_pad = 0, ff_n = 0;
    float* arena = nullptr;
    size_t arena_floats = 0;
    // bf16x3 split images of the GEMM weights (kernels/gemm_split.hip), keyed by the fp32 weight's device pointer
    unsigned char* img_arena = nullptr;
    size_t img_bytes = 0;
    std::unordered_map<const float*, const unsigned char*> split_img;
    // Route switches are PER CONTEXT (a flip never reaches another context's calls or captured graphs).  gemm_split: the large
    // channel contractions on the bf16 matrix cores through exact bf16x3 operand splits (default) or everything on the fp32 MFMA
    // instruction; head_pretanh (validation): the output head stores its value before the final tanh.
    bool gemm_split = true, head_pretanh = false;
    // which fused kernel takes the narrow ConvUnits (C <= 48) on the split route: conv_unit_ring_kernel (16 frames per wave, LDS-DMA
    // weight ring) or conv_unit_split_kernel (32 frames per wave, chunk barriers); l3ac_ctx_set_option(ctx, "narrow_ring", 0 / 1)
    // the wide ConvUnits (C = 96 .. 256) of FEW frames — a streaming chunk — as two launches over (frame tiles x channel slices) instead of
    // the fused kernel, whose waves own their frames end to end (conv_unit_wide.hip, 'the SLICED form'; the same bits)
    // batch kernels that keep two workgroups per CU resident (conv_unit_wide_kernel<96>) hand their units out by a counter instead of equal
    // static shares: the workgroup dispatched first is served first by every SIMD and finishes its share early (option "unit_counter")
    int unit_counter = 1;
    int wide_sliced = 1;  // 0: never, 1: where it is faster (up to 256 frame tiles of 16: measured), 2: wherever the form exists (the same today; tests)
    int narrow_ring = 1;  // 0: conv_unit_split_kernel everywhere, 1: the ring kernel where it is faster (C = 48), 2: wherever it exists (C = 24 too)
    // encoder down layers 24 -> 48 and 48 -> 96 (Conv1d(k = stride) + ChannelNorm) in one kernel on the bf16x3 route (the DOWN form of
    // up_fused_kernel) instead of a small-N fp32-MFMA GEMM + row kernel: option "down_fused" / env L3AC_DOWN_FUSED.  Default 0: it is
    // 0.12 ms faster at 256 clips and as accurate, but a different rounding of the encoder's first layers, and of the tokens compared
    // with the oracle so far one (stress weights, 1.9e-6 of a rounding boundary) changes sides with it — DESIGN.md section 4.
    // Round 6: value 2 (the DEFAULT) = down_exact_kernel, the same fusion with the unfused route's arithmetic bit for bit (exact fp32 MFMA
    // chain in gemm_f32_kernel's k order, row_kernel's ChannelNorm tree) on BOTH GEMM routes; 1 = the bf16x3 form; 0 = GEMM + row kernel.
    int down_fused = 2;
    const unsigned char* img(const float* w) const {  // null on the exact route: launch_gemm then takes the fp32 kernel
        if (!gemm_split) return nullptr;
        auto it = split_img.find(w);
        return it == split_img.end() ? nullptr : it->second;
    }

    FirstBlockW first{};
    std::vector<std::vector<ConvUnitW>> enc_units;  // per encoder stage (incl. the tail stage)
    std::vector<DownW> enc_down;
    ConvK3W enc_out{};
    std::vector<LocalTransW> en_enc;   // execution order
    DownW en_down{};                   // DownTrans.down_layer (compressed configs)
    std::vector<LocalTransW> en_dec;
    const float *q_win = nullptr, *q_bin = nullptr, *q_wout = nullptr, *q_bout = nullptr;
    ConvK3W dec_in{};
    std::vector<std::vector<ConvUnitW>> dec_units;
    std::vector<EnhW> dec_enh;
    std::vector<UpW> dec_up;
    std::vector<LegacyW> legacy;
    HeadW head{};

    // name -> block, for the l3ac_op_* parity entry points
    std::map<std::string, const ConvUnitW*> by_unit;
    std::map<std::string, const DownW*> by_down;
    std::map<std::string, const ConvK3W*> by_k3;
    std::map<std::string, const EnhW*> by_enh;
    std::map<std::string, const UpW*> by_up;
    std::map<std::string, const LocalTransW*> by_trans;

    Workspace ws;
    // Cross-stream ordering of the (single, in-place) workspace: every call that touches `ws` records `ws_done` on its stream
    // when it has enqueued its last kernel, and a later call on a DIFFERENT stream first makes that stream wait for it.
    // scratch of trans_stack_kernel's cooperative form (partial slabs, private residual streams, arrival counters); option
    // "trans_coop" (default 1) switches the form off without freeing it
    TransCoopState coop;
    int* bad_index_count = nullptr;  // device: indices outside [0, codebook size) seen by l3ac_decode since the last reset
    int* wide_counters = nullptr;    // device, 64 B, zeroed: conv_unit_wide_kernel's unit counters (every launch leaves them zeroed)
    float* grn_min_sumsq = nullptr;  // device: smallest per-clip sum of squares any GRN of this context has seen (grn_exact only)
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_done_valid = false;
};

int network_build(l3ac_ctx* ctx, const l3ac_tensor* tensors, int n_tensors);
void network_free(l3ac_ctx* ctx);
int workspace_ensure(l3ac_ctx* ctx, size_t x_floats, size_t a_floats, size_t h_floats, size_t yi_floats, size_t batch,
                     hipStream_t s);
int workspace_ensure_clip(l3ac_ctx* ctx, int batch, int samples, hipStream_t s);

// fused ConvUnit for the narrow stages (kernels/conv_unit_fused.hip); x must not alias y
bool conv_unit_fused_supported(int c);
int launch_conv_unit_fused(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames, bool split, int ring = 0);
// third form of the narrow ConvUnit (kernels/conv_unit_ring.hip): 16 frames per wave, weights through an LDS-DMA ring
bool conv_unit_ring_supported(int c);
bool conv_unit_ring_preferred(int c);  // the widths the pipeline routes there by default
int launch_conv_unit_ring(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames);
std::vector<unsigned char> conv_unit_ring_image(const float* w1, const float* w2, int c);  // w1 [4c][c], w2 [c][4c]
// bf16x3 variant (kernels/conv_unit_split.hip), chosen by launch_conv_unit_fused when the images exist and the split route is on
int launch_conv_unit_split(hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames);
std::vector<unsigned char> conv_unit_w1_image(const float* w1, int c);  // w1 [4c][c]
std::vector<unsigned char> conv_unit_w2_image(const float* w2, int c);  // w2 [c][4c]
// fused ConvUnit of the wide stages (kernels/conv_unit_wide.hip): hidden tensor in registers, weights streamed through an LDS ring
bool conv_unit_wide_supported(int c);
size_t conv_unit_wide_scratch_bytes(int c, int64_t rows);
int launch_conv_unit_wide(hipStream_t s, const ConvUnitW& w, const float* x, float* y, unsigned char* planes, size_t planes_bytes, int batch,
                          int frames, int sliced_mode, int* counters);
std::vector<unsigned char> conv_unit_wide_image(const float* w1, const float* w2, int c);  // w1 [4c][c], w2 [c][4c]
// one LocalTrans stack per launch, one workgroup per clip (kernels/trans_stack.hip); x [batch][frames][128] in place
bool trans_stack_supported(int dim, int dim_head, int heads, int ff_inner, int frames, int window, int n_layers);
void trans_stack_layer_image(std::vector<unsigned char>& img, const float* wqkv, const float* wout, const float* wff1, int ff_n,
                             const float* wff2, int ff_pad);
int64_t trans_stack_layer_image_bytes();
// `coop` (the context's TransCoopState; null = never): batches of at most trans_stack_coop_max_batch() clips that the per-device
// admission lets in run in the cooperative form, six workgroups per clip — same bits as the one-workgroup form
int launch_trans_stack(hipStream_t s, const LocalTransW& w, float* x, int batch, int frames, float scale, TransCoopState* coop = nullptr);
size_t trans_stack_coop_bytes();
size_t trans_stack_coop_counter_offset();  // the arrival counters inside the scratch: [offset, offset + bytes)
size_t trans_stack_coop_counter_bytes();
int trans_stack_coop_max_batch();
bool trans_coop_admit(TransCoopState& st, int batch);
void trans_coop_release(TransCoopState& st);
int trans_coop_claimed_on_device(int device);  // -1: bad ordinal
// EnhanceBlock gate + 1x1 conv + linear upsample + ChannelNorm of the narrow decoder stages in one kernel (kernels/up_fused.hip)
bool up_fused_supported(int cin, int cout);
std::vector<unsigned char> up_fused_image(const float* w, int cin, int cout);  // w [cout][cin]
int launch_up_fused(hipStream_t s, const EnhW& e, const UpW& w, const float* x, const float* yi, const float* stats, float* y, int batch, int frames);
// encoder down layer (Conv1d(k = stride) + ChannelNorm) in the same kernel's DOWN form
bool down_fused_supported(int cin, int stride, int cout);
int launch_down_fused(hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames_out);
std::vector<unsigned char> down_exact_image(const float* w, int k, int cout);
int launch_down_exact(hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames_out);
// fused LegacyUnit / head (kernels/last_block.hip); x must not alias y
bool last_block_fused_supported(int c, int max_dil);
// host builders of the LegacyUnit weight images: w1 [c][7][c] (tap-major rows), w2 [c][c]
std::vector<unsigned char> legacy_w1_image(const float* w1, int c);
std::vector<unsigned char> legacy_w2_image(const float* w2, int c);
int launch_legacy_unit_fused(hipStream_t s, const LegacyW& w, const float* x, float* y, int batch, int frames, bool split, int* counters);
int launch_head_fused(hipStream_t s, const HeadW& w, const float* x, int batch, int frames, float* audio, bool pretanh);
// one ConvUnit of a stage on the ping-pong buffers: fused kernel (result in *alt, buffers swapped) or in place
int conv_unit_step(l3ac_ctx* ctx, hipStream_t s, const ConvUnitW& w, float** cur, float** alt, int batch, int frames);
// all ConvUnits of one stage (wide units: clip groups outside the units, see network.hip)
int run_conv_units(l3ac_ctx* ctx, hipStream_t s, const std::vector<ConvUnitW>& units, float** cur, float** alt, int batch, int frames);

// blocks (x may alias y where noted)
int run_conv_unit(l3ac_ctx* ctx, hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames);  // x == y ok
int run_down(l3ac_ctx* ctx, hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames);
int run_conv_k3(l3ac_ctx* ctx, hipStream_t s, const ConvK3W& w, const float* x, float* y, int batch, int frames);
int run_enhance(l3ac_ctx* ctx, hipStream_t s, const EnhW& w, const float* x, float* y, int batch, int frames);  // x == y ok
int run_up(l3ac_ctx* ctx, hipStream_t s, const UpW& w, const float* x, float* tmp, float* y, int batch, int frames);
int run_enhance_up(l3ac_ctx* ctx, hipStream_t s, const EnhW& e, const UpW& w, float* x, float* tmp, float* y, int batch, int frames);
int run_last_block(l3ac_ctx* ctx, hipStream_t s, float* x, float* audio, int batch, int frames);  // x is clobbered
int run_local_trans(l3ac_ctx* ctx, hipStream_t s, const LocalTransW& w, float* x, int batch, int frames);  // in place

// sub-modules; `cur`/`alt` are the ping-pong activation buffers, on return *cur holds the result
int run_encoder(l3ac_ctx* ctx, hipStream_t s, const float* audio, int64_t audio_stride, int batch, int samples,
                int frames, float** cur, float** alt);
int run_en_encoder(l3ac_ctx* ctx, hipStream_t s, int batch, int frames, float** cur, float** alt, int* n_tok);
int run_en_decoder(l3ac_ctx* ctx, hipStream_t s, int batch, int n_tok, float** cur, float** alt, int* frames);
int run_decoder(l3ac_ctx* ctx, hipStream_t s, int batch, int frames, float** cur, float** alt, float* audio);
