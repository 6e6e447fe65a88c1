/*
 * l3ac_hip.h — C ABI of the MI355X-native L3AC encode -> quantize -> decode path (libl3ac_hip.so).
 *
 * The reference (zhai-lw/L3AC) is pure Python and exposes no FFI; its boundary for this path is the method
 * surface of l3ac/__init__.py.  Each entry point below names the reference interface it replaces:
 *
 *   l3ac_create / l3ac_destroy   <- l3ac.get_model + network.load_model   (l3ac/__init__.py:21-25, :104-106,
 *                                                                          l3ac/xtract/nn/module.py:43-54)
 *   l3ac_encode                  <- L3AC.encode_audio                      (l3ac/__init__.py:108-114)
 *   l3ac_decode                  <- L3AC.decode_audio                      (l3ac/__init__.py:116-121)
 *   l3ac_fsq_* / l3ac_vq_argmin  <- VQEmbed.forward / to_features          (l3ac/vq/__init__.py:20-30, vq/fsq.py:30-81)
 *   l3ac_op_*                    <- the individual blocks of modules.py / tconv / local_trans.py, exported so
 *                                   that every kernel can be parity-tested alone.
 *
 * Conventions
 *   - every function returns 0 on success, a negative L3AC_E* code otherwise; nothing throws across the ABI;
 *     l3ac_last_error() returns a human-readable message for the calling thread's last failure.
 *   - all data pointers are DEVICE pointers (e.g. torch tensor.data_ptr()), contiguous, 16-byte aligned,
 *     fp32 / int32.  Activations are frame-major: [batch][frame][channel], channel fastest.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); calls only enqueue
 *     work on it: no host synchronisation, and no allocation once l3ac_reserve() has sized the workspace,
 *     so a call sequence can be captured into a hipGraph.
 *   - one context per device; a context is not thread-safe (one host thread at a time).
 *   - a context owns ONE workspace that every encode / decode / op call reuses in place.  Calls may be issued on
 *     different streams: each call makes its stream wait for the completion event of the context's previous call
 *     when that ran on another stream, so pipelined callers (encode of batch n+1 on stream A, decode of batch n on
 *     stream B) are serialised on the device instead of corrupting each other.  Two exceptions, both the caller's
 *     to order: work captured into a hipGraph (no events are recorded or waited for during capture; keep a captured
 *     sequence on one stream) and replays of such a graph.  For true overlap use one context per stream.
 */
#ifndef L3AC_HIP_H
#define L3AC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L3AC_ABI_VERSION 5
#define L3AC_MAX_STAGES 8
#define L3AC_MAX_LEVELS 8

enum {
    L3AC_OK = 0,
    L3AC_EINVAL = -1,    /* bad argument / unsupported geometry */
    L3AC_EWEIGHT = -2,   /* missing / mis-shaped weight tensor */
    L3AC_EHIP = -3,      /* HIP runtime error */
    L3AC_ENOMEM = -4,    /* workspace too small while stream capture forbids growing it */
    L3AC_ECOOP = -5,     /* an EARLIER call's cooperative transformer launch timed out: that call's outputs are invalid
                            (see l3ac_coop_timeout_count).  Reported once, by the first call ENTERED after the failure has been
                            counted — the failure word is read at enqueue time without synchronising, so on an asynchronous stream
                            that is one or more calls after the failing one; the message bounds the suspects.  Only a
                            synchronising check (l3ac_coop_timeout_pending / _count, validate=True in the Python surface) is
                            authoritative for the calls issued so far */
};

/* Network geometry: the [network_config] table of the reference's TOML files
 * (l3ac/codec.py:13-36, l3ac/en_codec.py:9-19). */
typedef struct l3ac_config {
    int32_t abi_version;                     /* = L3AC_ABI_VERSION */
    int32_t feature_dim;
    int32_t n_enc;                           /* len(encoder_dims) */
    int32_t enc_dims[L3AC_MAX_STAGES];
    int32_t enc_depths[L3AC_MAX_STAGES];
    int32_t compress_rates[L3AC_MAX_STAGES]; /* n_enc - 1 entries */
    int32_t n_dec;                           /* len(decoder_dims) */
    int32_t dec_dims[L3AC_MAX_STAGES];
    int32_t dec_depths[L3AC_MAX_STAGES];
    int32_t decode_rates[L3AC_MAX_STAGES];   /* n_dec - 1 entries */
    int32_t n_levels;
    int32_t levels[L3AC_MAX_LEVELS];         /* vq_config.levels */
    int32_t en_coder_depth;
    int32_t en_coder_window_size;
    int32_t en_coder_compress_rate;
    int32_t grn_exact;                       /* 0: GRN normaliser taken as exactly 1.0f (true for ||x|| >= 0.25);
                                                1: evaluate g / (g + eps) per clip (layers.py:112-115) */
} l3ac_config;

/* One named fp32 host tensor.  Names are the reference's state-dict keys prefixed with the module file name,
 * weight-norm already folded (".parametrizations.weight.original{0,1}" -> ".weight"), e.g.
 * "encoder.blocks.1.0.module.pw_conv1.weight". */
typedef struct l3ac_tensor {
    const char* name;
    const float* data;
    int64_t numel;
} l3ac_tensor;

typedef struct l3ac_ctx l3ac_ctx;

const char* l3ac_last_error(void);
int l3ac_abi_version(void);

int l3ac_create(const l3ac_config* cfg, const l3ac_tensor* tensors, int32_t n_tensors, int32_t device,
                l3ac_ctx** out);
void l3ac_destroy(l3ac_ctx* ctx);

/* Size the workspace for clips of up to `samples` samples in batches of up to `batch` (allocates; call it
 * before stream capture).  encode/decode grow the workspace themselves when not capturing. */
int l3ac_reserve(l3ac_ctx* ctx, int32_t batch, int32_t samples);
int64_t l3ac_workspace_bytes(const l3ac_ctx* ctx);
/* decode_audio(indices = ...) takes its indices from outside (a wire stream): values outside [0, codebook size) are counted
 * per context and clamped into range instead of being decomposed into wrapped level indices.  This call synchronises the
 * device, writes the count since the last reset to *out and optionally resets it. */
int l3ac_bad_index_count(l3ac_ctx* ctx, int32_t reset, int64_t* out);
/* The cooperative form of the transformer kernel (option "trans_coop", below; reference l3ac/local_trans.py:42-48 — which cannot
 * return silently wrong tokens) gives each clip of a small batch six workgroups that wait for each other's partial results.  If they
 * are ever not co-resident (another PROCESS filling the device, a CU mask, a debugger) an arrival poll expires after
 * "coop_timeout_ms" (default 250 ms; every workgroup then stops waiting, so the launch ends within about one time limit), the kernel
 * counts it in host-visible memory and the outputs of that call are INVALID.  The host is told in two ways:
 *   - an encode / decode / op call ENTERED after the failing launch has run returns L3AC_ECOOP (once) before doing anything.  The
 *     entry check reads host memory and does not synchronise: calls enqueued on an asynchronous stream before the failing launch
 *     ran are not stopped (they ran the cooperative form too and are suspect), and a program that ends first is never told.  The
 *     error text says how many calls were entered since the last synchronising check: the invalid call is one of them;
 *   - AUTHORITATIVE: l3ac_coop_timeout_pending / l3ac_coop_timeout_count synchronise the device first.  _count writes the number
 *     of expired polls since the last reset to *out and acts on them (the report is then considered delivered: no L3AC_ECOOP
 *     follows); _pending writes the number NOT yet acted on and leaves the state alone — read before and after a call it attributes
 *     a failure to that call without hiding an earlier one (what validate=True of the Python surface does).
 * Either way the context then leaves the cooperative form (trans_coop = 0: same bits, one workgroup per clip) and its arrival
 * counters are zeroed again, so the repeated call is correct.  Inside ONE process cooperative launches cannot starve each other:
 * every context claims the CUs its launches need in a per-device registry and a launch that does not fit runs in the
 * one-workgroup form.  Other processes sharing the device must set trans_coop = 0 (env L3AC_TRANS_COOP=0).
 * A hipGraph captured from a context must not be replayed concurrently with itself or with other work of the same context
 * (the slabs and counters are per context, like the workspace). */
int l3ac_coop_timeout_count(l3ac_ctx* ctx, int32_t reset, int64_t* out);
int l3ac_coop_timeout_pending(l3ac_ctx* ctx, int64_t* out);
/* CUs of `device` that live contexts of this process have claimed for cooperative launches (the registry above); -1 for a bad
 * ordinal.  No device call. */
int32_t l3ac_coop_claimed_cus(int32_t device);
/* Guard of the GRN fast path (layers.py:112-115; l3ac_config.grn_exact).  A context created with grn_exact = 1 evaluates
 * g / (g + 1e-8) per clip and keeps the smallest per-clip norm g = ||x||_2 any of its GRN layers has seen; this call
 * synchronises the device, writes that minimum to *out (+inf if no GRN has run) and optionally resets it.  The fast path
 * (grn_exact = 0) is exact whenever the reported minimum is >= 0.25. */
int l3ac_grn_min_norm(l3ac_ctx* ctx, int32_t reset, float* out);
int32_t l3ac_hop_length(const l3ac_ctx* ctx);

/* encode_audio: audio [batch][samples] (row stride `audio_stride` floats) is right-padded with zeros to a
 * multiple of hop_length (codec.py:79-84); n_tok = ceil(samples / hop).
 *   q_feature      [batch][n_tok][feature_dim] fp32
 *   indices        [batch][n_tok]              int32
 *   level_indices  [batch][n_tok][n_levels]    fp32   (may be NULL) */
int l3ac_encode(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, int64_t audio_stride,
                float* q_feature, int32_t* indices, float* level_indices, void* stream);

/* decode_audio: from q_feature, or — when q_feature is NULL — from indices (vq/__init__.py:20-23).
 *   audio_out [batch][n_tok * hop_length] fp32, NOT trimmed (the caller slices, example.py:28). */
int l3ac_decode(l3ac_ctx* ctx, const float* q_feature, const int32_t* indices, int32_t batch, int32_t n_tok,
                float* audio_out, void* stream);

/* ---- quantiser kernels, context-free --------------------------------------------------------------- */

/* Fused FSQ (vq/__init__.py:25-30 + vq/fsq.py:30-68, eval): x [n][feat] -> q_feature [n][feat], indices [n],
 * level_indices [n][n_levels] (NULL to skip), latents [n][n_levels] (NULL to skip; test hook).
 * w_in [n_levels][feat], b_in [n_levels], w_out [feat][n_levels], b_out [feat] are device pointers.
 * With x == NULL the `latents` buffer is an INPUT (project_in skipped). */
int l3ac_fsq_forward(const float* x, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels,
                     const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                     float* q_feature, int32_t* indices, float* level_indices, float* latents, void* stream);

/* The rounding half of the quantiser on its own: SuperFSQ.quantize_act_value (vq/fsq.py:56-65) ->
 * level_indices_to_indices (:67-68) -> inv_act (:21) -> project_out (vq/__init__.py:29).  act [n][n_levels] holds the
 * activation values in [0, 1], i.e. what tanh_act (vq/fsq_act.py:38-39) returns, so no transcendental sits between the
 * input and the rounding: exact k + 0.5 products and their one-ulp neighbours can be presented bit for bit. */
int l3ac_fsq_quantize_act(const float* act, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels,
                          const float* w_out, const float* b_out, float* q_feature, int32_t* indices,
                          float* level_indices, void* stream);

/* VQEmbed.to_features (vq/__init__.py:20-23): indices [n] -> q_feature [n][feat]. */
int l3ac_fsq_decode(const int32_t* indices, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels,
                    const float* w_out, const float* b_out, float* q_feature, void* stream);

/* Measurement aid (no reference counterpart): the quantiser kernel's grid and per-lane access pattern at feat = 128, 6 levels
 * — 512 B read, 512 + 4 + 24 B written per token — with no arithmetic: the HBM rate this access pattern can reach on the
 * box, printed beside the quantiser's own rate by bench.py (`fsq_kernel.copy_ceiling`). */
int l3ac_fsq_copy_ceiling(const float* x, int64_t n, float* q_feature, int32_t* indices, float* level_indices, void* stream);
/* The same copy at a given residency (workgroups per CU, 1 .. 8; 0 = as above: the quantiser kernel's own residency).  The copy kernel uses
 * no LDS and few registers, so its best rate is at a higher residency than the quantiser's: bench.py measures both and divides by the best. */
int l3ac_fsq_copy_ceiling_at(const float* x, int64_t n, float* q_feature, int32_t* indices, float* level_indices, int32_t blocks_per_cu,
                             void* stream);

/* Explicit-codebook nearest neighbour (the search FSQ is the closed form of):
 * queries [n][dim] (= tanh(latents)), codebook [k][dim] (= indices_to_codes(arange(k)), vq/fsq.py:80-81);
 * out_idx[i] = argmin_k ||q_i - c_k||^2, lowest k on exact ties.  dim <= 8.
 * The result is DEFINED by dist = sum_d (q_d - c_d)^2 accumulated with fmaf in dimension order and strict '<' over
 * increasing k; from 5 120 queries on the candidates are first screened on the fp32 matrix cores and then decided by
 * exactly that arithmetic (kernels/fsq.hip, "screened form"), so every size returns the same bits.
 * `scratch` is a caller-owned device buffer of at least l3ac_vq_argmin_scratch_bytes(n, k, form) bytes (partial minima of
 * the codebook slices, code norms, the list of queries that need the full direct-form search): the call allocates nothing
 * and can be captured into a hipGraph.  n < 2^31.
 * `form`: 0 = automatic; 1 = the direct-form scan wherever the screened form would run (the two are compared by
 * tests/test_gpu_blocks.py and timed side by side by tools/vq_argmin_bench.py).  An argument of the call, not library state:
 * nothing another thread does can change what a call — or a graph captured from it — computes. */
int64_t l3ac_vq_argmin_scratch_bytes(int64_t n, int32_t k, int32_t form);
int l3ac_vq_argmin(const float* queries, int64_t n, const float* codebook, int32_t k, int32_t dim, int32_t* out_idx,
                   void* scratch, int64_t scratch_bytes, int32_t form, void* stream);

/* ---- token wire format (no reference counterpart: the reference keeps int32 indices, vq/fsq.py:68) ---------------
 * Per clip, token t occupies bits [t*bits, (t+1)*bits) of a little-endian bit stream, zero-padded to whole 32-bit
 * words; bits = ceil(log2(codebook size)) (17 at 1kbps, 18 at 3kbps).
 *   indices [batch][n_tok] int32  <->  packed [batch][words_per_clip] uint32,  words_per_clip >= ceil(n_tok*bits/32). */
int l3ac_pack_indices(const int32_t* indices, int32_t batch, int32_t n_tok, int32_t bits, uint32_t* packed,
                      int32_t words_per_clip, void* stream);
int l3ac_unpack_indices(const uint32_t* packed, int32_t batch, int32_t n_tok, int32_t bits, int32_t words_per_clip,
                        int32_t* indices, void* stream);

/* ---- single blocks of a context's network, for per-kernel parity tests ------------------------------ */
/* `block` is the reference state-dict prefix of the block inside its module file, e.g. "encoder.blocks.1.0.module".
 * Shapes: x / y are [batch][frames][channels] frame-major. */
int l3ac_op_first_block(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, float* y, void* stream);
int l3ac_op_conv_unit(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                      void* stream);
int l3ac_op_down_layer(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                       void* stream);
int l3ac_op_conv_k3(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                    void* stream);
int l3ac_op_enhance(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                    void* stream);
int l3ac_op_up_layer(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                     void* stream);
/* EnhanceBlock + up layer as the decoder pipeline runs them (the gate is applied to the 1x1 conv's A operand while it is
 * staged, tconv/__init__.py:35-44 + modules.py:160-164): y [batch][frames * scale][cout] */
int l3ac_op_enhance_up(l3ac_ctx* ctx, const char* enhance_block, const char* up_block, const float* x, int32_t batch,
                       int32_t frames, float* y, void* stream);
int l3ac_op_last_block(l3ac_ctx* ctx, const float* x, int32_t batch, int32_t frames, float* audio, void* stream);
int l3ac_op_local_trans(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                        void* stream);
/* whole sub-modules */
int l3ac_op_encoder(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, float* feature, void* stream);
int l3ac_op_en_encoder(l3ac_ctx* ctx, const float* feature, int32_t batch, int32_t frames, float* tokens, void* stream);
int l3ac_op_en_decoder(l3ac_ctx* ctx, const float* tokens, int32_t batch, int32_t n_tok, float* feature, void* stream);
int l3ac_op_decoder(l3ac_ctx* ctx, const float* feature, int32_t batch, int32_t frames, float* audio, void* stream);

/* snake activation on its own (layers.py:29-33): y = x + (alpha + 1e-8)^-1 * sin(alpha * x)^2 for x [rows][c], alpha [c]
 * (device pointers).  mode bit 0: evaluate the two-elements-per-lane form the GEMM epilogues and the fused units use;
 * mode bit 1: y = sin(x)^2 alone (the kernels' own sine; alpha is not used); mode 4: y = gelu(x), the exact (erf) GELU as the
 * kernels evaluate it (tconv/__init__.py:13, the transformer's GEGLU; alpha is not used).
 * A test entry (it synchronises and allocates): the pipeline applies snake inside its GEMM / unit kernels. */
int l3ac_op_snake(const float* x, float* y, int64_t rows, int32_t c, const float* alpha, int32_t mode, void* stream);
/* Validation switch of ONE context: while enabled its output head (modules.py:192-194) stores the Conv1d(c -> 1, k7) result
 * BEFORE the final tanh, so that decoder parity can be checked where tanh's saturation does not hide it.  Like every
 * call on a context, not to be issued while another thread uses (or captures a graph on) the same context. */
int l3ac_ctx_set_head_pretanh(l3ac_ctx* ctx, int32_t enable);

/* ---- per-launch profile (measurement aid; reference has no counterpart) -------------------------------
 * Between l3ac_profile_begin() and l3ac_profile_end() every kernel launched by the calling thread is bracketed
 * by HIP events on its own stream.  l3ac_profile_end() synchronises, aggregates per kernel name (launch count,
 * summed device time, summed algorithmic FLOPs and bytes) and writes at most `cap` entries. */
typedef struct l3ac_profile_entry {
    char name[64];
    int32_t launches;
    int32_t reserved;
    double ms_total;
    double flops;   /* algorithmic floating-point operations of those launches */
    double bytes;   /* algorithmic HBM bytes of those launches */
} l3ac_profile_entry;
int l3ac_profile_begin(void);
int l3ac_profile_end(l3ac_profile_entry* out, int32_t cap, int32_t* n_out);

/* Generic fp32 MFMA GEMM used by every channel contraction: c[m][n] = a[m][:] . w[n][:] + bias[n].
 * a [m][k] (row stride lda), w [n][k], k % 4 == 0.  Exported for the kernel micro-benchmark and tests. */
int l3ac_gemm_f32(const float* a, int64_t lda, const float* w, const float* bias, float* c, int64_t ldc,
                  int64_t m, int32_t n, int32_t k, void* stream);

/* ---- fp32 products on the bf16 matrix cores ("bf16x3" operand splitting; kernels/gemm_split.hip) -----------------
 * Every fp32 operand is split exactly into three bf16 planes (3 x 8 significant bits = the 24-bit significand) and the
 * product is the six plane products of order <= 2, accumulated in fp32: error vs fp64 no larger than the fp32 fmaf
 * chain's (tests/test_gpu_blocks.py::test_gemm_split_accuracy), at 2.67x fewer matrix-core cycles.  The network's large
 * channel contractions use it by default.  The route is a property of the CONTEXT: l3ac_ctx_set_gemm_split(ctx, 0)
 * routes every later product of that context through the exact v_mfma_f32_32x32x2_f32 kernel instead (a context starts on
 * the split route unless L3AC_GEMM_SPLIT=0 is in the environment when it is created); other contexts, and graphs already
 * captured from this one, are not affected.
 * (reference counterpart: none — torch.nn.functional.linear / conv1d on fp32 tensors.) */
/* The split itself, on the HOST (no GPU needed; this is what builds the weight images): planes [3][n] bf16 bit patterns with
 * x[i] == bf16(planes[0][i]) + bf16(planes[1][i]) + bf16(planes[2][i]) exactly for every finite fp32 x[i]
 * (tests/test_host.py::test_bf16x3_split_is_exact). */
void l3ac_split3_host(const float* x, int64_t n, uint16_t* planes);
int l3ac_ctx_set_gemm_split(l3ac_ctx* ctx, int32_t enable);
/* Route options of ONE context by name (same rules as the setters above): "gemm_split", "head_pretanh", and "narrow_ring" — which
 * fused kernel takes the ConvUnits with C <= 48 on the split route: 0 = conv_unit_split_kernel (32 frames per wave) everywhere,
 * 1 (default) = conv_unit_ring_kernel (16 frames per wave, weights resident in / streamed through LDS) at the width where it is the
 * faster one (C = 48), 2 = wherever it exists (C = 24 too).  Both evaluate the same operations; their results agree to rounding.
 * "trans_coop" (default 1): batches of at most 32 clips — a streaming chunk is one — run every LocalTrans stack in the cooperative
 * form of trans_stack_kernel (six co-resident workgroups per clip exchanging partial tiles through global memory); 0 keeps one
 * workgroup per clip.  Both forms return the same bits.  The cooperative form's six workgroups per clip wait for each other: they need
 * six CUs per clip (claimed per context in a process-wide registry; a launch that does not fit runs in the one-workgroup form);
 * failure reporting: l3ac_coop_timeout_count above.  "coop_timeout_ms" (default 250): the time limit of an arrival poll.
 * "coop_release_claim" (any value): returns the CUs this context has claimed for cooperative launches to the per-device registry (a claim
 * otherwise only grows until the context is destroyed); not while a graph captured from the context may still replay a cooperative launch.
 * "coop_test_fault" (test hook, default 0): j + 1 makes workgroup j of every clip withhold its first arrival.
 * "down_fused" (default 2): the encoder down layers 24 -> 48 and 48 -> 96 (Conv1d(k = stride) + ChannelNorm) in one kernel instead of an
 * fp32-MFMA GEMM + row kernel.  2: down_exact_kernel — the arithmetic of those two kernels bit for bit, on both GEMM routes; 1: the bf16x3
 * form of round 4 — faster, equally accurate, a DIFFERENT rounding of those layers (one token of the stress weights changes sides); 0: the
 * two kernels.
 * "wide_sliced" (default 1): the wide ConvUnits (C = 96 .. 256) of few frames — up to 256 tiles of 16, a streaming chunk — as two
 * launches over frame tiles x channel slices instead of the fused kernel whose waves own their frames end to end; 0 never, 2 wherever
 * the form exists.  The same bits either way.
 * "unit_counter" (default 1; its initial value can be set with the environment variable L3AC_UNIT_COUNTER): batch kernels that keep two
 * workgroups per CU resident (the C = 96 ConvUnits, the LegacyUnits) hand their units of work out by a device counter instead of equal
 * static shares (the workgroup dispatched first is served first by every SIMD and would finish its share early); 0: static shares;
 * 2 / 3 (measurement): only the ConvUnits / only the LegacyUnits.  Which workgroup computes a tile does not enter its arithmetic: the
 * same bits either way.
 * Unknown names return L3AC_EINVAL. */
int l3ac_ctx_set_option(l3ac_ctx* ctx, const char* name, int32_t value);
int32_t l3ac_ctx_get_gemm_split(const l3ac_ctx* ctx);
/* Weight image for l3ac_gemm_split_f32: w [n][k] fp32 -> `image` (device, l3ac_gemm_split_image_bytes(n, k) bytes;
 * 0 = shape not eligible: needs n >= 192, k >= 32, k % 8 == 0). */
int64_t l3ac_gemm_split_image_bytes(int32_t n, int32_t k);
int l3ac_gemm_split_image(const float* w, int32_t n, int32_t k, void* image, void* stream);
/* c[m][n] = a[m][:] . w[n][:] + bias[n] with w given as its split image (a [m][k] fp32, row stride lda). */
int l3ac_gemm_split_f32(const float* a, int64_t lda, const void* image, const float* bias, float* c, int64_t ldc,
                        int64_t m, int32_t n, int32_t k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* L3AC_HIP_H */
