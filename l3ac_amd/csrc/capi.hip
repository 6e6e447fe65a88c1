// extern "C" entry points of libl3ac_hip.so (include/l3ac_hip.h).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "network.hpp"
#include "kernels/split_bf16.hpp"

static thread_local std::string g_last_error;

void l3ac_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int l3ac_device_cu_count() {  // per device ordinal: several devices may be driven from one process
    static std::atomic<int> cus[L3AC_MAX_DEVICES] = {};
    const int slot = l3ac_device_slot();
    int v = slot >= 0 ? cus[slot].load(std::memory_order_relaxed) : 0;
    if (v <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        if (slot >= 0) cus[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

static thread_local Profiler* g_profiler = nullptr;
Profiler* l3ac_current_profiler() { return g_profiler; }
void l3ac_set_current_profiler(Profiler* p) { g_profiler = p; }

namespace {

struct DeviceGuard {  // make the context's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

#define L3AC_ENTER(ctx)                                     \
    L3AC_REQUIRE((ctx) != nullptr, "null context");         \
    DeviceGuard guard_((ctx)->device);                      \
    L3AC_REQUIRE(guard_.ok, "cannot select device %d", (ctx)->device)

// The workspace (ws.x0 / x1 / a / h / yi / stats) is reused in place by every call of a context.  Calls issued on one stream
// are ordered by the stream; a caller that pipelines encode(batch n+1) on stream A against decode(batch n) on stream B would
// otherwise overwrite live activations silently.  Each workspace-using entry point holds one of these: on entry the call's
// stream waits for the previous call's completion event when that call ran on another stream; on exit the event is
// re-recorded.  Inside a stream capture nothing is recorded or waited for (events from outside a capture cannot be joined
// into it): a captured sequence must stay on one stream per context, which hipGraph replay then preserves.
struct WorkspaceOrder {
    l3ac_ctx* ctx;
    hipStream_t s;
    bool capturing = false;
    int rc = L3AC_OK;
    WorkspaceOrder(l3ac_ctx* c, hipStream_t stream) : ctx(c), s(stream) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s) (void)hipStreamIsCapturing(s, &st);
        capturing = st != hipStreamCaptureStatusNone;
        if (!capturing && ctx->ws_done_valid && ctx->ws_stream != s && hipStreamWaitEvent(s, ctx->ws_done, 0) != hipSuccess) {
            l3ac_set_error("cannot order stream %p after the context's previous call on stream %p", (void*)s, (void*)ctx->ws_stream);
            rc = L3AC_EHIP;
        }
    }
    ~WorkspaceOrder() {
        if (capturing || !ctx->ws_done) return;
        if (hipEventRecord(ctx->ws_done, s) == hipSuccess) {
            ctx->ws_stream = s;
            ctx->ws_done_valid = true;
        }
    }
};

// A cooperative transformer launch whose arrival poll expired has added to the context's host-visible failure word
// (kernels/trans_stack.hip): results of the call it belonged to are invalid.  Acting on it: the context leaves the cooperative form
// (the condition that starved it — another process on the device, a CU mask — is likely still there), the arrival counters, which
// such a launch may leave non-zero, are zeroed again, and the count is remembered as acted on.  Returns the number of new expiries.
unsigned coop_acknowledge(l3ac_ctx* ctx, bool may_sync) {
    TransCoopState& st = ctx->coop;
    if (!st.fail_host) return 0;
    const unsigned now = *(volatile unsigned*)st.fail_host;
    const unsigned fresh = now - st.seen;
    if (fresh == 0) return 0;
    st.seen = now;
    st.enabled = 0;
    if (may_sync && st.scratch) {
        (void)hipDeviceSynchronize();
        (void)hipMemset((char*)st.scratch + trans_stack_coop_counter_offset(), 0, trans_stack_coop_counter_bytes());
    }
    return fresh;
}

// Every workspace-using entry point first looks at the failure word (a host memory read): a context whose EARLIER call lost a
// cooperative launch to a timeout says so now — once — instead of carrying on as if that call's outputs were good.
int coop_check_on_entry(l3ac_ctx* ctx, hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &st);
    const unsigned fresh = coop_acknowledge(ctx, st == hipStreamCaptureStatusNone);
    if (fresh == 0) {
        ++ctx->coop.calls_since_check;
        return L3AC_OK;
    }
    // The failure word is read HERE, at enqueue time, with no synchronisation: on an asynchronous stream the failing launch may have run
    // any number of calls after it was enqueued, so the report can only bound the affected calls — every cooperative call entered since
    // the last synchronising check that found nothing (l3ac_coop_timeout_pending / _count, validate = True) is suspect.
    const unsigned suspects = ctx->coop.calls_since_check;
    ctx->coop.calls_since_check = 0;
    l3ac_set_error("an earlier call on this context lost %u arrival poll(s) of the cooperative transformer kernel to the time limit "
                   "(its six workgroups per clip were not co-resident: another process or a CU mask on the device?): the outputs of "
                   "that call are INVALID — it is one of the last %u call(s) on this context (all entered since the last synchronising "
                   "check; calls still queued behind it ran the cooperative form too and are suspect as well); the context now runs "
                   "the one-workgroup form (option trans_coop = 0) — repeat those calls",
                   fresh, suspects);
    return L3AC_ECOOP;
}

#define L3AC_ENTER_WS(ctx, stream)                                          \
    L3AC_ENTER(ctx);                                                        \
    L3AC_TRY(coop_check_on_entry((ctx), (hipStream_t)(stream)));            \
    WorkspaceOrder order_((ctx), (hipStream_t)(stream));                    \
    if (order_.rc != L3AC_OK) return order_.rc

template <class Map>
auto lookup(const Map& m, const char* name, const char* kind) -> decltype(m.begin()->second) {
    if (!name) {
        l3ac_set_error("null block name");
        return nullptr;
    }
    auto it = m.find(name);
    if (it == m.end()) {
        l3ac_set_error("no %s block named '%s'", kind, name);
        return nullptr;
    }
    return it->second;
}

}  // namespace

extern "C" {

const char* l3ac_last_error(void) { return g_last_error.c_str(); }
int l3ac_abi_version(void) { return L3AC_ABI_VERSION; }

int l3ac_create(const l3ac_config* cfg, const l3ac_tensor* tensors, int32_t n_tensors, int32_t device, l3ac_ctx** out) {
    L3AC_REQUIRE(cfg && out, "l3ac_create: null argument");
    *out = nullptr;
    // first of all: a caller built against another version of the header must hear THAT, not a geometry or weight error
    L3AC_REQUIRE(cfg->abi_version == L3AC_ABI_VERSION, "l3ac_create: config abi_version %d but this library is ABI %d "
                 "(fill l3ac_config.abi_version from l3ac_abi_version())", cfg->abi_version, L3AC_ABI_VERSION);
    L3AC_REQUIRE(tensors && n_tensors > 0, "l3ac_create: no weight tensors");
    int n_dev = 0;
    L3AC_HIP_CHECK(hipGetDeviceCount(&n_dev));
    L3AC_REQUIRE(device >= 0 && device < n_dev, "l3ac_create: device %d out of range (%d visible)", device, n_dev);
    l3ac_ctx* ctx = new (std::nothrow) l3ac_ctx();
    L3AC_REQUIRE(ctx, "out of host memory");
    ctx->cfg = *cfg;
    ctx->device = device;
    ctx->gemm_split = gemm_split_default();
    {
        const char* e = std::getenv("L3AC_DOWN_FUSED");
        if (e) ctx->down_fused = std::atoi(e);
        e = std::getenv("L3AC_UNIT_COUNTER");  // the initial value of option "unit_counter" (A/B runs of bench.py)
        if (e) ctx->unit_counter = std::atoi(e);  // 0: static shares; 1: all; 2: conv_unit_wide only; 3: legacy units only (measurement)
    }
    DeviceGuard guard(device);
    int rc = guard.ok ? network_build(ctx, tensors, n_tensors) : L3AC_EHIP;
    if (rc == L3AC_OK && hipEventCreateWithFlags(&ctx->ws_done, hipEventDisableTiming) != hipSuccess) {
        l3ac_set_error("hipEventCreate failed");
        rc = L3AC_EHIP;
    }
    if (rc != L3AC_OK) {
        if (ctx->ws_done) (void)hipEventDestroy(ctx->ws_done);
        network_free(ctx);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return L3AC_OK;
}

void l3ac_destroy(l3ac_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->ws_done) (void)hipEventDestroy(ctx->ws_done);
    network_free(ctx);
    delete ctx;
}

int l3ac_reserve(l3ac_ctx* ctx, int32_t batch, int32_t samples) {
    L3AC_ENTER(ctx);
    L3AC_REQUIRE(batch > 0 && samples > 0, "l3ac_reserve: bad shape");
    return workspace_ensure_clip(ctx, batch, samples, nullptr);
}

int l3ac_bad_index_count(l3ac_ctx* ctx, int32_t reset, int64_t* out) {
    L3AC_ENTER(ctx);
    L3AC_REQUIRE(out != nullptr, "bad_index_count: null output");
    L3AC_HIP_CHECK(hipDeviceSynchronize());
    int v = 0;
    L3AC_HIP_CHECK(hipMemcpy(&v, ctx->bad_index_count, sizeof(int), hipMemcpyDeviceToHost));
    *out = v;
    if (reset) L3AC_HIP_CHECK(hipMemset(ctx->bad_index_count, 0, sizeof(int)));
    return L3AC_OK;
}

int l3ac_coop_timeout_count(l3ac_ctx* ctx, int32_t reset, int64_t* out) {
    L3AC_ENTER(ctx);
    L3AC_REQUIRE(out != nullptr, "coop_timeout_count: null output");
    L3AC_HIP_CHECK(hipDeviceSynchronize());
    TransCoopState& st = ctx->coop;
    const unsigned now = st.fail_host ? *(volatile unsigned*)st.fail_host : 0u;
    *out = (int64_t)(unsigned)(now - st.count_base);
    st.calls_since_check = 0;  // (drained: whatever happened has been counted, and is reported by this call)
    (void)coop_acknowledge(ctx, true);  // the caller has been told: fall back, re-zero the counters, no second report on the next call
    if (reset) st.count_base = now;
    return L3AC_OK;
}

// Expired polls the host has NOT yet acted on, after draining the device — and without acting on them: the context's state is left as
// it is, so the next entry point (or l3ac_coop_timeout_count) still reports them.  What a caller that wants to attribute a failure to
// ONE call reads before and after it (encode_audio / decode_audio(validate=True)).
int l3ac_coop_timeout_pending(l3ac_ctx* ctx, int64_t* out) {
    L3AC_ENTER(ctx);
    L3AC_REQUIRE(out != nullptr, "coop_timeout_pending: null output");
    L3AC_HIP_CHECK(hipDeviceSynchronize());
    TransCoopState& st = ctx->coop;
    const unsigned now = st.fail_host ? *(volatile unsigned*)st.fail_host : 0u;
    *out = (int64_t)(unsigned)(now - st.seen);
    if (now == st.seen) st.calls_since_check = 0;  // a synchronising check that found nothing: every call so far is good
    return L3AC_OK;
}

int32_t l3ac_coop_claimed_cus(int32_t device) { return trans_coop_claimed_on_device(device); }

int l3ac_grn_min_norm(l3ac_ctx* ctx, int32_t reset, float* out) {
    L3AC_ENTER(ctx);
    L3AC_REQUIRE(out != nullptr, "grn_min_norm: null output");
    L3AC_HIP_CHECK(hipDeviceSynchronize());
    float v = 0.f;
    L3AC_HIP_CHECK(hipMemcpy(&v, ctx->grn_min_sumsq, sizeof(float), hipMemcpyDeviceToHost));
    *out = std::sqrt(v);
    if (reset) {
        const float inf = INFINITY;
        L3AC_HIP_CHECK(hipMemcpy(ctx->grn_min_sumsq, &inf, sizeof(float), hipMemcpyHostToDevice));
    }
    return L3AC_OK;
}

int64_t l3ac_workspace_bytes(const l3ac_ctx* ctx) { return ctx ? (int64_t)ctx->ws.bytes() : 0; }
int32_t l3ac_hop_length(const l3ac_ctx* ctx) { return ctx ? ctx->hop : 0; }

int l3ac_encode(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, int64_t audio_stride,
                float* q_feature, int32_t* indices, float* level_indices, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    L3AC_REQUIRE(audio && q_feature && indices, "l3ac_encode: null buffer");
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && samples > 0 && audio_stride >= samples, "l3ac_encode: bad shape");
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, samples, s));
    const int frames = (int)round_up64(samples, ctx->hop);  // Codec.preprocess (codec.py:79-84)
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    L3AC_TRY(run_encoder(ctx, s, audio, audio_stride, batch, samples, frames, &cur, &alt));
    int n_tok = 0;
    L3AC_TRY(run_en_encoder(ctx, s, batch, frames / ctx->enc_rate, &cur, &alt, &n_tok));
    FsqArgs f{};
    f.x = cur; f.n = (int64_t)batch * n_tok; f.feat = ctx->cfg.feature_dim; f.n_levels = ctx->cfg.n_levels;
    for (int d = 0; d < f.n_levels; ++d) f.levels[d] = ctx->cfg.levels[d];
    f.w_in = ctx->q_win; f.b_in = ctx->q_bin; f.w_out = ctx->q_wout; f.b_out = ctx->q_bout;
    f.q_feature = q_feature; f.indices = indices; f.level_indices = level_indices;
    return launch_fsq(s, f);
}

int l3ac_decode(l3ac_ctx* ctx, const float* q_feature, const int32_t* indices, int32_t batch, int32_t n_tok,
                float* audio_out, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    L3AC_REQUIRE((q_feature || indices) && audio_out, "l3ac_decode: null buffer");
    L3AC_REQUIRE(batch > 0 && batch <= 65535 && n_tok > 0, "l3ac_decode: bad shape");
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, n_tok * ctx->hop, s));
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    const int64_t n = (int64_t)batch * n_tok;
    if (q_feature) {
        L3AC_HIP_CHECK(hipMemcpyAsync(cur, q_feature, (size_t)n * ctx->cfg.feature_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {  // VQEmbed.to_features (vq/__init__.py:20-23)
        FsqArgs f{};
        f.idx_in = indices; f.n = n; f.feat = ctx->cfg.feature_dim; f.n_levels = ctx->cfg.n_levels;
        for (int d = 0; d < f.n_levels; ++d) f.levels[d] = ctx->cfg.levels[d];
        f.w_out = ctx->q_wout; f.b_out = ctx->q_bout; f.q_feature = cur;
        f.bad_count = ctx->bad_index_count;
        L3AC_TRY(launch_fsq(s, f));
    }
    int frames = 0;
    L3AC_TRY(run_en_decoder(ctx, s, batch, n_tok, &cur, &alt, &frames));
    return run_decoder(ctx, s, batch, frames, &cur, &alt, audio_out);
}

// ---- quantiser kernels --------------------------------------------------------------------------------
int l3ac_fsq_forward(const float* x, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels,
                     const float* w_in, const float* b_in, const float* w_out, const float* b_out, float* q_feature,
                     int32_t* indices, float* level_indices, float* latents, void* stream) {
    L3AC_REQUIRE(levels && n_levels >= 1 && n_levels <= L3AC_MAX_LEVELS, "fsq: bad levels");
    FsqArgs f{};
    f.x = x; f.n = n; f.feat = feat; f.n_levels = n_levels;
    for (int d = 0; d < n_levels; ++d) f.levels[d] = levels[d];
    f.w_in = w_in; f.b_in = b_in; f.w_out = w_out; f.b_out = b_out;
    f.q_feature = q_feature; f.indices = indices; f.level_indices = level_indices; f.latents = latents;
    return launch_fsq((hipStream_t)stream, f);
}

int l3ac_fsq_quantize_act(const float* act, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels, const float* w_out,
                          const float* b_out, float* q_feature, int32_t* indices, float* level_indices, void* stream) {
    L3AC_REQUIRE(act && levels && n_levels >= 1 && n_levels <= L3AC_MAX_LEVELS, "fsq_quantize_act: bad arguments");
    FsqArgs f{};
    f.n = n; f.feat = feat; f.n_levels = n_levels;
    for (int d = 0; d < n_levels; ++d) f.levels[d] = levels[d];
    f.w_out = w_out; f.b_out = b_out; f.q_feature = q_feature; f.indices = indices; f.level_indices = level_indices;
    f.latents = const_cast<float*>(act);  // read only in this mode
    f.act_in = true;
    return launch_fsq((hipStream_t)stream, f);
}

int l3ac_fsq_decode(const int32_t* indices, int64_t n, int32_t feat, const int32_t* levels, int32_t n_levels,
                    const float* w_out, const float* b_out, float* q_feature, void* stream) {
    L3AC_REQUIRE(indices && levels && n_levels >= 1 && n_levels <= L3AC_MAX_LEVELS, "fsq_decode: bad arguments");
    FsqArgs f{};
    f.idx_in = indices; f.n = n; f.feat = feat; f.n_levels = n_levels;
    for (int d = 0; d < n_levels; ++d) f.levels[d] = levels[d];
    f.w_out = w_out; f.b_out = b_out; f.q_feature = q_feature;
    return launch_fsq((hipStream_t)stream, f);
}

int l3ac_fsq_copy_ceiling(const float* x, int64_t n, float* q_feature, int32_t* indices, float* level_indices, void* stream) {
    return launch_fsq_copy_ceiling((hipStream_t)stream, x, n, q_feature, indices, level_indices);
}
int l3ac_fsq_copy_ceiling_at(const float* x, int64_t n, float* q_feature, int32_t* indices, float* level_indices, int32_t blocks_per_cu, void* stream) {
    return launch_fsq_copy_ceiling((hipStream_t)stream, x, n, q_feature, indices, level_indices, blocks_per_cu);
}

int64_t l3ac_vq_argmin_scratch_bytes(int64_t n, int32_t k, int32_t form) {
    return (n > 0 && k > 0) ? (int64_t)vq_argmin_scratch_bytes(n, k, form) : 0;
}

int l3ac_vq_argmin(const float* queries, int64_t n, const float* codebook, int32_t k, int32_t dim, int32_t* out_idx,
                   void* scratch, int64_t scratch_bytes, int32_t form, void* stream) {
    L3AC_REQUIRE(queries && codebook && out_idx, "vq_argmin: null buffer");
    if (n == 0) return L3AC_OK;
    L3AC_REQUIRE(form == 0 || form == 1, "vq_argmin: form must be 0 (automatic) or 1 (direct-form scan)");
    L3AC_REQUIRE(scratch && k > 0 && scratch_bytes >= (int64_t)vq_argmin_scratch_bytes(n, k, form),
                 "vq_argmin: scratch of %lld bytes needed (l3ac_vq_argmin_scratch_bytes), %lld given",
                 (long long)(k > 0 ? vq_argmin_scratch_bytes(n, k, form) : 0), (long long)scratch_bytes);
    return launch_vq_argmin((hipStream_t)stream, queries, n, codebook, k, dim, scratch, out_idx, form);
}

// ---- per-block parity entry points --------------------------------------------------------------------
int l3ac_op_first_block(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, float* y, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    return launch_first_block((hipStream_t)stream, ctx->first, audio, samples, batch, samples, samples, y);
}

// (h: 4C floats per row, or the wide ConvUnit front end's planes of whole 32-frame tiles when that is more)
#define L3AC_OP_SCRATCH(c_max)                                                                                    \
    L3AC_TRY(workspace_ensure(ctx, (size_t)batch * frames * (c_max), (size_t)batch * frames * (c_max),           \
                              std::max((size_t)batch * frames * 4 * (c_max),                                      \
                                       (conv_unit_wide_scratch_bytes((c_max), (int64_t)batch * frames) + 3) / 4), \
                              (size_t)batch * frames * 4, (size_t)batch, (hipStream_t)stream))

int l3ac_op_conv_unit(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                      void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const ConvUnitW* w = lookup(ctx->by_unit, block, "ConvUnit");
    if (!w) return L3AC_EINVAL;
    L3AC_OP_SCRATCH(w->c);
    return run_conv_unit(ctx, (hipStream_t)stream, *w, x, y, batch, frames);
}

int l3ac_op_down_layer(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                       void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const DownW* w = lookup(ctx->by_down, block, "down-layer");
    if (!w) return L3AC_EINVAL;
    // (the layer's form — one fused kernel or GEMM + row kernel — must depend on the shape and the context's route only: the fused
    // kernel cannot run in place, so in-place calls are refused instead of silently taking the other rounding)
    L3AC_REQUIRE(x != y, "op_down_layer: x and y must not alias");
    return run_down(ctx, (hipStream_t)stream, *w, x, y, batch, frames);
}

int l3ac_op_conv_k3(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                    void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const ConvK3W* w = lookup(ctx->by_k3, block, "Conv1d(k3)");
    if (!w) return L3AC_EINVAL;
    return run_conv_k3(ctx, (hipStream_t)stream, *w, x, y, batch, frames);
}

int l3ac_op_enhance(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                    void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const EnhW* w = lookup(ctx->by_enh, block, "EnhanceBlock");
    if (!w) return L3AC_EINVAL;
    L3AC_OP_SCRATCH(w->c);
    return run_enhance(ctx, (hipStream_t)stream, *w, x, y, batch, frames);
}

int l3ac_op_up_layer(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                     void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const UpW* w = lookup(ctx->by_up, block, "up-layer");
    if (!w) return L3AC_EINVAL;
    L3AC_OP_SCRATCH(w->cout);
    return run_up(ctx, (hipStream_t)stream, *w, x, ctx->ws.a, y, batch, frames);
}

int l3ac_op_enhance_up(l3ac_ctx* ctx, const char* enhance_block, const char* up_block, const float* x, int32_t batch, int32_t frames,
                       float* y, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const EnhW* e = lookup(ctx->by_enh, enhance_block, "EnhanceBlock");
    const UpW* w = lookup(ctx->by_up, up_block, "up-layer");
    if (!e || !w) return L3AC_EINVAL;
    L3AC_REQUIRE(e->c == w->cin, "enhance_up: blocks of different widths (%d vs %d)", e->c, w->cin);
    L3AC_REQUIRE(x != y, "op_enhance_up: x and y must not alias (x is only read)");
    L3AC_OP_SCRATCH(w->cin > w->cout ? w->cin : w->cout);
    // the pipeline's fused form: gate applied inside the up conv's A staging; x is only read
    return run_enhance_up(ctx, (hipStream_t)stream, *e, *w, const_cast<float*>(x), ctx->ws.a, y, batch, frames);
}

int l3ac_op_last_block(l3ac_ctx* ctx, const float* x, int32_t batch, int32_t frames, float* audio, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const int c = ctx->head.c;
    L3AC_OP_SCRATCH(c);
    hipStream_t s = (hipStream_t)stream;
    L3AC_HIP_CHECK(hipMemcpyAsync(ctx->ws.x0, x, (size_t)batch * frames * c * sizeof(float), hipMemcpyDeviceToDevice, s));
    return run_last_block(ctx, s, ctx->ws.x0, audio, batch, frames);
}

int l3ac_op_local_trans(l3ac_ctx* ctx, const char* block, const float* x, int32_t batch, int32_t frames, float* y,
                        void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    const LocalTransW* w = lookup(ctx->by_trans, block, "LocalTrans");
    if (!w) return L3AC_EINVAL;
    const int dim = ctx->cfg.feature_dim;
    const size_t rows = (size_t)batch * frames;
    const size_t cols = (size_t)std::max(std::max(3 * ctx->inner, ctx->ff_n), 4 * dim);
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure(ctx, rows * dim, rows * std::max(ctx->inner, dim), rows * cols, 0, (size_t)batch, s));
    if (y != x) L3AC_HIP_CHECK(hipMemcpyAsync(y, x, rows * dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return run_local_trans(ctx, s, *w, y, batch, frames);
}

int l3ac_op_encoder(l3ac_ctx* ctx, const float* audio, int32_t batch, int32_t samples, float* feature, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    L3AC_REQUIRE(samples % ctx->enc_rate == 0, "op_encoder: samples must be a multiple of prod(compress_rates)");
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, samples, s));
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    L3AC_TRY(run_encoder(ctx, s, audio, samples, batch, samples, samples, &cur, &alt));
    const size_t n = (size_t)batch * (samples / ctx->enc_rate) * ctx->cfg.feature_dim;
    L3AC_HIP_CHECK(hipMemcpyAsync(feature, cur, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return L3AC_OK;
}

int l3ac_op_en_encoder(l3ac_ctx* ctx, const float* feature, int32_t batch, int32_t frames, float* tokens, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, frames * ctx->enc_rate, s));
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    const size_t n_in = (size_t)batch * frames * ctx->cfg.feature_dim;
    L3AC_HIP_CHECK(hipMemcpyAsync(cur, feature, n_in * sizeof(float), hipMemcpyDeviceToDevice, s));
    int n_tok = 0;
    L3AC_TRY(run_en_encoder(ctx, s, batch, frames, &cur, &alt, &n_tok));
    L3AC_HIP_CHECK(hipMemcpyAsync(tokens, cur, (size_t)batch * n_tok * ctx->cfg.feature_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return L3AC_OK;
}

int l3ac_op_en_decoder(l3ac_ctx* ctx, const float* tokens, int32_t batch, int32_t n_tok, float* feature, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, n_tok * ctx->hop, s));
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    L3AC_HIP_CHECK(hipMemcpyAsync(cur, tokens, (size_t)batch * n_tok * ctx->cfg.feature_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    int frames = 0;
    L3AC_TRY(run_en_decoder(ctx, s, batch, n_tok, &cur, &alt, &frames));
    L3AC_HIP_CHECK(hipMemcpyAsync(feature, cur, (size_t)batch * frames * ctx->cfg.feature_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return L3AC_OK;
}

int l3ac_op_decoder(l3ac_ctx* ctx, const float* feature, int32_t batch, int32_t frames, float* audio, void* stream) {
    L3AC_ENTER_WS(ctx, stream);
    hipStream_t s = (hipStream_t)stream;
    L3AC_TRY(workspace_ensure_clip(ctx, batch, frames * ctx->enc_rate, s));
    float* cur = ctx->ws.x0;
    float* alt = ctx->ws.x1;
    L3AC_HIP_CHECK(hipMemcpyAsync(cur, feature, (size_t)batch * frames * ctx->cfg.feature_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return run_decoder(ctx, s, batch, frames, &cur, &alt, audio);
}

int l3ac_pack_indices(const int32_t* indices, int32_t batch, int32_t n_tok, int32_t bits, uint32_t* packed,
                      int32_t words_per_clip, void* stream) {
    L3AC_REQUIRE(indices && packed, "pack: null buffer");
    return launch_pack_indices((hipStream_t)stream, indices, batch, n_tok, bits, packed, words_per_clip);
}

int l3ac_unpack_indices(const uint32_t* packed, int32_t batch, int32_t n_tok, int32_t bits, int32_t words_per_clip,
                        int32_t* indices, void* stream) {
    L3AC_REQUIRE(indices && packed, "unpack: null buffer");
    return launch_unpack_indices((hipStream_t)stream, packed, batch, n_tok, bits, words_per_clip, indices);
}

int l3ac_profile_begin(void) {
    L3AC_REQUIRE(g_profiler == nullptr, "profile already active on this thread");
    g_profiler = new (std::nothrow) Profiler();
    L3AC_REQUIRE(g_profiler, "out of host memory");
    return L3AC_OK;
}

int l3ac_profile_end(l3ac_profile_entry* out, int32_t cap, int32_t* n_out) {
    L3AC_REQUIRE(g_profiler != nullptr, "no active profile on this thread");
    Profiler* p = g_profiler;
    g_profiler = nullptr;
    int n = 0;
    int rc = L3AC_OK;
    for (ProfRecord& r : p->records) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) {
            l3ac_set_error("profile: event query failed for %s", r.name);
            rc = L3AC_EHIP;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
        if (!out || rc != L3AC_OK) continue;
        int slot = -1;
        for (int i = 0; i < n; ++i)
            if (std::strncmp(out[i].name, r.name, sizeof(out[i].name)) == 0) slot = i;
        if (slot < 0) {
            if (n >= cap) continue;
            slot = n++;
            std::memset(&out[slot], 0, sizeof(out[slot]));
            std::strncpy(out[slot].name, r.name, sizeof(out[slot].name) - 1);
        }
        out[slot].launches += 1;
        out[slot].ms_total += ms;
        out[slot].flops += r.flops;
        out[slot].bytes += r.bytes;
    }
    if (p->failed && rc == L3AC_OK) {
        l3ac_set_error("profile: event creation failed");
        rc = L3AC_EHIP;
    }
    delete p;
    if (n_out) *n_out = n;
    return rc;
}

int l3ac_gemm_f32(const float* a, int64_t lda, const float* w, const float* bias, float* c, int64_t ldc, int64_t m,
                  int32_t n, int32_t k, void* stream) {
    GemmArgs g{};
    g.a = a; g.lda = lda; g.w = w; g.ldw = k; g.c = c; g.ldc = ldc; g.m = m; g.n = n; g.k = k; g.bias = bias; g.epi = EPI_BIAS;
    return launch_gemm((hipStream_t)stream, g);
}

void l3ac_split3_host(const float* x, int64_t n, uint16_t* planes) {
    for (int64_t i = 0; i < n; ++i) {
        uint16_t pl[3];
        split3_host(x[i], pl);
        planes[i] = pl[0];
        planes[n + i] = pl[1];
        planes[2 * n + i] = pl[2];
    }
}

int l3ac_ctx_set_head_pretanh(l3ac_ctx* ctx, int32_t enable) {
    L3AC_REQUIRE(ctx != nullptr, "null context");
    ctx->head_pretanh = enable != 0;
    return L3AC_OK;
}

int l3ac_op_snake(const float* x, float* y, int64_t rows, int32_t c, const float* alpha, int32_t mode, void* stream) {
    L3AC_REQUIRE(x && y && alpha && rows >= 0 && c > 0 && c % 4 == 0 && c <= 4096 && mode >= 0 && mode <= 4, "op_snake: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* inv = nullptr;  // 1 / (alpha + 1e-8), evaluated in fp32 as layers.py:32 does
    std::vector<float> ha((size_t)c), hi((size_t)c);
    L3AC_HIP_CHECK(hipMemcpyAsync(ha.data(), alpha, (size_t)c * sizeof(float), hipMemcpyDeviceToHost, s));
    L3AC_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < c; ++i) hi[i] = 1.0f / (ha[i] + 1e-8f);
    L3AC_HIP_CHECK(hipMalloc((void**)&inv, (size_t)c * sizeof(float)));
    int rc = L3AC_OK;
    if (hipMemcpyAsync(inv, hi.data(), (size_t)c * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) {
        l3ac_set_error("op_snake: upload failed");
        rc = L3AC_EHIP;
    }
    if (rc == L3AC_OK) rc = launch_snake(s, x, y, rows, c, alpha, inv, mode);
    (void)hipStreamSynchronize(s);
    (void)hipFree(inv);
    return rc;
}

int l3ac_ctx_set_option(l3ac_ctx* ctx, const char* name, int32_t value) {
    L3AC_REQUIRE(ctx != nullptr && name != nullptr, "set_option: null argument");
    const std::string n(name);
    if (n == "gemm_split") ctx->gemm_split = value != 0;
    else if (n == "head_pretanh") ctx->head_pretanh = value != 0;
    else if (n == "wide_sliced") ctx->wide_sliced = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "unit_counter") ctx->unit_counter = value < 0 ? 0 : (value > 3 ? 3 : value);
    else if (n == "narrow_ring") ctx->narrow_ring = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "trans_coop") ctx->coop.enabled = value != 0;
    else if (n == "coop_timeout_ms") ctx->coop.timeout_ms = value < 1 ? 1 : (value > 20000 ? 20000 : value);
    else if (n == "coop_release_claim") {  // (any value) give this context's CUs back to the per-device registry: the next cooperative launch
        // claims what IT needs.  A claim only grows otherwise — a context that once ran 32 clips keeps 192 CUs and pushes other contexts'
        // small batches into the one-workgroup form (a speed matter, never a correctness one).  Not while a graph captured from this
        // context may still replay a cooperative launch: the registry would no longer cover it.
        trans_coop_release(ctx->coop);
    }
    else if (n == "coop_test_fault") ctx->coop.fault_part = value - 1;  // 0 = off, j + 1 = workgroup j of every clip withholds its first arrival
    else if (n == "down_fused") ctx->down_fused = value < 0 ? 0 : (value > 2 ? 2 : value);
    else {
        l3ac_set_error("set_option: unknown option '%s' (gemm_split, head_pretanh, narrow_ring, wide_sliced, unit_counter, trans_coop, coop_timeout_ms, coop_release_claim, coop_test_fault, down_fused)", name);
        return L3AC_EINVAL;
    }
    return L3AC_OK;
}

int l3ac_ctx_set_gemm_split(l3ac_ctx* ctx, int32_t enable) {
    L3AC_REQUIRE(ctx != nullptr, "null context");
    ctx->gemm_split = enable != 0;
    return L3AC_OK;
}
int32_t l3ac_ctx_get_gemm_split(const l3ac_ctx* ctx) { return ctx && ctx->gemm_split ? 1 : 0; }

int64_t l3ac_gemm_split_image_bytes(int32_t n, int32_t k) {
    return (n > 0 && k > 0 && gemm_split_eligible(n, k)) ? gemm_split_image_bytes(n, k) : 0;
}

int l3ac_gemm_split_image(const float* w, int32_t n, int32_t k, void* image, void* stream) {
    L3AC_REQUIRE(gemm_split_eligible(n, k), "split image: shape n=%d k=%d is not eligible (n >= 192, k >= 32, k %% 8 == 0)", n, k);
    return launch_gemm_split_image((hipStream_t)stream, w, k, n, k, (unsigned char*)image);
}

int l3ac_gemm_split_f32(const float* a, int64_t lda, const void* image, const float* bias, float* c, int64_t ldc, int64_t m,
                        int32_t n, int32_t k, void* stream) {
    GemmArgs g{};
    g.a = a; g.lda = lda; g.w_img = (const unsigned char*)image; g.c = c; g.ldc = ldc; g.m = m; g.n = n; g.k = k; g.bias = bias;
    g.epi = EPI_BIAS;
    return launch_gemm_split((hipStream_t)stream, g);
}

}  // extern "C"
