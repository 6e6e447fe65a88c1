// Network construction (weight lookup, layout transforms, upload) and the encode / decode pipelines.
#include "network.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int HEADS = 6;  // LocalTrans.builder (reference l3ac/local_trans.py:51)

struct Builder {
    std::unordered_map<std::string, const l3ac_tensor*> map;
    std::vector<float> host;  // staging image of the device arena
    float* dev = nullptr;
    size_t cap = 0;
    std::string err;
    struct GemmWeight { const float* d; int n, k; };
    std::vector<GemmWeight> gemm_ws;  // [n][k] weights that get a split image
    struct ExtraImage { std::vector<unsigned char> bytes; const unsigned char** target; };
    std::vector<ExtraImage> extra_imgs;  // other bf16x3 images (fused kernels), bound to *target after the upload
    const float* host_of(const float* d) const { return host.data() + (d - dev); }

    const float* gemm(const float* d, int n, int k) {
        if (d && gemm_split_eligible(n, k)) gemm_ws.push_back({d, n, k});
        return d;
    }
    const float* find(const std::string& name, int64_t numel) {
        auto it = map.find(name);
        if (it == map.end()) {
            if (err.empty()) err = "missing weight tensor '" + name + "'";
            return nullptr;
        }
        if (it->second->numel != numel) {
            if (err.empty())
                err = "weight tensor '" + name + "' has " + std::to_string(it->second->numel) + " elements, expected " +
                      std::to_string(numel);
            return nullptr;
        }
        return it->second->data;
    }
    // reserve n floats in the arena (256-byte aligned), return {host pointer, device pointer}
    float* alloc(size_t n, const float** dptr) {
        const size_t off = (host.size() + 63) / 64 * 64;
        if (off + n > cap) {
            if (err.empty()) err = "internal: weight arena overflow";
            *dptr = nullptr;
            static float sink[1];
            return sink;
        }
        host.resize(off + n, 0.f);
        *dptr = dev + off;
        return host.data() + off;
    }
    const float* copy(const std::string& name, int64_t numel) {
        const float* src = find(name, numel);
        const float* d = nullptr;
        float* h = alloc((size_t)numel, &d);
        if (src && d) std::memcpy(h, src, (size_t)numel * sizeof(float));
        return d;
    }
    // Conv1d weight [co][ci][k] -> [co][k][ci]  (k-contiguous rows for the implicit-conv GEMM)
    const float* conv(const std::string& name, int co, int ci, int k) {
        const float* src = find(name, (int64_t)co * ci * k);
        const float* d = nullptr;
        float* h = alloc((size_t)co * ci * k, &d);
        if (src && d)
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < ci; ++c)
                    for (int j = 0; j < k; ++j) h[((size_t)o * k + j) * ci + c] = src[((size_t)o * ci + c) * k + j];
        return d;
    }
    // depth-wise Conv1d weight [c][1][7] -> [7][c]
    const float* dwconv(const std::string& name, int c) {
        const float* src = find(name, (int64_t)c * 7);
        const float* d = nullptr;
        float* h = alloc((size_t)c * 7, &d);
        if (src && d)
            for (int ch = 0; ch < c; ++ch)
                for (int j = 0; j < 7; ++j) h[(size_t)j * c + ch] = src[(size_t)ch * 7 + j];
        return d;
    }
    // snake: 1 / (alpha + 1e-8) evaluated in fp32 exactly as layers.py:32 does
    const float* inv_alpha(const std::string& name, int n) {
        const float* src = find(name, n);
        const float* d = nullptr;
        float* h = alloc((size_t)n, &d);
        if (src && d)
            for (int i = 0; i < n; ++i) h[i] = 1.0f / (src[i] + 1e-8f);
        return d;
    }
};

static float silu(float x) { return x / (1.0f + std::exp(-x)); }

// DynamicPositionBias MLP over the integer distances 0 .. 2W-1 (local_attention.transformer), [heads][2W]
const float* build_bias_table(Builder& b, const std::string& prefix, int dim, int window) {
    const int hdim = dim / 2;
    const float* w0 = b.find(prefix + ".mlp.0.weight", hdim);
    const float* b0 = b.find(prefix + ".mlp.0.bias", hdim);
    const float* w2 = b.find(prefix + ".mlp.2.weight", (int64_t)hdim * hdim);
    const float* b2 = b.find(prefix + ".mlp.2.bias", hdim);
    const float* w4 = b.find(prefix + ".mlp.4.weight", (int64_t)HEADS * hdim);
    const float* b4 = b.find(prefix + ".mlp.4.bias", HEADS);
    const float* d = nullptr;
    float* h = b.alloc((size_t)HEADS * 2 * window, &d);
    if (!(w0 && b0 && w2 && b2 && w4 && b4 && d)) return d;
    std::vector<float> h1(hdim), h2(hdim);
    for (int dist = 0; dist < 2 * window; ++dist) {
        for (int i = 0; i < hdim; ++i) h1[i] = silu(w0[i] * (float)dist + b0[i]);
        for (int i = 0; i < hdim; ++i) {
            float s = b2[i];
            for (int j = 0; j < hdim; ++j) s += w2[(size_t)i * hdim + j] * h1[j];
            h2[i] = silu(s);
        }
        for (int hd = 0; hd < HEADS; ++hd) {
            float s = b4[hd];
            for (int j = 0; j < hdim; ++j) s += w4[(size_t)hd * hdim + j] * h2[j];
            h[(size_t)hd * 2 * window + dist] = s;
        }
    }
    return d;
}

ConvUnitW build_conv_unit(Builder& b, const std::string& p, int c) {
    ConvUnitW u{};
    u.c = c;
    u.dw_w = b.dwconv(p + ".dw_conv.weight", c);
    u.dw_b = b.copy(p + ".dw_conv.bias", c);
    u.ln_w = b.copy(p + ".norm.weight", c);
    u.ln_b = b.copy(p + ".norm.bias", c);
    u.w1 = b.gemm(b.copy(p + ".pw_conv1.weight", (int64_t)4 * c * c), 4 * c, c);
    u.b1 = b.copy(p + ".pw_conv1.bias", 4 * c);
    u.alpha = b.copy(p + ".act.alpha", 4 * c);
    u.inv_alpha = b.inv_alpha(p + ".act.alpha", 4 * c);
    u.gamma = b.copy(p + ".grn.gamma", 4 * c);
    u.beta = b.copy(p + ".grn.beta", 4 * c);
    u.w2 = b.gemm(b.copy(p + ".pw_conv2.weight", (int64_t)4 * c * c), c, 4 * c);
    u.b2 = b.copy(p + ".pw_conv2.bias", c);
    return u;
}

LocalTransW build_local_trans(Builder& b, l3ac_ctx* ctx, const std::string& p, int window, int depth) {
    LocalTransW t{};
    t.window = window;
    const int dim = ctx->cfg.feature_dim;
    const int inner = ctx->inner, ffi = ctx->ff_inner;
    for (int l = 0; l < depth; ++l) {
        const std::string a = p + ".layers." + std::to_string(l) + ".0";
        const std::string f = p + ".layers." + std::to_string(l) + ".1";
        TransLayerW w{};
        w.ln1w = b.copy(a + ".norm.weight", dim);
        w.ln1b = b.copy(a + ".norm.bias", dim);
        w.wqkv = b.gemm(b.copy(a + ".to_qkv.weight", (int64_t)3 * inner * dim), 3 * inner, dim);
        w.wout = b.gemm(b.copy(a + ".to_out.weight", (int64_t)dim * inner), dim, inner);
        w.ln2w = b.copy(f + ".0.weight", dim);
        w.ln2b = b.copy(f + ".0.bias", dim);
        {  // ff.1 [2*ffi][dim] -> (value, gate) 32-row tiles interleaved, zero padded: [ff_n][dim]
            const float* src = b.find(f + ".1.weight", (int64_t)2 * ffi * dim);
            const float* d = nullptr;
            float* h = b.alloc((size_t)ctx->ff_n * dim, &d);
            if (src && d) {
                for (int jb = 0; jb * 32 < ffi; ++jb)
                    for (int r = 0; r < 32; ++r) {
                        const int j = jb * 32 + r;
                        if (j >= ffi) continue;
                        std::memcpy(h + (size_t)(64 * jb + r) * dim, src + (size_t)j * dim, dim * sizeof(float));
                        std::memcpy(h + (size_t)(64 * jb + 32 + r) * dim, src + (size_t)(ffi + j) * dim, dim * sizeof(float));
                    }
            }
            w.wff1 = b.gemm(d, ctx->ff_n, dim);
        }
        {  // ff.4 [dim][ffi] -> [dim][ff_pad] zero padded along k
            const float* src = b.find(f + ".4.weight", (int64_t)dim * ffi);
            const float* d = nullptr;
            float* h = b.alloc((size_t)dim * ctx->ff_pad, &d);
            if (src && d)
                for (int o = 0; o < dim; ++o) std::memcpy(h + (size_t)o * ctx->ff_pad, src + (size_t)o * ffi, ffi * sizeof(float));
            w.wff2 = b.gemm(d, dim, ctx->ff_pad);
        }
        t.layers.push_back(w);
    }
    t.bias_table = build_bias_table(b, p + ".dynamic_pos_bias", dim, window);
    return t;
}

int free_buf(float*& p) {
    if (p) {
        L3AC_HIP_CHECK(hipFree(p));
        p = nullptr;
    }
    return L3AC_OK;
}

}  // namespace

int network_build(l3ac_ctx* ctx, const l3ac_tensor* tensors, int n_tensors) {
    const l3ac_config& c = ctx->cfg;
    L3AC_REQUIRE(c.abi_version == L3AC_ABI_VERSION, "config abi_version %d != %d", c.abi_version, L3AC_ABI_VERSION);
    L3AC_REQUIRE(c.n_enc >= 2 && c.n_enc <= L3AC_MAX_STAGES && c.n_dec >= 2 && c.n_dec <= L3AC_MAX_STAGES,
                 "bad stage counts (n_enc=%d n_dec=%d)", c.n_enc, c.n_dec);
    L3AC_REQUIRE(c.n_levels >= 1 && c.n_levels <= L3AC_MAX_LEVELS, "bad n_levels=%d", c.n_levels);
    L3AC_REQUIRE(c.feature_dim >= 8 && c.feature_dim % 8 == 0, "feature_dim=%d must be a multiple of 8", c.feature_dim);
    L3AC_REQUIRE(c.en_coder_compress_rate >= 1 && c.en_coder_window_size >= 1, "bad en_coder geometry");
    for (int i = 0; i < c.n_enc; ++i) L3AC_REQUIRE(c.enc_dims[i] % 4 == 0, "encoder_dims must be multiples of 4");
    for (int i = 0; i < c.n_dec; ++i) L3AC_REQUIRE(c.dec_dims[i] % 4 == 0, "decoder_dims must be multiples of 4");
    ctx->enc_rate = 1;
    for (int i = 0; i + 1 < c.n_enc; ++i) ctx->enc_rate *= c.compress_rates[i];
    int dec_rate = 1;
    for (int i = 0; i + 1 < c.n_dec; ++i) dec_rate *= c.decode_rates[i];
    L3AC_REQUIRE(dec_rate == ctx->enc_rate, "prod(decode_rates)=%d != prod(compress_rates)=%d", dec_rate, ctx->enc_rate);
    ctx->hop = ctx->enc_rate * c.en_coder_compress_rate;
    ctx->dim_head = c.feature_dim / 4;  // LocalTrans.builder: dim_head = feature_dim // 4
    ctx->inner = HEADS * ctx->dim_head;
    ctx->ff_inner = (int)((double)c.feature_dim * 4 * 2 / 3);  // FeedForward: int(dim * mult * 2 / 3)
    ctx->ff_pad = (int)round_up64(ctx->ff_inner, 4);
    ctx->ff_n = 64 * (int)ceil_div64(ctx->ff_inner, 32);
    const bool compressed = c.en_coder_compress_rate != 1;
    if (compressed) L3AC_REQUIRE(c.en_coder_depth >= 2, "compressed en_decoder needs en_coder_depth >= 2");

    Builder b;
    int64_t total = 0;
    for (int i = 0; i < n_tensors; ++i) {
        L3AC_REQUIRE(tensors[i].name && tensors[i].data && tensors[i].numel > 0, "tensor %d is malformed", i);
        b.map[tensors[i].name] = &tensors[i];
        total += tensors[i].numel;
    }
    b.cap = (size_t)total * 2 + (size_t)n_tensors * 128 + (size_t)HEADS * 2 * 8 * (c.en_coder_window_size * c.en_coder_compress_rate + 64) + (1 << 16);
    L3AC_HIP_CHECK(hipMalloc((void**)&ctx->arena, b.cap * sizeof(float)));
    b.dev = ctx->arena;
    b.host.reserve(b.cap);
    ctx->arena_floats = b.cap;

    // ---- encoder (modules.py:71-116) ------------------------------------------------------------------
    {
        const std::string p = "encoder.blocks.0";
        const float* d = nullptr;
        float* tw = b.alloc(5 * 4 * 7, &d);
        ctx->first.tw = d;
        float* tb = b.alloc(5 * 4, &d);
        ctx->first.tb = d;
        for (int i = 0; i < 5; ++i) {
            const float* w = b.find(p + ".blocks." + std::to_string(i) + ".1.weight", 28);
            const float* bb = b.find(p + ".blocks." + std::to_string(i) + ".1.bias", 4);
            if (w && bb) {
                std::memcpy(tw + i * 28, w, 28 * sizeof(float));
                std::memcpy(tb + i * 4, bb, 4 * sizeof(float));
            }
        }
        const int d0 = c.enc_dims[0];
        ctx->first.d0 = d0;
        ctx->first.w1 = b.copy(p + ".conv_1.weight", 80 * 20);
        ctx->first.b1 = b.copy(p + ".conv_1.bias", 80);
        const float* w2 = b.find(p + ".conv_2.weight", (int64_t)d0 * 81);
        float* w2t = b.alloc((size_t)81 * d0, &d);  // transposed to [81][d0]
        ctx->first.w2 = d;
        if (w2)
            for (int o = 0; o < d0; ++o)
                for (int i = 0; i < 81; ++i) w2t[(size_t)i * d0 + o] = w2[(size_t)o * 81 + i];
        ctx->first.b2 = b.copy(p + ".conv_2.bias", d0);
    }
    ctx->enc_units.assign(c.n_enc, {});
    ctx->enc_down.assign(c.n_enc - 1, {});
    int blk = 1;
    for (int i = 0; i + 1 < c.n_enc; ++i) {
        for (int j = 0; j < c.enc_depths[i]; ++j)
            ctx->enc_units[i].push_back(build_conv_unit(b, "encoder.blocks." + std::to_string(blk) + "." + std::to_string(j) + ".module", c.enc_dims[i]));
        DownW& d = ctx->enc_down[i];
        d.cin = c.enc_dims[i];
        d.cout = c.enc_dims[i + 1];
        d.stride = c.compress_rates[i];
        L3AC_REQUIRE(d.stride >= 1, "bad compress rate");
        const std::string p = "encoder.blocks." + std::to_string(blk + 1);
        d.w = b.gemm(b.conv(p + ".0.weight", d.cout, d.cin, d.stride), d.cout, d.stride * d.cin);
        d.b = b.copy(p + ".0.bias", d.cout);
        d.nw = b.copy(p + ".1.weight", d.cout);
        d.nb = b.copy(p + ".1.bias", d.cout);
        blk += 2;
    }
    for (int j = 0; j < c.enc_depths[c.n_enc - 1]; ++j)
        ctx->enc_units[c.n_enc - 1].push_back(build_conv_unit(b, "encoder.blocks." + std::to_string(blk) + "." + std::to_string(j) + ".module", c.enc_dims[c.n_enc - 1]));
    ctx->enc_out.cin = c.enc_dims[c.n_enc - 1];
    ctx->enc_out.cout = c.feature_dim;
    ctx->enc_out.w = b.conv("encoder.blocks." + std::to_string(blk + 1) + ".weight", c.feature_dim, ctx->enc_out.cin, 3);
    ctx->enc_out.b = b.copy("encoder.blocks." + std::to_string(blk + 1) + ".bias", c.feature_dim);
    const std::string enc_out_name = "encoder.blocks." + std::to_string(blk + 1);

    // ---- local-attention stacks (local_trans.py:56-94, :129-186; en_codec.py:25-44) -------------------------
    const int win = c.en_coder_window_size;
    if (compressed) {
        const int r = c.en_coder_compress_rate;
        ctx->en_enc.push_back(build_local_trans(b, ctx, "en_encoder.down_trans.trans", win * r, 3 / 2));
        ctx->en_enc.push_back(build_local_trans(b, ctx, "en_encoder.local_trans", win, 3 - 3 / 2));
        ctx->en_down.cin = ctx->en_down.cout = c.feature_dim;
        ctx->en_down.stride = r;
        ctx->en_down.w = b.gemm(b.conv("en_encoder.down_trans.down_layer.weight", c.feature_dim, c.feature_dim, r), c.feature_dim, r * c.feature_dim);
        ctx->en_down.b = b.copy("en_encoder.down_trans.down_layer.bias", c.feature_dim);
        ctx->en_dec.push_back(build_local_trans(b, ctx, "en_decoder.local_trans", win, c.en_coder_depth - 2));
        ctx->en_dec.push_back(build_local_trans(b, ctx, "en_decoder.up_trans.trans", win * r, 2));
    } else {
        ctx->en_enc.push_back(build_local_trans(b, ctx, "en_encoder.local_trans", win, 1));
        ctx->en_dec.push_back(build_local_trans(b, ctx, "en_decoder.local_trans", win, c.en_coder_depth));
    }

    // the stacks as trans_stack_kernel takes them (the stack vectors no longer reallocate: the image targets stay valid)
    if (b.err.empty()) {
        const int dim = c.feature_dim;
        for (auto* stacks : {&ctx->en_enc, &ctx->en_dec})
            for (LocalTransW& t : *stacks) {
                if (!trans_stack_supported(dim, ctx->dim_head, HEADS, ctx->ff_inner, 1, t.window, (int)t.layers.size())) continue;
                const float* d = nullptr;
                float* ln = b.alloc(t.layers.size() * 4 * (size_t)dim, &d);
                if (!d) continue;
                t.stack_ln = d;
                std::vector<unsigned char> img;
                img.reserve(t.layers.size() * (size_t)trans_stack_layer_image_bytes());
                for (size_t l = 0; l < t.layers.size(); ++l) {
                    const TransLayerW& w = t.layers[l];
                    const float* srcs[4] = {w.ln1w, w.ln1b, w.ln2w, w.ln2b};
                    for (int q = 0; q < 4; ++q) std::memcpy(ln + (l * 4 + q) * dim, b.host_of(srcs[q]), dim * sizeof(float));
                    trans_stack_layer_image(img, b.host_of(w.wqkv), b.host_of(w.wout), b.host_of(w.wff1), ctx->ff_n, b.host_of(w.wff2), ctx->ff_pad);
                }
                b.extra_imgs.push_back({std::move(img), &t.stack_img});
            }
    }

    // ---- quantiser (vq/__init__.py:13-14) -------------------------------------------------------------
    ctx->q_win = b.copy("quantizer.project_in.weight", (int64_t)c.n_levels * c.feature_dim);
    ctx->q_bin = b.copy("quantizer.project_in.bias", c.n_levels);
    ctx->q_wout = b.copy("quantizer.project_out.weight", (int64_t)c.feature_dim * c.n_levels);
    ctx->q_bout = b.copy("quantizer.project_out.bias", c.feature_dim);

    // ---- decoder (modules.py:135-201) -----------------------------------------------------------------
    ctx->dec_in.cin = c.feature_dim;
    ctx->dec_in.cout = c.dec_dims[0];
    ctx->dec_in.w = b.gemm(b.conv("decoder.blocks.0.weight", c.dec_dims[0], c.feature_dim, 3), c.dec_dims[0], 3 * c.feature_dim);
    ctx->dec_in.b = b.copy("decoder.blocks.0.bias", c.dec_dims[0]);
    ctx->dec_units.assign(c.n_dec - 1, {});
    ctx->dec_enh.assign(c.n_dec - 1, {});
    ctx->dec_up.assign(c.n_dec - 1, {});
    blk = 1;
    for (int i = 0; i + 1 < c.n_dec; ++i) {
        const int ci = c.dec_dims[i], co = c.dec_dims[i + 1];
        for (int j = 0; j < c.dec_depths[i]; ++j)
            ctx->dec_units[i].push_back(build_conv_unit(b, "decoder.blocks." + std::to_string(blk) + "." + std::to_string(j) + ".module", ci));
        EnhW& e = ctx->dec_enh[i];
        e.c = ci;
        const std::string ep = "decoder.blocks." + std::to_string(blk + 1);
        const float* d = nullptr;
        float* tw = b.alloc(4 * 7, &d);
        e.t.tw = d;
        float* tb = b.alloc(4, &d);
        e.t.tb = d;
        for (int p = 0; p < 4; ++p) {
            const float* w = b.find(ep + ".blocks." + std::to_string(p) + ".1.weight", 7);
            const float* bb = b.find(ep + ".blocks." + std::to_string(p) + ".1.bias", 1);
            if (w && bb) {
                std::memcpy(tw + p * 7, w, 7 * sizeof(float));
                tb[p] = bb[0];
            }
        }
        e.in_w = b.copy(ep + ".merge_layer.0.weight", 4);
        e.in_b = b.copy(ep + ".merge_layer.0.bias", 4);
        e.gate_w = b.copy(ep + ".merge_layer.1.weight", (int64_t)ci * 4);
        e.gate_b = b.copy(ep + ".merge_layer.1.bias", ci);
        UpW& u = ctx->dec_up[i];
        u.cin = ci;
        u.cout = co;
        u.scale = c.decode_rates[i];
        const std::string up = "decoder.blocks." + std::to_string(blk + 2);
        u.w = b.gemm(b.copy(up + ".0.weight", (int64_t)co * ci), co, ci);
        u.b = b.copy(up + ".0.bias", co);
        u.nw = b.copy(up + ".2.weight", co);
        u.nb = b.copy(up + ".2.bias", co);
        blk += 3;
    }
    {
        const int cl = c.dec_dims[c.n_dec - 1];
        const std::string lp = "decoder.blocks." + std::to_string(blk) + ".block";
        const int dils[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            const std::string p = lp + ".0." + std::to_string(u) + ".module.block";
            LegacyW l{};
            l.c = cl;
            l.dil = dils[u];
            l.a0 = b.copy(p + ".0.alpha", cl);
            l.ia0 = b.inv_alpha(p + ".0.alpha", cl);
            l.w1 = b.conv(p + ".1.weight", cl, cl, 7);
            l.b1 = b.copy(p + ".1.bias", cl);
            l.a1 = b.copy(p + ".2.alpha", cl);
            l.ia1 = b.inv_alpha(p + ".2.alpha", cl);
            l.w2 = b.copy(p + ".3.weight", (int64_t)cl * cl);
            l.b2 = b.copy(p + ".3.bias", cl);
            ctx->legacy.push_back(l);
        }
        if (b.err.empty()) {  // (the unit vectors no longer reallocate: the image targets stay valid)
            for (auto* stages : {&ctx->enc_units, &ctx->dec_units})
                for (auto& stage : *stages)
                    for (ConvUnitW& u : stage)
                        if (conv_unit_wide_supported(u.c)) {  // (C = 96 .. 256; on the exact route C = 96 takes conv_unit_fused_kernel: fp32 weights)
                            b.extra_imgs.push_back({conv_unit_wide_image(b.host_of(u.w1), b.host_of(u.w2), u.c), &u.wide_img});
                        } else if (conv_unit_fused_supported(u.c)) {
                            b.extra_imgs.push_back({conv_unit_w1_image(b.host_of(u.w1), u.c), &u.w1_img});
                            b.extra_imgs.push_back({conv_unit_w2_image(b.host_of(u.w2), u.c), &u.w2_img});
                            if (conv_unit_ring_supported(u.c))
                                b.extra_imgs.push_back({conv_unit_ring_image(b.host_of(u.w1), b.host_of(u.w2), u.c), &u.ring_img});
                        }
        }
        if (b.err.empty()) {  // (ctx->dec_up was sized before the loop above: the targets stay valid)
            for (UpW& u : ctx->dec_up)
                if (up_fused_supported(u.cin, u.cout)) b.extra_imgs.push_back({up_fused_image(b.host_of(u.w), u.cin, u.cout), &u.fused_img});
            for (DownW& d : ctx->enc_down)  // (sized before the loops above as well)
                if (d.nw && down_fused_supported(d.cin, d.stride, d.cout))
                {
                    b.extra_imgs.push_back({up_fused_image(b.host_of(d.w), d.cin * d.stride, d.cout), &d.fused_img});
                    b.extra_imgs.push_back({down_exact_image(b.host_of(d.w), d.cin * d.stride, d.cout), &d.exact_img});
                }
        }
        if (b.err.empty() && last_block_fused_supported(cl, 9)) {  // (ctx->legacy no longer reallocates: the targets stay valid)
            for (LegacyW& l : ctx->legacy) {
                b.extra_imgs.push_back({legacy_w1_image(b.host_of(l.w1), cl), &l.w1_img});
                b.extra_imgs.push_back({legacy_w2_image(b.host_of(l.w2), cl), &l.w2_img});
            }
        }
        ctx->head.c = cl;
        ctx->head.alpha = b.copy(lp + ".1.alpha", cl);
        ctx->head.inv_alpha = b.inv_alpha(lp + ".1.alpha", cl);
        ctx->head.w = b.conv(lp + ".2.weight", 1, cl, 7);  // [1][7][c]
        ctx->head.b = b.copy(lp + ".2.bias", 1);
    }
    if (!b.err.empty()) {
        l3ac_set_error("%s", b.err.c_str());
        return L3AC_EWEIGHT;
    }
    L3AC_HIP_CHECK(hipMemcpy(ctx->arena, b.host.data(), b.host.size() * sizeof(float), hipMemcpyHostToDevice));

    // ---- bf16x3 split images of the GEMM weights (from the arena's host staging copy: final layouts) and of the
    //      fused kernels' weights, in one device allocation ------------------------------------------------------
    {
        auto pad = [](size_t n) { return (n + 255) / 256 * 256; };
        size_t total_img = 0;
        for (const auto& g : b.gemm_ws) total_img += pad((size_t)gemm_split_image_bytes(g.n, g.k));
        for (const auto& e : b.extra_imgs) total_img += pad(e.bytes.size());
        if (total_img) {
            std::vector<unsigned char> himg(total_img, 0);
            L3AC_HIP_CHECK(hipMalloc((void**)&ctx->img_arena, total_img));
            ctx->img_bytes = total_img;
            size_t off = 0;
            for (const auto& g : b.gemm_ws) {
                gemm_split_image_host(b.host_of(g.d), g.k, g.n, g.k, himg.data() + off);
                ctx->split_img[g.d] = ctx->img_arena + off;
                off += pad((size_t)gemm_split_image_bytes(g.n, g.k));
            }
            for (const auto& e : b.extra_imgs) {
                std::memcpy(himg.data() + off, e.bytes.data(), e.bytes.size());
                *e.target = ctx->img_arena + off;
                off += pad(e.bytes.size());
            }
            L3AC_HIP_CHECK(hipMemcpy(ctx->img_arena, himg.data(), total_img, hipMemcpyHostToDevice));
        }
    }

    L3AC_HIP_CHECK(hipMalloc((void**)&ctx->bad_index_count, sizeof(int)));
    L3AC_HIP_CHECK(hipMemset(ctx->bad_index_count, 0, sizeof(int)));
    L3AC_HIP_CHECK(hipMalloc((void**)&ctx->wide_counters, 64));
    L3AC_HIP_CHECK(hipMemset(ctx->wide_counters, 0, 64));
    {  // cooperative form of the transformer stacks (few clips: the streaming chunk): its scratch, counters zeroed ONCE here —
       // every launch leaves them zeroed again
        bool any = false;
        for (const std::vector<LocalTransW>* v : {&ctx->en_enc, &ctx->en_dec})
            for (const LocalTransW& t : *v) any = any || t.stack_img != nullptr;
        if (any) {
            L3AC_HIP_CHECK(hipMalloc(&ctx->coop.scratch, trans_stack_coop_bytes()));
            L3AC_HIP_CHECK(hipMemset(ctx->coop.scratch, 0, trans_stack_coop_bytes()));
            // the failure word lives in pinned host memory the device can add to: the host reads it without a device call
            L3AC_HIP_CHECK(hipHostMalloc((void**)&ctx->coop.fail_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
            *ctx->coop.fail_host = 0;
            L3AC_HIP_CHECK(hipHostGetDevicePointer((void**)&ctx->coop.fail_dev, ctx->coop.fail_host, 0));
        }
        const char* e = std::getenv("L3AC_TRANS_COOP");
        if (e) ctx->coop.enabled = std::atoi(e);
    }
    {  // GRN guard: starts at +inf
        const float inf = INFINITY;
        L3AC_HIP_CHECK(hipMalloc((void**)&ctx->grn_min_sumsq, sizeof(float)));
        L3AC_HIP_CHECK(hipMemcpy(ctx->grn_min_sumsq, &inf, sizeof(float), hipMemcpyHostToDevice));
    }
    // ---- name index for the per-block entry points ------------------------------------------------------
    blk = 1;
    for (int i = 0; i < c.n_enc; ++i) {
        for (size_t j = 0; j < ctx->enc_units[i].size(); ++j)
            ctx->by_unit["encoder.blocks." + std::to_string(blk) + "." + std::to_string(j) + ".module"] = &ctx->enc_units[i][j];
        if (i + 1 < c.n_enc) ctx->by_down["encoder.blocks." + std::to_string(blk + 1)] = &ctx->enc_down[i];
        blk += 2;
    }
    ctx->by_k3[enc_out_name] = &ctx->enc_out;
    ctx->by_k3["decoder.blocks.0"] = &ctx->dec_in;
    blk = 1;
    for (int i = 0; i + 1 < c.n_dec; ++i) {
        for (size_t j = 0; j < ctx->dec_units[i].size(); ++j)
            ctx->by_unit["decoder.blocks." + std::to_string(blk) + "." + std::to_string(j) + ".module"] = &ctx->dec_units[i][j];
        ctx->by_enh["decoder.blocks." + std::to_string(blk + 1)] = &ctx->dec_enh[i];
        ctx->by_up["decoder.blocks." + std::to_string(blk + 2)] = &ctx->dec_up[i];
        blk += 3;
    }
    if (compressed) {
        ctx->by_trans["en_encoder.down_trans.trans"] = &ctx->en_enc[0];
        ctx->by_trans["en_encoder.local_trans"] = &ctx->en_enc[1];
        ctx->by_down["en_encoder.down_trans.down_layer"] = &ctx->en_down;
        ctx->by_trans["en_decoder.local_trans"] = &ctx->en_dec[0];
        ctx->by_trans["en_decoder.up_trans.trans"] = &ctx->en_dec[1];
    } else {
        ctx->by_trans["en_encoder.local_trans"] = &ctx->en_enc[0];
        ctx->by_trans["en_decoder.local_trans"] = &ctx->en_dec[0];
    }
    return L3AC_OK;
}

void network_free(l3ac_ctx* ctx) {
    if (ctx->bad_index_count) (void)hipFree(ctx->bad_index_count);
    ctx->bad_index_count = nullptr;
    if (ctx->wide_counters) (void)hipFree(ctx->wide_counters);
    ctx->wide_counters = nullptr;
    trans_coop_release(ctx->coop);
    if (ctx->coop.scratch) (void)hipFree(ctx->coop.scratch);
    ctx->coop.scratch = nullptr;
    if (ctx->coop.fail_host) (void)hipHostFree(ctx->coop.fail_host);
    ctx->coop.fail_host = ctx->coop.fail_dev = nullptr;
    if (ctx->grn_min_sumsq) (void)hipFree(ctx->grn_min_sumsq);
    ctx->grn_min_sumsq = nullptr;
    if (ctx->arena) (void)hipFree(ctx->arena);
    ctx->arena = nullptr;
    if (ctx->img_arena) (void)hipFree(ctx->img_arena);
    ctx->img_arena = nullptr;
    ctx->split_img.clear();
    Workspace& w = ctx->ws;
    for (float** p : {&w.x0, &w.x1, &w.a, &w.h, &w.yi, &w.stats, &w.sumsq}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

// ---------------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------------
int workspace_ensure(l3ac_ctx* ctx, size_t x_floats, size_t a_floats, size_t h_floats, size_t yi_floats, size_t batch,
                     hipStream_t s) {
    Workspace& w = ctx->ws;
    if (x_floats <= w.x_cap && a_floats <= w.a_cap && h_floats <= w.h_cap && yi_floats <= w.yi_cap && batch <= w.b_cap)
        return L3AC_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &st);
    if (st != hipStreamCaptureStatusNone) {
        l3ac_set_error("workspace too small while the stream is capturing: call l3ac_reserve() first");
        return L3AC_ENOMEM;
    }
    L3AC_HIP_CHECK(hipDeviceSynchronize());  // buffers may still be in use by earlier launches
    auto grow = [&](float*& p, size_t& cap, size_t want, size_t mult) -> int {
        if (want <= cap) return L3AC_OK;
        L3AC_TRY(free_buf(p));
        cap = 0;
        L3AC_HIP_CHECK(hipMalloc((void**)&p, (want * mult + 64) * sizeof(float)));
        cap = want;
        return L3AC_OK;
    };
    const size_t x_want = x_floats > w.x_cap ? x_floats : w.x_cap;
    if (x_want > w.x_cap) {
        L3AC_TRY(free_buf(w.x0));
        L3AC_TRY(free_buf(w.x1));
        w.x_cap = 0;
        L3AC_HIP_CHECK(hipMalloc((void**)&w.x0, (x_want + 64) * sizeof(float)));
        L3AC_HIP_CHECK(hipMalloc((void**)&w.x1, (x_want + 64) * sizeof(float)));
        w.x_cap = x_want;
    }
    L3AC_TRY(grow(w.a, w.a_cap, a_floats, 1));
    L3AC_TRY(grow(w.h, w.h_cap, h_floats, 1));
    L3AC_TRY(grow(w.yi, w.yi_cap, yi_floats, 1));
    if (batch > w.b_cap) {
        L3AC_TRY(free_buf(w.stats));
        L3AC_TRY(free_buf(w.sumsq));
        w.b_cap = 0;
        L3AC_HIP_CHECK(hipMalloc((void**)&w.stats, (batch * 8 + 64) * sizeof(float)));
        L3AC_HIP_CHECK(hipMalloc((void**)&w.sumsq, (batch + 64) * sizeof(float)));
        w.b_cap = batch;
    }
    return L3AC_OK;
}

int workspace_ensure_clip(l3ac_ctx* ctx, int batch, int samples, hipStream_t s) {
    const l3ac_config& c = ctx->cfg;
    const int64_t frames0 = round_up64(samples, ctx->hop);
    size_t x = 0, a = 0, h = 0, yi = 0;
    auto upd = [](size_t& m, int64_t v) { if ((size_t)v > m) m = (size_t)v; };
    int64_t f = frames0;
    // (h is also the scratch of the wide ConvUnit's front end: bf16x3 planes of whole 32-frame tiles of ALL the batch's
    // rows, conv_unit_wide_scratch_bytes — larger than 4C floats per row when batch * frames is small)
    size_t wide_bytes = 0;
    auto wide = [&](int cdim, int64_t fr) {
        if (conv_unit_wide_supported(cdim)) wide_bytes = std::max(wide_bytes, conv_unit_wide_scratch_bytes(cdim, (int64_t)batch * fr));
    };
    for (int i = 0; i < c.n_enc; ++i) {
        upd(x, f * c.enc_dims[i]);
        upd(h, f * 4 * c.enc_dims[i]);
        wide(c.enc_dims[i], f);
        if (i + 1 < c.n_enc) f /= c.compress_rates[i];
    }
    const int64_t feat_frames = f;
    upd(x, f * c.feature_dim);
    const int64_t tr_cols = std::max<int64_t>(std::max<int64_t>(3 * ctx->inner, ctx->ff_n), 4 * c.feature_dim);
    upd(h, f * tr_cols);
    upd(a, f * std::max<int64_t>(ctx->inner, c.feature_dim));
    for (int i = 0; i < c.n_dec; ++i) {
        upd(x, f * c.dec_dims[i]);
        upd(h, f * 4 * c.dec_dims[i]);
        wide(c.dec_dims[i], f);
        upd(yi, f * 4);
        if (i + 1 < c.n_dec) {
            upd(x, f * c.dec_dims[i + 1]);
            f *= c.decode_rates[i];
        }
    }
    (void)feat_frames;
    upd(a, (int64_t)x);
    return workspace_ensure(ctx, x * batch, a * batch, std::max(h * batch, (wide_bytes + 3) / 4), yi * batch, (size_t)batch, s);
}

// ---------------------------------------------------------------------------------------------------------
// blocks
// ---------------------------------------------------------------------------------------------------------
// the wide fused kernel computes on the bf16 matrix cores only (bf16x3): it belongs to the split route
static bool use_wide(const l3ac_ctx* ctx, const ConvUnitW& w) {
    return !ctx->cfg.grn_exact && w.wide_img && ctx->gemm_split && conv_unit_wide_supported(w.c);
}

int conv_unit_step(l3ac_ctx* ctx, hipStream_t s, const ConvUnitW& w, float** cur, float** alt, int batch, int frames) {
    if (use_wide(ctx, w)) {
        L3AC_TRY(launch_conv_unit_wide(s, w, *cur, *alt, reinterpret_cast<unsigned char*>(ctx->ws.h), ctx->ws.h_cap * sizeof(float), batch, frames, ctx->wide_sliced, ctx->unit_counter == 1 || ctx->unit_counter == 2 ? ctx->wide_counters : nullptr));
        float* t = *cur;
        *cur = *alt;
        *alt = t;
        return L3AC_OK;
    }
    if (!ctx->cfg.grn_exact && conv_unit_fused_supported(w.c)) {
        L3AC_TRY(launch_conv_unit_fused(s, w, *cur, *alt, batch, frames, ctx->gemm_split, ctx->narrow_ring));
        float* t = *cur;
        *cur = *alt;
        *alt = t;
        return L3AC_OK;
    }
    return run_conv_unit(ctx, s, w, *cur, *cur, batch, frames);
}

static int run_conv_unit_rows(l3ac_ctx* ctx, hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames) {
    const int64_t rows = (int64_t)batch * frames;
    Workspace& ws = ctx->ws;
    RowArgs r{};  // dw_conv + LayerNorm (modules.py:33-35)
    r.x = x; r.y = ws.a; r.batch = batch; r.frames_in = frames; r.frames_out = frames; r.c = w.c;
    r.src = SRC_DWCONV7; r.norm = NORM_LN; r.dw_w = w.dw_w; r.dw_b = w.dw_b; r.nw = w.ln_w; r.nb = w.ln_b; r.eps = 1e-8f;
    L3AC_TRY(launch_rows(s, r));
    GemmArgs g{};  // pw_conv1 -> snake -> GRN (modules.py:36-38)
    g.a = ws.a; g.lda = w.c; g.w = w.w1; g.w_img = ctx->img(w.w1); g.ldw = w.c; g.c = ws.h; g.ldc = 4 * w.c; g.m = rows; g.n = 4 * w.c; g.k = w.c;
    g.bias = w.b1; g.alpha = w.alpha; g.inv_alpha = w.inv_alpha; g.gamma = w.gamma; g.beta = w.beta;
    g.epi = ctx->cfg.grn_exact ? EPI_SNAKE : EPI_SNAKE_GRN;
    L3AC_TRY(launch_gemm(s, g));
    if (ctx->cfg.grn_exact) {
        L3AC_TRY(launch_grn_sumsq(s, ws.h, batch, (int64_t)frames * 4 * w.c, ws.sumsq));
        L3AC_TRY(launch_grn_apply(s, ws.h, batch, frames, 4 * w.c, ws.sumsq, w.gamma, w.beta, ctx->grn_min_sumsq));
    }
    GemmArgs g2{};  // pw_conv2 + residual (modules.py:39, xtract/nn/layers.py:59-62)
    g2.a = ws.h; g2.lda = 4 * w.c; g2.w = w.w2; g2.w_img = ctx->img(w.w2); g2.ldw = 4 * w.c; g2.c = y; g2.ldc = w.c; g2.m = rows; g2.n = w.c; g2.k = 4 * w.c;
    g2.bias = w.b2; g2.epi = EPI_BIAS_RES; g2.res = x; g2.ldres = w.c;
    return launch_gemm(s, g2);
}

static int conv_unit_group(const l3ac_ctx* ctx, const ConvUnitW& w, int batch, int frames);

// All ConvUnits of one stage.  Wide (unfused) units run in place, so the clip-group loop can sit OUTSIDE the units: a
// group's activations then stay in the Infinity Cache from one unit to the next as well.
int run_conv_units(l3ac_ctx* ctx, hipStream_t s, const std::vector<ConvUnitW>& units, float** cur, float** alt, int batch,
                   int frames) {
    if (units.empty()) return L3AC_OK;
    const bool fused = (!ctx->cfg.grn_exact && conv_unit_fused_supported(units[0].c)) || use_wide(ctx, units[0]);
    if (fused || units.size() == 1) {
        for (const ConvUnitW& u : units) L3AC_TRY(conv_unit_step(ctx, s, u, cur, alt, batch, frames));
        return L3AC_OK;
    }
    const int group = conv_unit_group(ctx, units[0], batch, frames);
    for (int b0 = 0; b0 < batch; b0 += group) {
        const int nb = std::min(group, batch - b0);
        float* xg = *cur + (int64_t)b0 * frames * units[0].c;
        for (const ConvUnitW& u : units) L3AC_TRY(run_conv_unit_rows(ctx, s, u, xg, xg, nb, frames));
    }
    return L3AC_OK;
}

int run_conv_unit(l3ac_ctx* ctx, hipStream_t s, const ConvUnitW& w, const float* x, float* y, int batch, int frames) {
    if (x != y && use_wide(ctx, w))
        return launch_conv_unit_wide(s, w, x, y, reinterpret_cast<unsigned char*>(ctx->ws.h), ctx->ws.h_cap * sizeof(float), batch, frames, ctx->wide_sliced, ctx->unit_counter == 1 || ctx->unit_counter == 2 ? ctx->wide_counters : nullptr);
    if (!ctx->cfg.grn_exact && x != y && conv_unit_fused_supported(w.c)) return launch_conv_unit_fused(s, w, x, y, batch, frames, ctx->gemm_split, ctx->narrow_ring);
    const int group = conv_unit_group(ctx, w, batch, frames);
    for (int b0 = 0; b0 < batch; b0 += group) {
        const int nb = std::min(group, batch - b0);
        const int64_t off = (int64_t)b0 * frames * w.c;
        L3AC_TRY(run_conv_unit_rows(ctx, s, w, x + off, y + off, nb, frames));
    }
    return L3AC_OK;
}

static int conv_unit_group(const l3ac_ctx* ctx, const ConvUnitW& w, int batch, int frames) {
    // Clips are independent, so the unit can run over groups of clips whose hidden tensor (4C floats per frame) stays in the
    // 256 MB Infinity Cache between the two products instead of making an HBM round trip (measured, 1kbps x 256: 19.15 ->
    // 18.65 ms per step at 192 MB; 96 MB and below lose more to the smaller launches than they save).  L3AC_UNIT_CHUNK_MB
    // overrides the target size of that tensor per group (0 = whole batch in one go).
    static const int64_t chunk_mb = [] {
        const char* e = std::getenv("L3AC_UNIT_CHUNK_MB");
        return e ? (int64_t)std::atoi(e) : (int64_t)192;
    }();
    const int64_t per_clip = (int64_t)frames * 4 * w.c * sizeof(float);
    int group = batch;
    if (chunk_mb > 0 && !ctx->cfg.grn_exact) group = (int)std::max<int64_t>(1, std::min<int64_t>(batch, (chunk_mb << 20) / per_clip));
    return group;
}

// the one-kernel form of a down layer (up_fused_kernel<K, Cout, DOWN>): bf16x3 route, the narrow encoder stages' widths
static bool use_down_fused(const l3ac_ctx* ctx, const DownW& w) { return ctx->down_fused == 1 && ctx->gemm_split && w.fused_img && w.nw; }
// the exact one-kernel form (down_exact_kernel): the bits of the GEMM + row kernel it replaces, on either route
static bool use_down_exact(const l3ac_ctx* ctx, const DownW& w) { return ctx->down_fused == 2 && w.exact_img && w.nw; }

int run_down(l3ac_ctx* ctx, hipStream_t s, const DownW& w, const float* x, float* y, int batch, int frames) {
    L3AC_REQUIRE(frames % w.stride == 0, "down layer: frames=%d not a multiple of stride %d", frames, w.stride);
    if (use_down_fused(ctx, w) && x != y) return launch_down_fused(s, w, x, y, batch, frames / w.stride);
    if (use_down_exact(ctx, w) && x != y) return launch_down_exact(s, w, x, y, batch, frames / w.stride);
    const int64_t rows_out = (int64_t)batch * (frames / w.stride);
    GemmArgs g{};  // Conv1d(k = stride): non-overlapping patches are contiguous in the frame-major layout
    g.a = x; g.lda = (int64_t)w.stride * w.cin; g.w = w.w; g.w_img = ctx->img(w.w); g.ldw = (int64_t)w.stride * w.cin; g.c = y; g.ldc = w.cout;
    g.m = rows_out; g.n = w.cout; g.k = w.stride * w.cin; g.bias = w.b; g.epi = EPI_BIAS;
    L3AC_TRY(launch_gemm(s, g));
    if (w.nw) {  // ChannelNorm channels_first (modules.py:98), in place
        RowArgs r{};
        r.x = y; r.y = y; r.batch = batch; r.frames_in = frames / w.stride; r.frames_out = frames / w.stride; r.c = w.cout;
        r.src = SRC_PLAIN; r.norm = NORM_CN; r.nw = w.nw; r.nb = w.nb; r.eps = 1e-8f;
        L3AC_TRY(launch_rows(s, r));
    }
    (void)ctx;
    return L3AC_OK;
}

int run_conv_k3(l3ac_ctx* ctx, hipStream_t s, const ConvK3W& w, const float* x, float* y, int batch, int frames) {
    GemmArgs g{};
    g.a = x; g.lda = w.cin; g.taps = 3; g.dil = 1; g.cin = w.cin; g.frames = frames;
    g.w = w.w; g.w_img = ctx->img(w.w); g.ldw = 3 * w.cin; g.c = y; g.ldc = w.cout; g.m = (int64_t)batch * frames; g.n = w.cout; g.k = 3 * w.cin;
    g.bias = w.b; g.epi = EPI_BIAS;
    (void)ctx;
    return launch_gemm(s, g);
}

int run_enhance(l3ac_ctx* ctx, hipStream_t s, const EnhW& w, const float* x, float* y, int batch, int frames) {
    Workspace& ws = ctx->ws;
    L3AC_TRY(launch_enhance_branches(s, w.t, x, batch, frames, w.c, ws.yi));
    L3AC_TRY(launch_enhance_stats(s, ws.yi, batch, frames, ws.stats));
    RowArgs r{};
    r.x = x; r.y = y; r.batch = batch; r.frames_in = frames; r.frames_out = frames; r.c = w.c;
    r.src = SRC_GATE; r.norm = NORM_NONE; r.yi = ws.yi; r.stats = ws.stats; r.in_w = w.in_w; r.in_b = w.in_b;
    r.gate_w = w.gate_w; r.gate_b = w.gate_b;
    return launch_rows(s, r);
}

int run_up(l3ac_ctx* ctx, hipStream_t s, const UpW& w, const float* x, float* tmp, float* y, int batch, int frames) {
    GemmArgs g{};  // 1x1 conv (modules.py:161)
    g.a = x; g.lda = w.cin; g.w = w.w; g.w_img = ctx->img(w.w); g.ldw = w.cin; g.c = tmp; g.ldc = w.cout; g.m = (int64_t)batch * frames; g.n = w.cout; g.k = w.cin;
    g.bias = w.b; g.epi = EPI_BIAS;
    L3AC_TRY(launch_gemm(s, g));
    RowArgs r{};  // Upsample(linear) + ChannelNorm (modules.py:162-163)
    r.x = tmp; r.y = y; r.batch = batch; r.frames_in = frames; r.frames_out = (int64_t)frames * w.scale; r.c = w.cout;
    r.src = SRC_LERP; r.scale = w.scale; r.norm = NORM_CN; r.nw = w.nw; r.nb = w.nb; r.eps = 1e-8f;
    (void)ctx;
    return launch_rows(s, r);
}

// the one-kernel form of EnhanceBlock gate + up layer (kernels/up_fused.hip): bf16x3 route, the narrow stages' widths
static bool use_up_fused(const l3ac_ctx* ctx, const UpW& w) { return ctx->gemm_split && w.fused_img != nullptr; }

// EnhanceBlock + UpLayer of one decoder stage: the gate is applied inside the up conv's A staging (no pass of its own).
// x is left untouched; tmp holds the conv output at the input rate.
int run_enhance_up(l3ac_ctx* ctx, hipStream_t s, const EnhW& e, const UpW& w, float* x, float* tmp, float* y, int batch, int frames) {
    // separate passes (gate as a row kernel, in place; then the up layer) where the gated GEMM does not cover the geometry — and where
    // the 1x1 conv is wide enough for the bf16x3 route (512 -> 256: the gated fp32-MFMA GEMM ran it at 81 TFLOP/s, 0.173 ms per step
    // and 41 us for a single clip; a memory-bound gate pass + the split GEMM take 0.10 ms / 16 us).  The choice depends on the
    // weight's shape and the context's route only, never on the batch.
    const bool wide_up = ctx->img(w.w) != nullptr && gemm_split_eligible(w.cout, w.cin);
    if (w.cin % 16 != 0 || e.c != w.cin || wide_up) {
        L3AC_REQUIRE(tmp != nullptr, "enhance_up: scratch missing");
        // the pipeline calls this in place (x == y: the gate may overwrite x); any other caller's x is only read, so the gated
        // rows go to the hidden-tensor scratch (free here: the up layer's GEMM writes tmp)
        float* gated = x;
        if (x != y) {
            L3AC_REQUIRE((size_t)batch * frames * e.c <= ctx->ws.h_cap, "enhance_up: workspace too small for the gated rows");
            gated = ctx->ws.h;
        }
        L3AC_TRY(run_enhance(ctx, s, e, x, gated, batch, frames));
        return run_up(ctx, s, w, gated, tmp, y, batch, frames);
    }
    Workspace& ws = ctx->ws;
    L3AC_TRY(launch_enhance_branches(s, e.t, x, batch, frames, e.c, ws.yi));
    L3AC_TRY(launch_enhance_stats(s, ws.yi, batch, frames, ws.stats));
    if (use_up_fused(ctx, w) && x != y)  // gate, conv, upsample and ChannelNorm in one kernel; it reads x while it writes y
        return launch_up_fused(s, e, w, x, ws.yi, ws.stats, y, batch, frames);
    L3AC_REQUIRE(tmp != nullptr, "enhance_up: scratch missing");
    GemmArgs g{};  // gate (tconv/__init__.py:35-44) + 1x1 conv (modules.py:161)
    g.a = x; g.lda = w.cin; g.w = w.w; g.ldw = w.cin; g.c = tmp; g.ldc = w.cout; g.m = (int64_t)batch * frames; g.n = w.cout; g.k = w.cin;
    g.bias = w.b; g.epi = EPI_BIAS;
    g.gate_yi = ws.yi; g.gate_stats = ws.stats; g.gate_in_w = e.in_w; g.gate_in_b = e.in_b; g.gate_w = e.gate_w; g.gate_b = e.gate_b;
    g.gate_frames = frames;
    L3AC_TRY(launch_gemm(s, g));
    RowArgs r{};  // Upsample(linear) + ChannelNorm (modules.py:162-163)
    r.x = tmp; r.y = y; r.batch = batch; r.frames_in = frames; r.frames_out = (int64_t)frames * w.scale; r.c = w.cout;
    r.src = SRC_LERP; r.scale = w.scale; r.norm = NORM_CN; r.nw = w.nw; r.nb = w.nb; r.eps = 1e-8f;
    return launch_rows(s, r);
}

int run_last_block(l3ac_ctx* ctx, hipStream_t s, float* x, float* audio, int batch, int frames) {
    Workspace& ws = ctx->ws;
    const int64_t rows = (int64_t)batch * frames;
    int max_dil = 1;
    for (const LegacyW& l : ctx->legacy) max_dil = l.dil > max_dil ? l.dil : max_dil;
    if (last_block_fused_supported(ctx->head.c, max_dil)) {  // fused units ping-pong between x and the scratch buffer
        float* cur = x;
        float* alt = ws.a;
        for (const LegacyW& l : ctx->legacy) {
            L3AC_TRY(launch_legacy_unit_fused(s, l, cur, alt, batch, frames, ctx->gemm_split, ctx->unit_counter == 1 || ctx->unit_counter == 3 ? ctx->wide_counters + 4 : nullptr));
            float* t = cur;
            cur = alt;
            alt = t;
        }
        return launch_head_fused(s, ctx->head, cur, batch, frames, audio, ctx->head_pretanh);
    }
    for (const LegacyW& l : ctx->legacy) {  // modules.py:47-64
        L3AC_TRY(launch_snake(s, x, ws.a, rows, l.c, l.a0, l.ia0));
        GemmArgs g{};
        g.a = ws.a; g.lda = l.c; g.taps = 7; g.dil = l.dil; g.cin = l.c; g.frames = frames;
        g.w = l.w1; g.w_img = ctx->img(l.w1); g.ldw = 7 * l.c; g.c = ws.h; g.ldc = l.c; g.m = rows; g.n = l.c; g.k = 7 * l.c;
        g.bias = l.b1; g.epi = EPI_SNAKE; g.alpha = l.a1; g.inv_alpha = l.ia1;
        L3AC_TRY(launch_gemm(s, g));
        GemmArgs g2{};
        g2.a = ws.h; g2.lda = l.c; g2.w = l.w2; g2.w_img = ctx->img(l.w2); g2.ldw = l.c; g2.c = x; g2.ldc = l.c; g2.m = rows; g2.n = l.c; g2.k = l.c;
        g2.bias = l.b2; g2.epi = EPI_BIAS_RES; g2.res = x; g2.ldres = l.c;
        L3AC_TRY(launch_gemm(s, g2));
    }
    const HeadW& hd = ctx->head;  // Snake1d -> Conv1d(c -> 1, k7) -> Tanh (modules.py:192-194)
    L3AC_TRY(launch_snake(s, x, ws.a, rows, hd.c, hd.alpha, hd.inv_alpha));
    return launch_head(s, ws.a, batch, frames, hd.c, hd.w, hd.b, audio, ctx->head_pretanh);
}

// the fused stack kernel computes on the bf16 matrix cores only (bf16x3): it belongs to the split route
static bool use_trans_stack(const l3ac_ctx* ctx, const LocalTransW& w, int frames) {
    return ctx->gemm_split && w.stack_img && w.stack_ln &&
           trans_stack_supported(ctx->cfg.feature_dim, ctx->dim_head, HEADS, ctx->ff_inner, frames, w.window, (int)w.layers.size());
}

int run_local_trans(l3ac_ctx* ctx, hipStream_t s, const LocalTransW& w, float* x, int batch, int frames) {
    if (use_trans_stack(ctx, w, frames))  // one launch for the whole stack, one workgroup per clip
        return launch_trans_stack(s, w, x, batch, frames, (float)std::pow((double)ctx->dim_head, -0.5), &ctx->coop);
    Workspace& ws = ctx->ws;
    const int dim = ctx->cfg.feature_dim;
    const int64_t rows = (int64_t)batch * frames;
    for (const TransLayerW& l : w.layers) {
        RowArgs r{};  // LocalMHA prenorm
        r.x = x; r.y = ws.a; r.batch = batch; r.frames_in = frames; r.frames_out = frames; r.c = dim;
        r.src = SRC_PLAIN; r.norm = NORM_LN; r.nw = l.ln1w; r.nb = l.ln1b; r.eps = 1e-5f;
        L3AC_TRY(launch_rows(s, r));
        GemmArgs g{};  // to_qkv (no bias)
        g.a = ws.a; g.lda = dim; g.w = l.wqkv; g.w_img = ctx->img(l.wqkv); g.ldw = dim; g.c = ws.h; g.ldc = 3 * ctx->inner; g.m = rows; g.n = 3 * ctx->inner; g.k = dim;
        g.epi = EPI_BIAS;
        L3AC_TRY(launch_gemm(s, g));
        L3AC_TRY(launch_attention(s, ws.h, ws.a, w.bias_table, batch, frames, HEADS, ctx->dim_head, w.window));
        GemmArgs go{};  // to_out + residual (local_trans.py:45)
        go.a = ws.a; go.lda = ctx->inner; go.w = l.wout; go.w_img = ctx->img(l.wout); go.ldw = ctx->inner; go.c = x; go.ldc = dim; go.m = rows; go.n = dim; go.k = ctx->inner;
        go.epi = EPI_BIAS_RES; go.res = x; go.ldres = dim;
        L3AC_TRY(launch_gemm(s, go));
        r.nw = l.ln2w; r.nb = l.ln2b;  // FeedForward LayerNorm
        L3AC_TRY(launch_rows(s, r));
        GemmArgs f1{};  // Linear(dim, 2*inner) + GEGLU, value/gate tiles interleaved at upload
        f1.a = ws.a; f1.lda = dim; f1.w = l.wff1; f1.w_img = ctx->img(l.wff1); f1.ldw = dim; f1.c = ws.h; f1.ldc = ctx->ff_pad; f1.m = rows; f1.n = ctx->ff_n; f1.k = dim;
        f1.epi = EPI_GEGLU; f1.n_out = ctx->ff_inner;
        L3AC_TRY(launch_gemm(s, f1));
        GemmArgs f2{};  // Linear(inner, dim) + residual (local_trans.py:46)
        f2.a = ws.h; f2.lda = ctx->ff_pad; f2.w = l.wff2; f2.w_img = ctx->img(l.wff2); f2.ldw = ctx->ff_pad; f2.c = x; f2.ldc = dim; f2.m = rows; f2.n = dim; f2.k = ctx->ff_pad;
        f2.epi = EPI_BIAS_RES; f2.res = x; f2.ldres = dim;
        L3AC_TRY(launch_gemm(s, f2));
    }
    return L3AC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// sub-modules
// ---------------------------------------------------------------------------------------------------------
static inline void swap_bufs(float** a, float** b) {
    float* t = *a;
    *a = *b;
    *b = t;
}

int run_encoder(l3ac_ctx* ctx, hipStream_t s, const float* audio, int64_t audio_stride, int batch, int samples,
                int frames, float** cur, float** alt) {
    const l3ac_config& c = ctx->cfg;
    L3AC_TRY(launch_first_block(s, ctx->first, audio, audio_stride, batch, samples, frames, *cur));
    int f = frames;
    for (int i = 0; i < c.n_enc; ++i) {
        L3AC_TRY(run_conv_units(ctx, s, ctx->enc_units[i], cur, alt, batch, f));
        if (i + 1 < c.n_enc) {
            L3AC_TRY(run_down(ctx, s, ctx->enc_down[i], *cur, *alt, batch, f));
            swap_bufs(cur, alt);
            f /= c.compress_rates[i];
        }
    }
    L3AC_TRY(run_conv_k3(ctx, s, ctx->enc_out, *cur, *alt, batch, f));
    swap_bufs(cur, alt);
    return L3AC_OK;
}

int run_en_encoder(l3ac_ctx* ctx, hipStream_t s, int batch, int frames, float** cur, float** alt, int* n_tok) {
    // input (B, C, T) permuted to (B, T, C) by the reference (local_trans.py:162): already frame-major here
    if (ctx->en_enc.size() == 2) {
        L3AC_TRY(run_local_trans(ctx, s, ctx->en_enc[0], *cur, batch, frames));
        L3AC_TRY(run_down(ctx, s, ctx->en_down, *cur, *alt, batch, frames));
        swap_bufs(cur, alt);
        frames /= ctx->en_down.stride;
        L3AC_TRY(run_local_trans(ctx, s, ctx->en_enc[1], *cur, batch, frames));
    } else {
        L3AC_TRY(run_local_trans(ctx, s, ctx->en_enc[0], *cur, batch, frames));
    }
    *n_tok = frames;
    return L3AC_OK;
}

int run_en_decoder(l3ac_ctx* ctx, hipStream_t s, int batch, int n_tok, float** cur, float** alt, int* frames) {
    L3AC_TRY(run_local_trans(ctx, s, ctx->en_dec[0], *cur, batch, n_tok));
    int f = n_tok;
    if (ctx->en_dec.size() == 2) {
        const int r = ctx->cfg.en_coder_compress_rate;
        RowArgs u{};  // UpTransV2.up_layer (local_trans.py:121-124)
        u.x = *cur; u.y = *alt; u.batch = batch; u.frames_in = f; u.frames_out = (int64_t)f * r; u.c = ctx->cfg.feature_dim;
        u.src = SRC_LERP; u.scale = r; u.norm = NORM_NONE;
        L3AC_TRY(launch_rows(s, u));
        swap_bufs(cur, alt);
        f *= r;
        L3AC_TRY(run_local_trans(ctx, s, ctx->en_dec[1], *cur, batch, f));
    }
    *frames = f;
    return L3AC_OK;
}

int run_decoder(l3ac_ctx* ctx, hipStream_t s, int batch, int frames, float** cur, float** alt, float* audio) {
    const l3ac_config& c = ctx->cfg;
    L3AC_TRY(run_conv_k3(ctx, s, ctx->dec_in, *cur, *alt, batch, frames));
    swap_bufs(cur, alt);
    int f = frames;
    for (int i = 0; i + 1 < c.n_dec; ++i) {
        L3AC_TRY(run_conv_units(ctx, s, ctx->dec_units[i], cur, alt, batch, f));
        if (use_up_fused(ctx, ctx->dec_up[i])) {  // the one-kernel form cannot work in place: result in the other buffer
            L3AC_TRY(run_enhance_up(ctx, s, ctx->dec_enh[i], ctx->dec_up[i], *cur, nullptr, *alt, batch, f));
            swap_bufs(cur, alt);
        } else {
            L3AC_TRY(run_enhance_up(ctx, s, ctx->dec_enh[i], ctx->dec_up[i], *cur, *alt, *cur, batch, f));
        }
        f *= c.decode_rates[i];
    }
    return run_last_block(ctx, s, *cur, audio, batch, f);
}
