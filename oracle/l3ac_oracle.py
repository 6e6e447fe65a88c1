"""CPU oracle for the L3AC encode -> quantize -> decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``l3ac_amd/`` imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, as the checker.

It is a functional, fp32, PyTorch-CPU restatement of the reference's inference path, written
op-for-op from the reference sources (each function cites the file:line it follows) and taking the
*folded* weight dictionary ``{module}.{key}`` produced by ``l3ac_amd.weights.folded_weights``.

Parity status
-------------
* conv stacks, FSQ quantiser, pre-processing: PINNED against the imported reference
  (``tests/golden/make_golden.py`` runs ``/root/reference`` in the build container and commits the
  vectors; ``tests/test_oracle_golden.py`` replays them here).
* local-attention transformer (``local_attention==1.11.2``, an un-vendored PyPI dependency that is absent
  from ``/root/reference`` and from this image): restated from the package's published algorithm
  (bucketed causal local attention with look-back 1, DynamicPositionBias MLP, GEGLU feed-forward);
  **parity unpinned** for that arithmetic.  The *wiring* in reference ``local_trans.py`` (window sizes,
  depth split, permutes, down/up layers) is pinned by running the reference's own ``local_trans.py``
  classes over this file's LocalMHA/FeedForward/DynamicPositionBias restatement.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-8  # reference l3ac/xtract/nn/utils.py:33
HEADS = 6  # reference l3ac/local_trans.py:51
_NORM_EPS = torch.tensor(EPS).item()  # ChannelNorm.eps is a 0-dim fp32 tensor (layers.py:71) -> 9.99999994e-9


# =============================================================================================
# primitives (reference l3ac/layers.py)
# =============================================================================================
def snake(x, alpha):
    """layers.py:29-33: x + (alpha + EPS)^-1 * sin(alpha * x)^2."""
    return x + (alpha + EPS).reciprocal() * torch.sin(alpha * x).pow(2)


def channel_norm_first(x, weight, bias):
    """layers.py:50-56 (channels_first): biased variance over dim 1, divide by sqrt."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + torch.tensor(EPS))
    return weight.view(1, -1, 1) * x + bias.view(1, -1, 1)


def channel_norm_last(x, weight, bias):
    """layers.py:79-80 (channels_last): F.layer_norm with eps = float(fp32(1e-8))."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, _NORM_EPS)


def grn(x, gamma, beta):
    """layers.py:112-115.  x is (B, T, C); the norm is over dims [1, 2] so g is (B,1,1) and its mean over
    the size-1 channel dim is itself: n = g / (g + eps)  (== 1.0f whenever g >= 0.25)."""
    g = torch.norm(x, p=2, dim=[1, 2], keepdim=True)
    n = g / (g.mean(dim=-1, keepdim=True) + torch.tensor(EPS))
    return gamma * (x * n) + beta + x


# =============================================================================================
# trend-conv blocks (reference l3ac/tconv)
# =============================================================================================
def trend_pool(x, k):
    """tconv/base.py:8-13: avg_pool(max_pool(|x|)), stride 1, pad k//2; k == 1 is the identity (no abs)."""
    if k > 1:
        args = dict(kernel_size=k, stride=1, padding=k // 2)
        return F.avg_pool1d(F.max_pool1d(x.abs(), **args), **args)
    return x


def trend_branches(w, prefix, x, pool_kernels, dilation_rate):
    """tconv/base.py:27-45 (BaseBlock): per branch TrendPool -> Conv1d(1 -> each_dim, k7, dilated)."""
    outs = []
    for i, pk in enumerate(pool_kernels):
        dil = pk // dilation_rate + 1
        pad = (7 - 1) * dil // 2
        h = trend_pool(x, pk)
        outs.append(F.conv1d(h, w[f"{prefix}.blocks.{i}.1.weight"], w[f"{prefix}.blocks.{i}.1.bias"],
                             dilation=dil, padding=pad))
    return torch.cat(outs, dim=1)


def first_block(w, prefix, x):
    """tconv/__init__.py:8-27 (V3FirstBlock via FirstBlock: pools (1,5,11,21,45), dilation_rate 99 -> dil 1)."""
    h = trend_branches(w, prefix, x, (1, 5, 11, 21, 45), 99)
    h = F.conv1d(h, w[f"{prefix}.conv_1.weight"], w[f"{prefix}.conv_1.bias"])
    h = F.gelu(h)
    y = torch.cat([h, x], dim=1)
    return F.conv1d(y, w[f"{prefix}.conv_2.weight"], w[f"{prefix}.conv_2.bias"])


def enhance_block(w, prefix, x):
    """tconv/__init__.py:30-44: gate from channel 0 only; InstanceNorm1d(4, affine) -> plain Conv1d(4 -> C, 1)."""
    xi = x[:, :1, :]
    yi = trend_branches(w, prefix, xi, (1, 3, 5, 9), 2)
    yi = F.instance_norm(yi, weight=w[f"{prefix}.merge_layer.0.weight"], bias=w[f"{prefix}.merge_layer.0.bias"],
                         use_input_stats=True, eps=1e-5)
    y = F.conv1d(yi, w[f"{prefix}.merge_layer.1.weight"], w[f"{prefix}.merge_layer.1.bias"])
    return x + y * x


# =============================================================================================
# conv stacks (reference l3ac/modules.py)
# =============================================================================================
def conv_unit(w, prefix, x):
    """modules.py:10-41 inside Residual (xtract/nn/layers.py:59-62): x + pw2(GRN(snake(pw1(LN(dw7(x))))))."""
    c = x.shape[1]
    h = F.conv1d(x, w[f"{prefix}.dw_conv.weight"], w[f"{prefix}.dw_conv.bias"], padding=3, groups=c)
    h = h.permute(0, 2, 1)
    h = channel_norm_last(h, w[f"{prefix}.norm.weight"], w[f"{prefix}.norm.bias"])
    h = F.linear(h, w[f"{prefix}.pw_conv1.weight"], w[f"{prefix}.pw_conv1.bias"])
    h = snake(h, w[f"{prefix}.act.alpha"])
    h = grn(h, w[f"{prefix}.grn.gamma"], w[f"{prefix}.grn.beta"])
    h = F.linear(h, w[f"{prefix}.pw_conv2.weight"], w[f"{prefix}.pw_conv2.bias"])
    return x + h.permute(0, 2, 1)


def legacy_unit(w, prefix, x, dilation):
    """modules.py:47-64: x + Conv1x1(snake(Conv_k7_dilated(snake(x)))), channels_first alphas."""
    h = snake(x, w[f"{prefix}.block.0.alpha"])
    h = F.conv1d(h, w[f"{prefix}.block.1.weight"], w[f"{prefix}.block.1.bias"], dilation=dilation,
                 padding=(7 - 1) * dilation // 2)
    h = snake(h, w[f"{prefix}.block.2.alpha"])
    h = F.conv1d(h, w[f"{prefix}.block.3.weight"], w[f"{prefix}.block.3.bias"])
    return x + h


def encoder(w, mc, x, prefix="encoder", taps=None):
    """modules.py:71-116.  x: (B, 1, T) -> (B, feature_dim, T / prod(compress_rates))."""
    dims, depths, strides = mc.encoder_dims, mc.encoder_depths, mc.compress_rates
    x = first_block(w, f"{prefix}.blocks.0", x)
    _tap(taps, "enc.first", x)
    b = 1
    for i, s in enumerate(strides):
        for j in range(depths[i]):
            x = conv_unit(w, f"{prefix}.blocks.{b}.{j}.module", x)
        _tap(taps, f"enc.stage{i}", x)
        x = F.conv1d(x, w[f"{prefix}.blocks.{b + 1}.0.weight"], w[f"{prefix}.blocks.{b + 1}.0.bias"], stride=s)
        x = channel_norm_first(x, w[f"{prefix}.blocks.{b + 1}.1.weight"], w[f"{prefix}.blocks.{b + 1}.1.bias"])
        _tap(taps, f"enc.down{i}", x)
        b += 2
    for j in range(depths[-1]):
        x = conv_unit(w, f"{prefix}.blocks.{b}.{j}.module", x)
    _tap(taps, "enc.tail", x)
    x = F.conv1d(x, w[f"{prefix}.blocks.{b + 1}.weight"], w[f"{prefix}.blocks.{b + 1}.bias"], padding=1)
    _tap(taps, "enc.out", x)
    return x


def decoder(w, mc, x, prefix="decoder", taps=None):
    """modules.py:135-201 with decoder_last_layer='legacy'.  x: (B, feature_dim, T) -> (B, 1, T * prod(rates))."""
    dims, depths, strides = mc.decoder_dims, mc.decoder_depths, mc.decode_rates
    x = F.conv1d(x, w[f"{prefix}.blocks.0.weight"], w[f"{prefix}.blocks.0.bias"], padding=1)
    _tap(taps, "dec.in", x)
    b = 1
    for i, s in enumerate(strides):
        for j in range(depths[i]):
            x = conv_unit(w, f"{prefix}.blocks.{b}.{j}.module", x)
        _tap(taps, f"dec.stage{i}", x)
        x = enhance_block(w, f"{prefix}.blocks.{b + 1}", x)
        _tap(taps, f"dec.enh{i}", x)
        x = F.conv1d(x, w[f"{prefix}.blocks.{b + 2}.0.weight"], w[f"{prefix}.blocks.{b + 2}.0.bias"])
        x = F.interpolate(x, scale_factor=s, mode="linear", align_corners=False)
        x = channel_norm_first(x, w[f"{prefix}.blocks.{b + 2}.2.weight"], w[f"{prefix}.blocks.{b + 2}.2.bias"])
        _tap(taps, f"dec.up{i}", x)
        b += 3
    lp = f"{prefix}.blocks.{b}.block"
    for u, d in enumerate((1, 3, 9)):
        x = legacy_unit(w, f"{lp}.0.{u}.module", x, d)
    _tap(taps, "dec.legacy", x)
    x = snake(x, w[f"{lp}.1.alpha"])
    x = F.conv1d(x, w[f"{lp}.2.weight"], w[f"{lp}.2.bias"], padding=3)
    return torch.tanh(x)


def _tap(taps, name, x):
    if taps is not None:
        taps[name] = x


# =============================================================================================
# local-attention transformer (PyPI local-attention==1.11.2; see module docstring: parity unpinned)
# =============================================================================================
def dynamic_position_bias(w, prefix, i, j):
    """local_attention.transformer.DynamicPositionBias(dim, heads).forward(i, j): an MLP over the integer
    distance, gathered at |(j - i + a) - b|.  Returns (heads, i, j).  Call site: local_trans.py:43."""
    rel = torch.arange(j, dtype=w[f"{prefix}.mlp.0.weight"].dtype).unsqueeze(-1)  # (fp32; fp64 when the test evaluates the stack in double)
    h = F.silu(F.linear(rel, w[f"{prefix}.mlp.0.weight"], w[f"{prefix}.mlp.0.bias"]))
    h = F.silu(F.linear(h, w[f"{prefix}.mlp.2.weight"], w[f"{prefix}.mlp.2.bias"]))
    table = F.linear(h, w[f"{prefix}.mlp.4.weight"], w[f"{prefix}.mlp.4.bias"])  # (j, heads)
    idx = (torch.arange(j - i, j).unsqueeze(1) - torch.arange(j).unsqueeze(0)).abs()
    return table[idx].permute(2, 0, 1)


def position_bias_table(w, prefix, window):
    """The (2W, heads) distance table the bias is gathered from (input independent)."""
    rel = torch.arange(2 * window, dtype=w[f"{prefix}.mlp.0.weight"].dtype).unsqueeze(-1)
    h = F.silu(F.linear(rel, w[f"{prefix}.mlp.0.weight"], w[f"{prefix}.mlp.0.bias"]))
    h = F.silu(F.linear(h, w[f"{prefix}.mlp.2.weight"], w[f"{prefix}.mlp.2.bias"]))
    return F.linear(h, w[f"{prefix}.mlp.4.weight"], w[f"{prefix}.mlp.4.bias"])


def local_attention(q, k, v, window, attn_bias):
    """local_attention.LocalAttention.forward with causal=True, autopad=True, look_backward=1,
    look_forward=0, exact_windowsize=False, scale=None, no rotary.  q,k,v: (B, H, N, D)."""
    b, hh, n, d = q.shape
    q, k, v = (t.reshape(b * hh, n, d) for t in (q, k, v))
    pad = (-n) % window
    if pad:  # autopad: zeros appended at the end of the sequence
        q, k, v = (F.pad(t, (0, 0, 0, pad)) for t in (q, k, v))
    npad = n + pad
    nw = npad // window
    pos = torch.arange(npad).reshape(1, nw, window)
    bq, bk, bv = (t.reshape(b * hh, nw, window, d) for t in (q, k, v))
    bq = bq * (d ** -0.5)

    def look_back(t, pad_value):  # [previous window | this window]; the first window's "previous" is padding
        prev = torch.cat([torch.full_like(t[:, :1], pad_value), t[:, :-1]], dim=1)
        return torch.cat([prev, t], dim=2)

    bk = look_back(bk, -1.0)
    bv = look_back(bv, -1.0)
    pos_k = look_back(pos, -1)
    sim = torch.einsum("bwie,bwje->bwij", bq, bk)
    heads = attn_bias.shape[0]
    sim = sim + attn_bias.repeat(b * hh // heads, 1, 1).unsqueeze(1)
    mask_value = -torch.finfo(sim.dtype).max
    pq = pos.unsqueeze(-1)
    pk = pos_k.unsqueeze(-2)
    sim = sim.masked_fill(pq < pk, mask_value)  # causal
    sim = sim.masked_fill(pk == -1, mask_value)  # look-around padding
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bwij,bwje->bwie", attn, bv).reshape(b * hh, npad, d)
    return out[:, :n].reshape(b, hh, n, d)


def local_mha(w, prefix, x, window, attn_bias):
    """local_attention.transformer.LocalMHA.forward (prenorm LayerNorm eps 1e-5, bias-free qkv / out)."""
    b, n, dim = x.shape
    h = F.layer_norm(x, (dim,), w[f"{prefix}.norm.weight"], w[f"{prefix}.norm.bias"], 1e-5)
    q, k, v = F.linear(h, w[f"{prefix}.to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = (t.reshape(b, n, HEADS, -1).permute(0, 2, 1, 3) for t in (q, k, v))
    out = local_attention(q, k, v, window, attn_bias)
    out = out.permute(0, 2, 1, 3).reshape(b, n, -1)
    return F.linear(out, w[f"{prefix}.to_out.weight"])


def feed_forward(w, prefix, x):
    """local_attention.transformer.FeedForward: LayerNorm -> Linear(dim, 2*inner) -> GEGLU -> Linear(inner, dim)."""
    h = F.layer_norm(x, (x.shape[-1],), w[f"{prefix}.0.weight"], w[f"{prefix}.0.bias"], 1e-5)
    h = F.linear(h, w[f"{prefix}.1.weight"])
    a, gate = h.chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), w[f"{prefix}.4.weight"])


def local_trans(w, prefix, x, window, depth):
    """reference local_trans.py:42-48: bias recomputed per forward, shared by the layers."""
    bias = dynamic_position_bias(w, f"{prefix}.dynamic_pos_bias", window, 2 * window)
    for l in range(depth):
        x = local_mha(w, f"{prefix}.layers.{l}.0", x, window, bias) + x
        x = feed_forward(w, f"{prefix}.layers.{l}.1", x) + x
    return x


def local_trans_dense(w, prefix, x, window, depth):
    """Dense restatement used to cross-check the bucketed algorithm above: query i sees
    keys j <= i in its own or the previous window; bias = table[i - j]."""
    b, n, dim = x.shape
    table = position_bias_table(w, f"{prefix}.dynamic_pos_bias", window)  # (2W, H)
    i = torch.arange(n).unsqueeze(1)
    j = torch.arange(n).unsqueeze(0)
    allowed = (j <= i) & ((j // window) >= (i // window) - 1)
    dist = (i - j).clamp(min=0, max=2 * window - 1)
    bias = table[dist].permute(2, 0, 1)  # (H, n, n)
    for l in range(depth):
        p = f"{prefix}.layers.{l}.0"
        h = F.layer_norm(x, (dim,), w[f"{p}.norm.weight"], w[f"{p}.norm.bias"], 1e-5)
        q, k, v = F.linear(h, w[f"{p}.to_qkv.weight"]).chunk(3, dim=-1)
        q, k, v = (t.reshape(b, n, HEADS, -1).permute(0, 2, 1, 3) for t in (q, k, v))
        sim = torch.einsum("bhie,bhje->bhij", q * (q.shape[-1] ** -0.5), k) + bias
        sim = sim.masked_fill(~allowed, -torch.finfo(sim.dtype).max)
        o = torch.einsum("bhij,bhje->bhie", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(b, n, -1)
        x = F.linear(o, w[f"{p}.to_out.weight"]) + x
        x = feed_forward(w, f"{prefix}.layers.{l}.1", x) + x
    return x


def en_encoder(w, mc, feature, prefix="en_encoder"):
    """local_trans.py:145-165 (compressed) / :56-74 (plain).  (B, C, T) -> (B, T_tok, C)."""
    x = feature.permute(0, 2, 1)
    if mc.compressed:
        r = mc.en_coder_compress_rate
        win = mc.en_coder_window_size + mc.en_coder_cache_size
        first = 3 // 2  # depth fixed at 3 (en_codec.py:35)
        x = local_trans(w, f"{prefix}.down_trans.trans", x, win * r, first)
        x = F.conv1d(x.permute(0, 2, 1), w[f"{prefix}.down_trans.down_layer.weight"],
                     w[f"{prefix}.down_trans.down_layer.bias"], stride=r).permute(0, 2, 1)
        return local_trans(w, f"{prefix}.local_trans", x, win, 3 - first)
    return local_trans(w, f"{prefix}.local_trans", x, mc.en_coder_window_size, 1)  # depth 1 (en_codec.py:27)


def en_decoder(w, mc, feature, prefix="en_decoder"):
    """local_trans.py:168-186 (compressed) / :77-94 (plain).  (B, T_tok, C) -> (B, C, T)."""
    if mc.compressed:
        r = mc.en_coder_compress_rate
        win = mc.en_coder_window_size + mc.en_coder_cache_size
        x = local_trans(w, f"{prefix}.local_trans", feature, win, mc.en_coder_depth - 2)
        x = F.interpolate(x.permute(0, 2, 1), scale_factor=r, mode="linear", align_corners=False).permute(0, 2, 1)
        x = local_trans(w, f"{prefix}.up_trans.trans", x, win * r, 2)
        return x.permute(0, 2, 1)
    return local_trans(w, f"{prefix}.local_trans", feature, mc.en_coder_window_size, mc.en_coder_depth).permute(0, 2, 1)


# =============================================================================================
# FSQ quantiser (reference l3ac/vq)
# =============================================================================================
def fsq_levels_basis(levels):
    """vq/fsq.py:14-16: int32 levels and basis = cumprod([1] + levels[:-1])."""
    lv = torch.tensor(list(levels), dtype=torch.int32)
    basis = torch.cumprod(torch.tensor([1] + list(levels)[:-1]), dim=0, dtype=torch.int32)
    return lv, basis


def fsq_quantize_act(act, levels):
    """vq/fsq.py:56-68 + :21 from the activation values: act (..., D) in [0, 1] (what tanh_act returns).
    Returns (q_z, indices int32, level_indices fp32)."""
    lv, basis = fsq_levels_basis(levels)
    shape = act.shape
    act = act.reshape(-1, shape[-1])
    li = (act * (lv - 1)).round()  # :59 special_edge; torch.round = half-to-even
    q_act = li / (lv - 1)  # :60
    idx = (li * basis).sum(dim=-1).to(torch.int32)  # :67-68
    q_z = q_act * 2 - 1  # :21
    return q_z.reshape(shape), idx.reshape(shape[:-1]), li.reshape(shape)


def fsq_quantize(z, levels):
    """vq/fsq.py:30-68 in eval mode (noise_rate -> 0).  z: (..., D) latents.
    Returns (q_z, indices int32, level_indices fp32)."""
    return fsq_quantize_act((torch.tanh(z) + 1) / 2, levels)  # fsq_act.py:38-39


def fsq_indices_to_codes(indices, levels):
    """vq/fsq.py:70-81: (idx // basis) % levels -> / (L-1) -> *2-1."""
    lv, basis = fsq_levels_basis(levels)
    li = (indices.unsqueeze(-1) // basis) % lv
    return (li / (lv - 1)) * 2 - 1


def quantizer(w, mc, x, prefix="quantizer"):
    """vq/__init__.py:25-30: project_in -> SuperFSQ -> project_out.  x: (B, T, feature_dim)."""
    lat = F.linear(x, w[f"{prefix}.project_in.weight"], w[f"{prefix}.project_in.bias"])
    q_z, idx, li = fsq_quantize(lat, mc.levels)
    q_feat = F.linear(q_z, w[f"{prefix}.project_out.weight"], w[f"{prefix}.project_out.bias"])
    return q_feat, {"indices": idx, "level_indices": li}, lat


def to_features(w, mc, indices, prefix="quantizer"):
    """vq/__init__.py:20-23."""
    codes = fsq_indices_to_codes(indices, mc.levels)
    return F.linear(codes, w[f"{prefix}.project_out.weight"], w[f"{prefix}.project_out.bias"])


def codebook(levels):
    """The implicit product codebook C[k] = indices_to_codes(k) (vq/fsq.py:80-81), (K, D) fp32."""
    k = math.prod(levels)
    return fsq_indices_to_codes(torch.arange(k, dtype=torch.int32), levels)


# =============================================================================================
# the drop-in surface (reference l3ac/__init__.py:108-121)
# =============================================================================================
def preprocess(mc, audio):
    """codec.py:79-84: right zero-pad to a multiple of hop_length."""
    length = audio.shape[-1]
    pad = math.ceil(length / mc.hop_length) * mc.hop_length - length
    return F.pad(audio, (0, pad)), length


@torch.inference_mode()
def encode_audio(w, mc, audio, taps=None):
    """__init__.py:108-114.  audio (B, T) -> (q_feature (B, T_tok, C), {"indices", "level_indices"})."""
    x, _ = preprocess(mc, audio)
    feature = encoder(w, mc, x.unsqueeze(1), taps=taps)
    trans = en_encoder(w, mc, feature)
    _tap(taps, "en_encoder.out", trans)
    q_feat, ind, lat = quantizer(w, mc, trans)
    _tap(taps, "latents", lat)
    return q_feat, ind


@torch.inference_mode()
def decode_audio(w, mc, audio_feature=None, indices=None, taps=None):
    """__init__.py:116-121."""
    if audio_feature is None:
        audio_feature = to_features(w, mc, indices)
    q = en_decoder(w, mc, audio_feature)
    _tap(taps, "en_decoder.out", q)
    return decoder(w, mc, q, taps=taps).squeeze(1)
