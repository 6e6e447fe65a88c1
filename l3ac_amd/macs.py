"""Analytic multiply-accumulate count of the encode -> quantize -> decode path (the figure reference
l3ac/__init__.py:28-51 obtains by tracing the model with ptflops; here derived from the geometry alone).

Counts every conv / linear product (weight-normed or plain) and, separately, the local transformer's linear layers and
its causal-attention products.  SURVEY.md Appendix A pins the conv/linear part: 3 831.3 MMAC per 1 s clip at 1kbps,
3 365.8 MMAC at 3kbps (tests/test_host.py).
"""
from __future__ import annotations

import math

from .weights import HEADS, en_decoder_layout, en_encoder_layout, trans_geometry


def _conv_unit(c: int, t: int) -> int:
    """modules.py:10-41: depth-wise k7 (7C) + Linear C -> 4C + Linear 4C -> C per frame."""
    return t * (7 * c + 8 * c * c)


def path_macs(mc, samples: int) -> dict:
    """MACs of one clip of `samples` samples (right-padded to a hop multiple, codec.py:79-84)."""
    hop = mc.hop_length
    t = math.ceil(samples / hop) * hop
    conv = t * (5 * 4 * 7 + 20 * 80 + 81 * mc.encoder_dims[0])  # FirstBlock: trend convs, conv_1, conv_2 (tconv/__init__.py:8-27)
    for i, c in enumerate(mc.encoder_dims):
        conv += mc.encoder_depths[i] * _conv_unit(c, t)
        if i + 1 < len(mc.encoder_dims):  # down layer Conv1d(k = stride) (modules.py:96-99)
            s = mc.compress_rates[i]
            t //= s
            conv += t * s * c * mc.encoder_dims[i + 1]
    conv += t * 3 * mc.encoder_dims[-1] * mc.feature_dim  # encoder tail k3 (modules.py:110)
    frames = t
    dim = mc.feature_dim
    dh, inner, ffi = trans_geometry(dim)
    lin_per_token = 3 * inner * dim + inner * dim + 2 * ffi * dim + ffi * dim  # to_qkv, to_out, ff.1 (GEGLU), ff.4
    trans_lin = trans_attn = 0

    def layer(n: int):
        nonlocal trans_lin, trans_attn
        trans_lin += n * lin_per_token
        trans_attn += HEADS * dh * n * (n + 1)  # QK^T and PV over the causal triangle (T <= window)

    n = frames
    for prefix, _, depth in en_encoder_layout(mc):
        for _ in range(depth):
            layer(n)
        if prefix == "down_trans.trans":  # DownTrans.down_layer (local_trans.py:136)
            n //= mc.en_coder_compress_rate
            trans_lin += n * mc.en_coder_compress_rate * dim * dim
    conv += n * 2 * len(mc.levels) * dim  # project_in / project_out (vq/__init__.py:14-15)
    for prefix, _, depth in en_decoder_layout(mc):
        if prefix == "up_trans.trans":
            n *= mc.en_coder_compress_rate
        for _ in range(depth):
            layer(n)
    t = frames
    conv += t * 3 * dim * mc.decoder_dims[0]  # decoder head k3 (modules.py:150)
    for i, s in enumerate(mc.decode_rates):
        c = mc.decoder_dims[i]
        conv += mc.decoder_depths[i] * _conv_unit(c, t)
        conv += t * (4 * 7 + 4 * c)                 # EnhanceBlock: 4 trend convs k7 + merge conv 4 -> C (tconv/__init__.py:30-44)
        conv += t * c * mc.decoder_dims[i + 1]      # up layer 1x1 (modules.py:161)
        t *= s
    c = mc.decoder_dims[-1]
    conv += t * (3 * (7 * c * c + c * c) + 7 * c)   # 3 LegacyUnits + head conv (modules.py:47-64, :193)
    return {"conv_linear": conv, "transformer_linear": trans_lin, "attention": trans_attn,
            "total": conv + trans_lin + trans_attn}
