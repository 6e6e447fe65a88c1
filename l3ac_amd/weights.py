"""Weight files of the L3AC hot path: schema, loading, weight-norm folding, seeded synthetic weights.

On-disk contract (reference l3ac/xtract/nn/module.py:36-54, l3ac/__init__.py:70-72): one plain
``state_dict`` per sub-module, ``{model_dir}/{name}.{version}/{encoder,quantizer,decoder,en_encoder,
en_decoder}.pt``.  Weight-normed layers (reference l3ac/layers.py:11-25) are stored as
``<m>.parametrizations.weight.original0`` (g) / ``original1`` (v); the hot path folds them ONCE at load
time (``w = g * v / ||v||``, the reference recomputes it on every forward).

The schema below is written from the module structure (reference modules.py:71-201, tconv/__init__.py,
local_trans.py, vq/__init__.py) and is checked key-for-key against the reference's own ``state_dict()``
by tests/golden/make_golden.py.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterator

import torch

MODULE_NAMES = ("encoder", "quantizer", "decoder", "en_encoder", "en_decoder")  # en_codec.py:46-51

# transformer geometry fixed by LocalTrans.builder (reference local_trans.py:50-53)
HEADS = 6
FF_MULT = 4


def trans_geometry(dim: int):
    dim_head = dim // 4
    inner = HEADS * dim_head
    ff_inner = int(dim * FF_MULT * 2 / 3)  # local_attention FeedForward
    return dim_head, inner, ff_inner


# ---------------------------------------------------------------------------------------------
# schema: (key, shape, kind).  kind selects the synthetic initialiser; "wn" entries expand to g/v/bias.
# ---------------------------------------------------------------------------------------------
def _wn(prefix, shape):
    yield prefix, tuple(shape), "wn"


def _conv_unit(prefix, c):
    yield from _wn(f"{prefix}.dw_conv", (c, 1, 7))
    yield f"{prefix}.norm.weight", (c,), "norm_w"
    yield f"{prefix}.norm.bias", (c,), "norm_b"
    yield from _wn(f"{prefix}.pw_conv1", (4 * c, c))
    yield f"{prefix}.act.alpha", (1, 1, 4 * c), "alpha"
    yield f"{prefix}.grn.gamma", (1, 4 * c), "grn"
    yield f"{prefix}.grn.beta", (1, 4 * c), "grn"
    yield from _wn(f"{prefix}.pw_conv2", (c, 4 * c))


def encoder_schema(mc) -> Iterator[tuple]:
    dims, depths, strides = mc.encoder_dims, mc.encoder_depths, mc.compress_rates
    for i in range(5):  # FirstBlock: 5 trend branches (tconv/__init__.py:25-27)
        yield from _wn(f"blocks.0.blocks.{i}.1", (4, 1, 7))
    yield from _wn("blocks.0.conv_1", (80, 20, 1))
    yield from _wn("blocks.0.conv_2", (dims[0], 81, 1))
    b = 1
    for i, (ci, co, s) in enumerate(zip(dims[:-1], dims[1:], strides)):
        for j in range(depths[i]):
            yield from _conv_unit(f"blocks.{b}.{j}.module", ci)
        yield from _wn(f"blocks.{b + 1}.0", (co, ci, s))
        yield f"blocks.{b + 1}.1.weight", (co,), "norm_w"
        yield f"blocks.{b + 1}.1.bias", (co,), "norm_b"
        b += 2
    for j in range(depths[-1]):
        yield from _conv_unit(f"blocks.{b}.{j}.module", dims[-1])
    yield from _wn(f"blocks.{b + 1}", (mc.feature_dim, dims[-1], 3))


def decoder_schema(mc) -> Iterator[tuple]:
    dims, depths, strides = mc.decoder_dims, mc.decoder_depths, mc.decode_rates
    yield from _wn("blocks.0", (dims[0], mc.feature_dim, 3))
    b = 1
    for i, (ci, co, s) in enumerate(zip(dims[:-1], dims[1:], strides)):
        for j in range(depths[i]):
            yield from _conv_unit(f"blocks.{b}.{j}.module", ci)
        for p in range(4):  # EnhanceBlock trend convs (tconv/__init__.py:30-37)
            yield from _wn(f"blocks.{b + 1}.blocks.{p}.1", (1, 1, 7))
        yield f"blocks.{b + 1}.merge_layer.0.weight", (4,), "norm_w"
        yield f"blocks.{b + 1}.merge_layer.0.bias", (4,), "norm_b"
        yield f"blocks.{b + 1}.merge_layer.1.weight", (ci, 4, 1), "plain_w"
        yield f"blocks.{b + 1}.merge_layer.1.bias", (ci,), "plain_b:4"
        yield from _wn(f"blocks.{b + 2}.0", (co, ci, 1))
        yield f"blocks.{b + 2}.2.weight", (co,), "norm_w"
        yield f"blocks.{b + 2}.2.bias", (co,), "norm_b"
        b += 3
    c = dims[-1]
    for u in range(3):  # three LegacyUnits, dilation 1/3/9 (modules.py:174-179)
        p = f"blocks.{b}.block.0.{u}.module.block"
        yield f"{p}.0.alpha", (1, c, 1), "alpha"
        yield from _wn(f"{p}.1", (c, c, 7))
        yield f"{p}.2.alpha", (1, c, 1), "alpha"
        yield from _wn(f"{p}.3", (c, c, 1))
    yield f"blocks.{b}.block.1.alpha", (1, c, 1), "alpha"
    yield from _wn(f"blocks.{b}.block.2", (1, c, 7))


def quantizer_schema(mc) -> Iterator[tuple]:
    d, f = len(mc.levels), mc.feature_dim
    yield "project_in.weight", (d, f), "plain_w"
    yield "project_in.bias", (d,), f"plain_b:{f}"
    yield "project_out.weight", (f, d), "plain_w"
    yield "project_out.bias", (f,), f"plain_b:{d}"


def _local_trans(prefix, dim, depth):
    _, inner, ffi = trans_geometry(dim)
    for l in range(depth):
        a, f = f"{prefix}.layers.{l}.0", f"{prefix}.layers.{l}.1"
        yield f"{a}.norm.weight", (dim,), "norm_w"
        yield f"{a}.norm.bias", (dim,), "norm_b"
        yield f"{a}.to_qkv.weight", (3 * inner, dim), "plain_w"
        yield f"{a}.to_out.weight", (dim, inner), "plain_w"
        yield f"{f}.0.weight", (dim,), "norm_w"
        yield f"{f}.0.bias", (dim,), "norm_b"
        yield f"{f}.1.weight", (2 * ffi, dim), "plain_w"
        yield f"{f}.4.weight", (dim, ffi), "plain_w"
    h = dim // 2  # DynamicPositionBias(dim=dim // 2, heads) (local_trans.py:30)
    m = f"{prefix}.dynamic_pos_bias.mlp"
    yield f"{m}.0.weight", (h, 1), "plain_w"
    yield f"{m}.0.bias", (h,), "plain_b:1"
    yield f"{m}.2.weight", (h, h), "plain_w"
    yield f"{m}.2.bias", (h,), f"plain_b:{h}"
    yield f"{m}.4.weight", (HEADS, h), "plain_w"
    yield f"{m}.4.bias", (HEADS,), f"plain_b:{h}"


def en_encoder_layout(mc):
    """[(prefix, window, depth)] of the LocalTrans stacks in execution order (local_trans.py:145-165, 56-74)."""
    if mc.compressed:
        first = 3 // 2  # depth is fixed to 3 for the compressed encoder (en_codec.py:35)
        w = mc.en_coder_window_size + mc.en_coder_cache_size
        return [("down_trans.trans", w * mc.en_coder_compress_rate, first), ("local_trans", w, 3 - first)]
    return [("local_trans", mc.en_coder_window_size, 1)]  # depth fixed to 1 (en_codec.py:27)


def en_decoder_layout(mc):
    if mc.compressed:
        w = mc.en_coder_window_size + mc.en_coder_cache_size
        return [("local_trans", w, mc.en_coder_depth - 2), ("up_trans.trans", w * mc.en_coder_compress_rate, 2)]
    return [("local_trans", mc.en_coder_window_size, mc.en_coder_depth)]


def en_encoder_schema(mc) -> Iterator[tuple]:
    dim = mc.feature_dim
    for prefix, _, depth in en_encoder_layout(mc):
        yield from _local_trans(prefix, dim, depth)
        if prefix == "down_trans.trans":
            yield from _wn("down_trans.down_layer", (dim, dim, mc.en_coder_compress_rate))


def en_decoder_schema(mc) -> Iterator[tuple]:
    for prefix, _, depth in en_decoder_layout(mc):
        yield from _local_trans(prefix, mc.feature_dim, depth)


SCHEMAS = {
    "encoder": encoder_schema,
    "quantizer": quantizer_schema,
    "decoder": decoder_schema,
    "en_encoder": en_encoder_schema,
    "en_decoder": en_decoder_schema,
}


def raw_keys(mc, module: str) -> list[tuple[str, tuple]]:
    """Keys/shapes exactly as they appear in the reference's ``{module}.pt``."""
    out = []
    for key, shape, kind in SCHEMAS[module](mc):
        if kind == "wn":
            out.append((f"{key}.bias", (shape[0],)))
            out.append((f"{key}.parametrizations.weight.original0", (shape[0],) + (1,) * (len(shape) - 1)))
            out.append((f"{key}.parametrizations.weight.original1", shape))
        else:
            out.append((key, shape))
    return out


# ---------------------------------------------------------------------------------------------
# seeded synthetic weights (no network on either box: SURVEY F5)
# ---------------------------------------------------------------------------------------------
def synthetic_state_dicts(mc, seed: int = 0, gain: float = 0.8, profile: str = "mild") -> dict[str, dict[str, torch.Tensor]]:
    """Deterministic random weights in the reference's file format.

    ``profile="mild"`` (every fixture and test of rounds 1-3): weight-normed tensors follow the reference initialiser
    (trunc-normal std .02, layers.py:15) but with a perturbed gain ``g`` and non-zero biases; parameters the reference
    initialises to 0/1 (norm affine, snake alpha, GRN gamma/beta) are perturbed too, so that a bug in any of them is visible
    in parity tests.

    ``profile="stress"`` stands in for the statistics of a TRAINED network, which the mild profile does not reach (no pretrained
    weights can be had offline): weight-norm gains with a heavy tail (a quarter of the output channels x 4), snake alpha
    log-uniform in [0.05, 20] (layers.py:29-47 — far from its init of 1: large sine arguments and large 1/alpha), GRN gamma / beta
    ~ N(0, 1) (layers.py:101-102 initialises them to 0: the mild profile's 0.1 leaves GRN nearly an identity), heavy-tailed
    biases, norm affines far from (1, 0), and a quantiser ``project_in`` scaled up until a good part of the latents sit in
    tanh's saturation (vq/fsq_act.py:38-39), i.e. at the outermost levels."""
    if profile not in ("mild", "stress"):
        raise ValueError(f"unknown weight profile {profile!r}")
    stress = profile == "stress"
    gen = torch.Generator(device="cpu").manual_seed(1_000_003 * (seed + 1) + (7919 if stress else 0))

    def randn(shape, std=1.0):
        return torch.randn(shape, generator=gen, dtype=torch.float32) * std

    def uniform(shape, bound):
        return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound

    def rand(shape):
        return torch.rand(shape, generator=gen, dtype=torch.float32)

    out = {}
    for module in MODULE_NAMES:
        sd = {}
        for key, shape, kind in SCHEMAS[module](mc):
            if kind == "wn":
                v = randn(shape, 0.02).clamp_(-0.04, 0.04)
                # ||w_row|| = g: a gain near 1 keeps a unit-variance input at roughly unit variance
                g_shape = (shape[0],) + (1,) * (len(shape) - 1)
                g = gain * (0.75 + 0.5 * rand(g_shape))
                bias = randn((shape[0],), 0.05)
                if stress:
                    # a quarter of the output channels at 4 x the gain of the rest, the layer's rms gain unchanged (x 4 on top of
                    # the mild gains compounds to 1e2 at the quantiser and a waveform that is 90 % tanh-saturated)
                    g = g * torch.where(rand(g_shape) < 0.25, 4.0, 1.0) / (0.75 + 0.25 * 16.0) ** 0.5
                    bias = 0.4 * bias * (1.0 + 9.0 * (rand((shape[0],)) < 0.1).float()) / rand((shape[0],)).clamp_min(0.05).sqrt()
                sd[f"{key}.bias"] = bias
                sd[f"{key}.parametrizations.weight.original0"] = g.contiguous()
                sd[f"{key}.parametrizations.weight.original1"] = v.contiguous()
            elif kind == "norm_w":
                sd[key] = 1.0 + randn(shape, 0.5 if stress else 0.1)
            elif kind == "norm_b":
                sd[key] = randn(shape, 0.1 if stress else 0.05)
            elif kind == "alpha":
                if stress:
                    lo, hi = torch.log(torch.tensor(0.05)), torch.log(torch.tensor(20.0))
                    sd[key] = torch.exp(lo + (hi - lo) * rand(shape))
                else:
                    sd[key] = (1.0 + randn(shape, 0.25)).clamp_(0.3, 2.5)
            elif kind == "grn":
                # stress: gamma ~ N(0, 1); beta ~ N(0, 0.3) — a per-channel constant of O(1) on 4C hidden channels drowns the signal
                # (measured: frame-to-frame variation of the encoder output 0.13 against channel means of 1.1)
                sd[key] = randn(shape, (1.0 if key.endswith("gamma") else 0.3) if stress else 0.1)
            elif kind == "plain_w":
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                sd[key] = uniform(shape, fan_in ** -0.5)
            elif kind.startswith("plain_b:"):
                sd[key] = uniform(shape, int(kind.split(":")[1]) ** -0.5)
            else:  # pragma: no cover
                raise AssertionError(kind)
        if stress and module == "quantizer":
            sd["project_in.weight"] = sd["project_in.weight"] * STRESS_PROJECT_IN_SCALE
        out[module] = sd
    return out


# project_in (vq/__init__.py:14) scale of the stress profile: chosen so that >= 10 % of the latents of the seeded 1kbps / 3kbps
# models land on an outermost level (tests/golden/make_golden.py prints and asserts the fraction)
STRESS_PROJECT_IN_SCALE = 1.5


# ---------------------------------------------------------------------------------------------
# load / save / fold
# ---------------------------------------------------------------------------------------------
def save_state_dicts(state_dicts, model_path) -> None:
    model_path = Path(model_path)
    model_path.mkdir(parents=True, exist_ok=True)
    for name, sd in state_dicts.items():
        torch.save(sd, model_path / f"{name}.pt")


def load_state_dicts(model_path, mc=None) -> dict[str, dict[str, torch.Tensor]]:
    """Read the five ``{module}.pt`` files.  Unlike the reference (which only logs a missing file and keeps
    random weights, xtract/nn/module.py:52-54) a missing or incomplete file is an error here."""
    model_path = Path(model_path)
    out = {}
    for name in MODULE_NAMES:
        f = model_path / f"{name}.pt"
        if not f.exists():
            raise FileNotFoundError(f"weight file missing: {f}")
        out[name] = torch.load(f, map_location="cpu", weights_only=True)
    if mc is not None:
        check_state_dicts(out, mc)
    return out


def check_state_dicts(state_dicts, mc) -> None:
    for module in MODULE_NAMES:
        sd = state_dicts[module]
        want = dict(raw_keys(mc, module))
        missing = sorted(set(want) - set(sd))
        extra = sorted(set(sd) - set(want))
        if missing or extra:
            raise KeyError(f"{module}: missing keys {missing[:4]}... unexpected keys {extra[:4]}...")
        for k, shape in want.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{module}.{k}: shape {tuple(sd[k].shape)} != {shape}")


_G, _V = ".parametrizations.weight.original0", ".parametrizations.weight.original1"


def fold_state_dict(sd: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """Fold weight-norm: ``<m>.weight = torch._weight_norm(v, g, 0)`` — the very op the reference's
    parametrisation evaluates on each forward (torch.nn.utils.parametrizations._WeightNorm.forward)."""
    out = {}
    for k, t in sd.items():
        t = t.detach().to(torch.float32).cpu()
        if k.endswith(_G):
            base = k[: -len(_G)]
            out[base + ".weight"] = torch._weight_norm(sd[base + _V].float().cpu(), t, 0).contiguous()
        elif k.endswith(_V):
            continue
        else:
            out[k] = t.contiguous()
    return out


def folded_weights(state_dicts) -> dict[str, torch.Tensor]:
    """Flat ``{module}.{key}`` → fp32 CPU tensor with weight-norm folded (what the C-ABI consumes)."""
    flat = {}
    for module in MODULE_NAMES:
        for k, t in fold_state_dict(state_dicts[module]).items():
            flat[f"{module}.{k}"] = t
    return flat
