#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE (/root/reference) on CPU.

Runs only in the build container (the reference tree never travels to the GPU box); the ``.npz`` files it
writes are committed.  Usage:  python tests/golden/make_golden.py

What is pinned (see oracle/l3ac_oracle.py docstring):
  * ``*_conv.npz``   — reference ``Codec`` (encoder / quantizer / to_features / decoder) outputs, produced by
                       reference code only.  Weights come from l3ac_amd.weights.synthetic_state_dicts and are
                       loaded into the reference with ``load_state_dict(strict=True)`` (also pins the schema).
                       ``stress_*``: the same with ``synthetic_state_dicts(profile="stress")`` — heavy-tailed gains, snake alpha in
                       [0.05, 20], GRN gamma / beta of O(1), saturating latents: the statistics of a trained network.
  * ``*_e2e.npz``    — reference ``EnCodec`` wiring (l3ac/local_trans.py, en_codec.py) run end to end, with the
                       absent PyPI dependency ``local_attention`` replaced by a stand-in built on the oracle's
                       own restatement.  Pins the WIRING only; the attention arithmetic stays unpinned.
  * ``fsq_kat.npz``  — known-answer vectors from the reference ``SuperFSQ`` (half-even ties, saturation, decode).
  * ``binding_names.npz`` — INTEGRATION.md §2's ``folded_tensors()`` run on the reference ``EnCodec``: tensor names + checksums.
  * ``chunk_kat.npz`` — the reference's ``ChunkData`` cut / merge results and ``Codec.extract_unit`` / ``decode_unit`` run on the
                       tiny model (pins oracle/chunk_oracle.py).
"""
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")

import numpy as np
import pydantic
import torch
import torch.nn as nn

# --- harness stub: pydantic-settings is not installed here; l3ac/xtract/config.py only needs the names --------
_ps = types.ModuleType("pydantic_settings")
_ps.BaseSettings = type("BaseSettings", (pydantic.BaseModel,), {})
_ps.SettingsConfigDict = dict
_ps.PydanticBaseSettingsSource = object
_ps.TomlConfigSettingsSource = object
sys.modules["pydantic_settings"] = _ps

from oracle import l3ac_oracle as O  # noqa: E402


# --- harness stand-in for the absent `local_attention` package (wiring check only) ---------------------------
def _install_local_attention_standin():
    class DynamicPositionBias(nn.Module):
        def __init__(self, dim, heads):
            super().__init__()
            self.mlp = nn.Sequential(nn.Linear(1, dim), nn.SiLU(), nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, heads))

        def forward(self, i, j):
            w = {f"p.mlp.{k}": v for k, v in self.mlp.state_dict().items()}
            return O.dynamic_position_bias(w, "p", i, j)

    class LocalMHA(nn.Module):
        def __init__(self, *, dim, window_size, dim_head, heads, dropout, causal, prenorm, qk_rmsnorm,
                     use_xpos, xpos_scale_base, exact_windowsize, use_rotary_pos_emb):
            super().__init__()
            assert causal and prenorm and not qk_rmsnorm and not use_xpos and not exact_windowsize
            assert not use_rotary_pos_emb and heads == O.HEADS and dropout == 0.
            self.window_size = window_size
            self.norm = nn.LayerNorm(dim)
            self.to_qkv = nn.Linear(dim, dim_head * heads * 3, bias=False)
            self.to_out = nn.Linear(dim_head * heads, dim, bias=False)

        def forward(self, x, mask=None, attn_bias=None):
            assert mask is None
            w = {f"p.{k}": v for k, v in self.state_dict().items()}
            return O.local_mha(w, "p", x, self.window_size, attn_bias)

    class GEGLU(nn.Module):
        def forward(self, x):
            a, gate = x.chunk(2, dim=-1)
            return a * torch.nn.functional.gelu(gate)

    def FeedForward(dim, mult=4, dropout=0.):
        inner = int(dim * mult * 2 / 3)
        return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), GEGLU(), nn.Dropout(dropout),
                             nn.Linear(inner, dim, bias=False))

    pkg = types.ModuleType("local_attention")
    tr = types.ModuleType("local_attention.transformer")
    tr.DynamicPositionBias, tr.LocalMHA, tr.FeedForward = DynamicPositionBias, LocalMHA, FeedForward
    pkg.transformer = tr
    sys.modules["local_attention"] = pkg
    sys.modules["local_attention.transformer"] = tr


_install_local_attention_standin()

import l3ac.codec  # noqa: E402  (reference)
import l3ac.en_codec  # noqa: E402  (reference)

from l3ac_amd.config import L3ACConfig, resolve_config_file  # noqa: E402
from l3ac_amd import weights as W  # noqa: E402


def seeded_audio(batch, samples, seed=1234):
    """(rand * 2 - 1) * 0.5 from a CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(batch, samples, generator=g) * 2 - 1) * 0.5


def build_reference(cfg_file, seed, profile="mild"):
    cfg = L3ACConfig(config_file=cfg_file)
    mc = cfg.network_config
    ref_mc = l3ac.en_codec.ModelConfig(**mc.model_dump(exclude={"hop_length"}))
    assert ref_mc.hop_length == mc.hop_length
    ref = l3ac.en_codec.EnCodec(ref_mc).eval()
    sds = W.synthetic_state_dicts(mc, seed=seed, profile=profile)
    for name, module in ref.trainable_modules.items():
        want = {k: tuple(v.shape) for k, v in module.state_dict().items()}
        have = dict(W.raw_keys(mc, name))
        assert want == have, f"{name}: schema mismatch {set(want) ^ set(have)}"
        module.load_state_dict(sds[name], strict=True)
    return mc, ref, sds


def strided(t, n=4096):
    flat = t.reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].numpy().copy()


@torch.inference_mode()
def make_model_fixtures(tag, cfg_file, seed, batch, samples, full_tensors, profile="mild"):
    mc, ref, _ = build_reference(cfg_file, seed, profile)
    audio = seeded_audio(batch, samples)
    out = {"seed": np.int64(seed), "audio_seed": np.int64(1234), "batch": np.int64(batch), "samples": np.int64(samples)}
    if profile != "mild":  # (the mild fixtures carry no such key: a regeneration then has exactly the committed files' arrays)
        out["profile"] = np.array(profile)

    # ---- conv stacks + quantiser: reference code only ------------------------------------------------
    x, length = ref.preprocess(audio)
    hooks, taps = [], {}
    for name, m in list(ref.encoder.blocks.named_children()) + []:
        hooks.append(m.register_forward_hook(lambda _m, _i, o, n=name: taps.__setitem__(f"enc_block{n}", o)))
    feature = ref.encoder(x.unsqueeze(1))
    for h in hooks:
        h.remove()
    q_feat, ind, loss = ref.quantizer(feature.permute(0, 2, 1))  # Codec.forward order (codec.py:101-103)
    feats_from_idx = ref.quantizer.to_features(ind["indices"])
    lat = ref.quantizer.project_in(feature.permute(0, 2, 1))
    hooks, dtaps = [], {}
    for name, m in ref.decoder.blocks.named_children():
        hooks.append(m.register_forward_hook(lambda _m, _i, o, n=name: dtaps.__setitem__(f"dec_block{n}", o)))
    wave = ref.decoder(q_feat.permute(0, 2, 1)).squeeze(1)
    for h in hooks:
        h.remove()
    conv = dict(out)
    conv.update(padded_len=np.int64(x.shape[-1]), orig_len=np.int64(length),
                indices=ind["indices"].numpy(), level_indices=ind["level_indices"].numpy(),
                vq_loss=loss.numpy())
    assert torch.equal(feats_from_idx, q_feat), "to_features(indices) != q_feat"
    big = {"feature": feature, "latents": lat, "q_feat": q_feat, "wave": wave, **taps, **dtaps}
    for k, v in big.items():
        if full_tensors:
            conv[k] = v.numpy()
        else:
            conv[k + "_strided"] = strided(v)
            conv[k + "_sum"] = np.float64(v.double().sum().item())
            conv[k + "_abssum"] = np.float64(v.double().abs().sum().item())
            conv[k + "_shape"] = np.array(v.shape, dtype=np.int64)
    np.savez_compressed(HERE / f"{tag}_conv.npz", **conv)

    # ---- end to end through the reference's EnCodec wiring (stand-in attention) ----------------------
    feature = ref.encoder(x.unsqueeze(1))  # l3ac/__init__.py:108-114
    trans = ref.en_encoder(feature)
    q_trans, ind2, _ = ref.quantizer(trans)
    q_feature = ref.en_decoder(q_trans)  # l3ac/__init__.py:116-121
    wave2 = ref.decoder(q_feature).squeeze(1)
    wave_from_idx = ref.decoder(ref.en_decoder(ref.quantizer.to_features(ind2["indices"]))).squeeze(1)
    fwd = ref(audio)  # EnCodec.forward (en_codec.py:53-72)
    assert torch.equal(fwd["indices"], ind2["indices"])
    assert torch.equal(fwd["generated_audio"], wave2[..., :length])
    assert torch.equal(wave_from_idx, wave2)
    e2e = dict(out)
    e2e.update(indices=ind2["indices"].numpy(), level_indices=ind2["level_indices"].numpy())
    big = {"trans": trans, "latents": ref.quantizer.project_in(trans), "q_trans": q_trans,
           "q_feature": q_feature, "wave": wave2}
    for k, v in big.items():
        if full_tensors:
            e2e[k] = v.numpy()
        else:
            e2e[k + "_strided"] = strided(v)
            e2e[k + "_sum"] = np.float64(v.double().sum().item())
            e2e[k + "_abssum"] = np.float64(v.double().abs().sum().item())
            e2e[k + "_shape"] = np.array(v.shape, dtype=np.int64)
    np.savez_compressed(HERE / f"{tag}_e2e.npz", **e2e)
    hist = np.bincount(ind2["level_indices"].numpy().astype(np.int64).reshape(-1), minlength=max(mc.levels))
    if profile == "stress":  # the profile's promise (l3ac_amd/weights.py): a good part of the latents at an outermost level
        li, lv = ind2["level_indices"], torch.tensor(mc.levels, dtype=torch.float32)
        sat = float(((li == 0) | (li == lv - 1)).float().mean())
        print(f"[{tag}] outermost-level fraction {sat:.3f}")
        assert sat >= 0.10
    print(f"[{tag}] hop={mc.hop_length} tokens={ind2['indices'].shape} wave std={wave2.std():.4f} "
          f"|wave|max={wave2.abs().max():.4f} level hist={hist.tolist()}")


@torch.inference_mode()
def make_fsq_kat():
    from l3ac.vq.fsq import SuperFSQ  # reference
    out = {}
    for tag, levels in (("l7", [7] * 6), ("l9977", [9, 9, 9, 7, 7, 7]), ("even", [2, 4, 6, 8]), ("tiny", [5, 3, 4])):
        fsq = SuperFSQ(levels=levels, noise_rate=0.5).eval()
        d = len(levels)
        g = torch.Generator().manual_seed(7)
        rows = [torch.zeros(1, d), torch.full((1, d), 20.0), torch.full((1, d), -20.0),
                torch.full((1, d), float("inf")), torch.full((1, d), -float("inf")),
                torch.randn(4096, d, generator=g) * 1.5]
        z = torch.cat(rows)
        q, ind = fsq(z)
        k = int(torch.prod(torch.tensor(levels)))
        all_idx = torch.arange(k, dtype=torch.int32)
        sel = all_idx if k <= 4096 else all_idx[:: k // 4096]
        out[f"{tag}_levels"] = np.array(levels, dtype=np.int32)
        out[f"{tag}_z"] = z.numpy()
        out[f"{tag}_q"] = q.numpy()
        out[f"{tag}_indices"] = ind["indices"].numpy()
        out[f"{tag}_level_indices"] = ind["level_indices"].numpy()
        out[f"{tag}_dec_idx"] = sel.numpy()
        out[f"{tag}_dec_codes"] = fsq.indices_to_codes(sel).numpy()
        # closed form == nearest neighbour over the explicit codebook: record reference answers
        codes = fsq.indices_to_codes(all_idx)
        zq = torch.tanh(z[5:5 + 512])
        nn_idx = torch.cdist(zq, codes).argmin(dim=1).to(torch.int32)
        out[f"{tag}_nn_query"] = zq.numpy()
        out[f"{tag}_nn_idx"] = nn_idx.numpy()
    np.savez_compressed(HERE / "fsq_kat.npz", **out)
    print("[fsq_kat] written")


@torch.inference_mode()
def make_fsq_boundary_kat():
    """Rounding-boundary known answers from the reference's own ``SuperFSQ.quantize_act_value`` (vq/fsq.py:56-65),
    ``level_indices_to_indices`` (:67-68) and ``inv_act`` (:21), fed with activation values in [0, 1] (i.e. after
    ``tanh_act``, fsq_act.py:38-39, so that no transcendental sits between the constructed input and the rounding):
    for every level count L of a level set, every dimension and every k in [0, L-2], the fp32 value nearest to
    (k + 0.5) / (L - 1) and its +-1 and +-2 ulp neighbours, all other dimensions held at a generic value.  The products
    act * (L - 1) land exactly on k + 0.5 for some of them (torch.round goes half-to-even there) and one ulp to either
    side for the rest."""
    from l3ac.vq.fsq import SuperFSQ  # reference
    out = {}
    for tag, levels in (("l7", [7] * 6), ("l9977", [9, 9, 9, 7, 7, 7]), ("even", [2, 4, 6, 8]), ("tiny", [5, 3, 4])):
        fsq = SuperFSQ(levels=levels).eval()
        d = len(levels)
        rows = []
        for dim, L in enumerate(levels):
            for k in range(L - 1):
                centre = np.float32((k + 0.5) / (L - 1))
                cand = [centre]
                lo = hi = centre
                for _ in range(2):
                    lo = np.nextafter(lo, np.float32(-1.0), dtype=np.float32)
                    hi = np.nextafter(hi, np.float32(2.0), dtype=np.float32)
                    cand += [lo, hi]
                for c in cand:
                    row = np.full(d, 0.3137, dtype=np.float32)
                    row[dim] = c
                    rows.append(row)
        act = torch.from_numpy(np.stack(rows))
        q_act, li = fsq.quantize_act_value(act)
        idx = fsq.level_indices_to_indices(li)
        q = fsq.inv_act(q_act)
        prod = act * (fsq.levels - 1)
        n_tie = int(((prod - prod.floor()) == 0.5).any(dim=1).sum())
        print(f"[fsq_boundary {tag}] {act.shape[0]} rows, {n_tie} with an exact k+0.5 product")
        assert n_tie > 0
        out[f"{tag}_levels"] = np.array(levels, dtype=np.int32)
        out[f"{tag}_act"] = act.numpy()
        out[f"{tag}_level_indices"] = li.numpy()
        out[f"{tag}_indices"] = idx.numpy()
        out[f"{tag}_q"] = q.numpy()
    np.savez_compressed(HERE / "fsq_boundary_kat.npz", **out)
    print("[fsq_boundary_kat] written")


@torch.inference_mode()
def make_chunk_kat():
    """Long-audio bookkeeping, produced by the reference's own code: ``l3ac.codec.ChunkData`` (codec.py:159-188) cutting and
    merging integer sequences (ragged tails, chunk_len dividing / not dividing the length, a single chunk), and
    ``Codec.extract_unit`` / ``decode_unit`` (codec.py:124-156) run as written on the tiny model.

    Harness note: as written the method cannot run — ``Codec.compress`` (codec.py:113-116) returns the quantiser's ``indices``
    DICT, which ``extract_unit`` indexes with ``[0]`` (:141: KeyError with ``SuperFSQ``), and :144 reads ``self.hop_length``, which
    only the config has (AttributeError); both are recorded in the fixture as ``extract_unit_errors_as_written``.  To run the rest
    of the method as written, ``compress`` is wrapped so that it returns the dict's ``"indices"`` tensor and the instance is given
    its config's ``hop_length``; nothing else is touched."""
    from l3ac.codec import ChunkData  # reference
    out = {}
    cases = [(37, 10, 3), (40, 10, 3), (9, 10, 3), (10, 10, 1), (101, 25, 24), (64, 16, 1), (1, 5, 2), (23, 7, 6)]
    out["cd_cases"] = np.array(cases, dtype=np.int64)
    for i, (n, cl, pl) in enumerate(cases):
        data = torch.arange(n, dtype=torch.int64) * 3 + 1
        chunks = ChunkData(chunk_len=cl, prefix_len=pl, original_data=data).chunk_data
        out[f"cd{i}_lens"] = np.array([len(c) for c in chunks], dtype=np.int64)
        out[f"cd{i}_cat"] = torch.cat(chunks).numpy()
        merged = ChunkData(chunk_len=cl, prefix_len=pl, chunk_data=chunks).data
        assert torch.equal(merged, data)
        out[f"cd{i}_merged"] = merged.numpy()
        # merge of chunks whose contents are NOT slices of one sequence (what decode_unit does with decoded chunks)
        g = torch.Generator().manual_seed(100 + i)
        other = [torch.randint(0, 1000, (len(c), 2), generator=g) for c in chunks]
        out[f"cd{i}_other_cat"] = torch.cat(other).numpy()
        out[f"cd{i}_other_merged"] = ChunkData(chunk_len=cl, prefix_len=pl, chunk_data=other).data.numpy()
    # ---- extract_unit / decode_unit on the tiny model's conv codec (the base ``Codec``: the methods skip en_encoder / en_decoder,
    #      and on an ``EnCodec`` their token bookkeeping — process_window // hop with the EnCodec hop — does not match the tokens
    #      ``compress`` returns) -----------------------------------------------------------------------------------------
    mc, en_ref, sds = build_reference(HERE / "tiny.toml", 3)
    base_mc = l3ac.codec.ModelConfig(**mc.model_dump(exclude={"hop_length", "en_coder_depth", "en_coder_window_size", "en_coder_dynamic_pos",
                                                             "en_coder_compress_rate", "en_coder_cache_size"}))
    ref = l3ac.codec.Codec(base_mc).eval()
    for name, module in ref.trainable_modules.items():
        module.load_state_dict(sds[name], strict=True)
    audio = seeded_audio(1, 1000, seed=77)
    as_written = []
    try:
        ref.extract_unit(audio, process_window=300)
    except Exception as e:  # noqa: BLE001
        as_written.append(type(e).__name__)
    orig_compress = ref.compress

    def compress(x):
        ind, q = orig_compress(x)
        return (ind["indices"] if isinstance(ind, dict) else ind), q
    ref.compress = compress
    try:
        ref.extract_unit(audio, process_window=300)
    except Exception as e:  # noqa: BLE001
        as_written.append(type(e).__name__)
    ref.hop_length = base_mc.hop_length  # codec.py:144 reads self.hop_length, which only the config has
    out["extract_unit_errors_as_written"] = np.array(as_written)  # KeyError (dict indexed with [0], :141), AttributeError (:144)
    eu = []
    for j, (samples, window) in enumerate(((1000, 300), (1201, 500), (250, 5 * 16000), (960, 96), (2000, 333))):
        audio = seeded_audio(1, samples, seed=77 + j)
        ci, cq = ref.extract_unit(audio, process_window=window)
        wave = ref.decode_unit(chunk_indices=ci)
        wave_q = ref.decode_unit(chunk_q_feature=cq)
        assert torch.equal(wave, wave_q)
        eu.append((samples, window))
        out[f"eu{j}_chunk_len"] = np.int64(ci.chunk_len)
        out[f"eu{j}_prefix_len"] = np.int64(ci.prefix_len)
        out[f"eu{j}_chunk_tokens"] = np.array([len(c) for c in ci.chunk_data], dtype=np.int64)
        out[f"eu{j}_indices"] = ci.data.numpy()
        out[f"eu{j}_q_feature"] = cq.data.numpy()
        out[f"eu{j}_wave"] = wave.numpy()
    out["eu_cases"] = np.array(eu, dtype=np.int64)
    out["eu_seed"] = np.int64(3)
    np.savez_compressed(HERE / "chunk_kat.npz", **out)
    print(f"[chunk_kat] written ({len(cases)} ChunkData cases, {len(eu)} extract_unit cases; extract_unit as written raises: {as_written})")


def make_binding_fixture():
    """INTEGRATION.md §2's `folded_tensors()` — the reference-side stub a maintainer would add — run as written on the reference's
    own ``EnCodec`` (weight-norm parametrizations, ``trainable_modules``: en_codec.py:46-51, layers.py:11-25): the names it hands
    to ``l3ac_create`` and a checksum of every folded tensor.  tests/test_host.py asserts that ``l3ac_amd.weights.folded_weights``
    (the fold the package applies to the ``.pt`` files) yields exactly these, i.e. that the library looks up the names the snippet
    produces."""
    from l3ac_amd import _capi, build
    from tests.helpers import integration_snippet
    if not _capi.LIB_PATH.exists():  # the snippet dlopens the library as it stands in INTEGRATION.md: a clean checkout builds it first
        build.build_library()
    ns = integration_snippet()
    out = {}
    for tag, cfg_file, seed in (("tiny", HERE / "tiny.toml", 3), ("1kbps", resolve_config_file("1kbps"), 0),
                                ("3kbps", resolve_config_file("3kbps"), 0)):
        mc, ref, _ = build_reference(cfg_file, seed)
        tensors = ns["folded_tensors"](ref)
        names = sorted(tensors)
        out[f"{tag}_seed"] = np.int64(seed)
        out[f"{tag}_names"] = np.array(names)
        out[f"{tag}_numel"] = np.array([tensors[k].numel() for k in names], dtype=np.int64)
        out[f"{tag}_sum"] = np.array([tensors[k].double().sum().item() for k in names], dtype=np.float64)
        out[f"{tag}_abssum"] = np.array([tensors[k].double().abs().sum().item() for k in names], dtype=np.float64)
        # reference ModelConfig attributes HipPath.__init__ reads
        for attr in ("feature_dim", "encoder_dims", "encoder_depths", "compress_rates", "decoder_dims", "decoder_depths", "decode_rates",
                     "en_coder_depth", "en_coder_window_size", "en_coder_compress_rate", "hop_length"):
            out[f"{tag}_mc_{attr}"] = np.array(getattr(ref.mc, attr), dtype=np.int64)
        out[f"{tag}_mc_levels"] = np.array(ref.mc.vq_config["levels"], dtype=np.int64)
        print(f"[binding {tag}] {len(names)} tensors, {int(out[f'{tag}_numel'].sum())} values")
    np.savez_compressed(HERE / "binding_names.npz", **out)
    print("[binding_names] written")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--binding-only" in sys.argv:
        make_binding_fixture()
        sys.exit(0)
    if "--boundary-only" in sys.argv:
        make_fsq_boundary_kat()
        sys.exit(0)
    if "--chunk-only" in sys.argv:
        make_chunk_kat()
        sys.exit(0)
    if "--stress-only" in sys.argv:
        for name in ("1kbps", "3kbps"):
            make_model_fixtures(f"stress_{name}", resolve_config_file(name), seed=0, batch=2, samples=16000, full_tensors=False, profile="stress")
        sys.exit(0)
    make_chunk_kat()
    make_binding_fixture()
    make_fsq_kat()
    make_fsq_boundary_kat()
    make_model_fixtures("tiny", HERE / "tiny.toml", seed=3, batch=2, samples=250, full_tensors=True)
    make_model_fixtures("1kbps", resolve_config_file("1kbps"), seed=0, batch=2, samples=16000, full_tensors=False)
    make_model_fixtures("3kbps", resolve_config_file("3kbps"), seed=0, batch=2, samples=16000, full_tensors=False)
    for name in ("1kbps", "3kbps"):  # trained-weight statistics stand-in (weights.synthetic_state_dicts(profile="stress"))
        make_model_fixtures(f"stress_{name}", resolve_config_file(name), seed=0, batch=2, samples=16000, full_tensors=False, profile="stress")
