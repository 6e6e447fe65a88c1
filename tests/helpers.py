"""Shared helpers for the parity tests (test infrastructure)."""
from pathlib import Path

import numpy as np
import torch

from l3ac_amd import weights as W
from l3ac_amd.config import L3ACConfig, resolve_config_file

GOLDEN = Path(__file__).resolve().parent / "golden"


def seeded_audio(batch, samples, seed=1234):
    """(rand * 2 - 1) * 0.5 from a CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(batch, samples, generator=g) * 2 - 1) * 0.5


STRUCTURED_KINDS = ("sweep", "sine", "harmonics", "burst_after_silence", "dc_offset", "clipped", "quiet", "loud")


def structured_audio(per_kind, samples, seed=4321, sample_rate=16000):
    """Inputs white noise does not reach (TrendPool / EnhanceBlock / InstanceNorm branches, snake at large arguments, the
    zero-pad edge): `per_kind` clips of each of STRUCTURED_KINDS, deterministic in `seed`.  Returns (audio [8 * per_kind,
    samples] fp32, kinds list)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.arange(samples, dtype=torch.float64) / sample_rate
    dur = samples / sample_rate
    clips, kinds = [], []
    for kind in STRUCTURED_KINDS:
        for i in range(per_kind):
            u = torch.rand(4, generator=g, dtype=torch.float64)
            noise = torch.rand(samples, generator=g, dtype=torch.float64) * 2 - 1
            if kind == "sweep":  # linear chirp f0 -> f1, 0.5 amplitude
                f0, f1 = 40 + 400 * u[0], 2000 + 5500 * u[1]
                x = 0.5 * torch.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t) + 6.28 * u[2])
            elif kind == "sine":
                x = (0.1 + 0.8 * u[1]) * torch.sin(2 * np.pi * (60 + 3000 * u[0] ** 2) * t + 6.28 * u[2])
            elif kind == "harmonics":  # voiced-speech-like: decaying harmonics of a 90-300 Hz fundamental, slow amplitude modulation
                f = 90 + 210 * u[0]
                x = sum((0.6 ** h) * torch.sin(2 * np.pi * f * (h + 1) * t + h) for h in range(10))
                x = 0.3 * x * (0.6 + 0.4 * torch.sin(2 * np.pi * (2 + 4 * u[1]) * t))
            elif kind == "burst_after_silence":  # digital silence, then a noise burst with a hard onset, silence again
                on = int((0.2 + 0.5 * u[0]) * samples)
                off = min(samples, on + int((0.05 + 0.2 * u[1]) * samples))
                x = torch.zeros(samples, dtype=torch.float64)
                x[on:off] = 0.7 * noise[on:off]
            elif kind == "dc_offset":
                x = (0.45 if i % 2 == 0 else -0.45) + 0.1 * noise
            elif kind == "clipped":  # hard-clipped at +-1.0
                x = (3.0 * (0.5 * noise + 0.5 * torch.sin(2 * np.pi * (100 + 900 * u[0]) * t))).clamp(-1.0, 1.0)
            elif kind == "quiet":
                x = 0.005 * noise
            else:  # loud: full-scale noise
                x = 1.0 * noise
            clips.append(x.to(torch.float32))
            kinds.append(kind)
    return torch.stack(clips), kinds


def load_case(tag):
    """(mc, folded weights, conv fixture, e2e fixture) for a golden case; weights regenerate from the seed."""
    name = tag[len("stress_"):] if tag.startswith("stress_") else tag  # stress_<config>: the trained-statistics weight profile
    cfg_file = GOLDEN / "tiny.toml" if name == "tiny" else resolve_config_file(name)
    mc = L3ACConfig(config_file=cfg_file).network_config
    conv = np.load(GOLDEN / f"{tag}_conv.npz")
    e2e = np.load(GOLDEN / f"{tag}_e2e.npz")
    profile = str(conv["profile"]) if "profile" in conv.files else "mild"
    assert profile == ("stress" if tag.startswith("stress_") else "mild")
    sds = W.synthetic_state_dicts(mc, seed=int(conv["seed"]), profile=profile)
    return mc, W.folded_weights(sds), conv, e2e


def strided(t, n=4096):
    flat = t.reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n]


def index_mismatch_report(idx, idx_ref, latents_ref, levels, tau):
    """Compare quantiser indices.  Every mismatch must be a +-1 step in level space caused by a latent whose
    act*(L-1) lies within `tau` of a rounding boundary (SURVEY §7 'hard parts').  Returns (n_mismatch, ok)."""
    idx = np.asarray(idx).reshape(-1).astype(np.int64)
    idx_ref = np.asarray(idx_ref).reshape(-1).astype(np.int64)
    bad = np.nonzero(idx != idx_ref)[0]
    if bad.size == 0:
        return 0, True
    lv = np.asarray(levels, dtype=np.int64)
    basis = np.concatenate([[1], np.cumprod(lv[:-1])])
    lat = np.asarray(latents_ref, dtype=np.float64).reshape(-1, len(lv))
    ok = True
    for r in bad:
        li = (idx[r] // basis) % lv
        li_ref = (idx_ref[r] // basis) % lv
        diff = np.nonzero(li != li_ref)[0]
        scaled = (np.tanh(lat[r]) + 1) / 2 * (lv - 1)
        margin = np.abs(np.abs(scaled - np.floor(scaled)) - 0.5)
        for d in diff:
            if abs(li[d] - li_ref[d]) != 1 or margin[d] > tau:
                ok = False
    return int(bad.size), ok


def index_agreement(idx, idx_ref, latents_ref, levels):
    """Full accounting of quantiser-index agreement against the oracle: every token compared, and for every mismatch
    the distance of the responsible latent from its rounding boundary, |frac(act * (L - 1)) - 0.5| in level units
    (evaluated in fp64 from the ORACLE's latents).  `single_step` is False if any mismatching token differs by more than
    one level in some dimension."""
    idx = np.asarray(idx).reshape(-1).astype(np.int64)
    idx_ref = np.asarray(idx_ref).reshape(-1).astype(np.int64)
    lv = np.asarray(levels, dtype=np.int64)
    basis = np.concatenate([[1], np.cumprod(lv[:-1])])
    lat = np.asarray(latents_ref, dtype=np.float64).reshape(-1, len(lv))
    bad = np.nonzero(idx != idx_ref)[0]
    max_margin, single = 0.0, True
    for r in bad:
        li = (idx[r] // basis) % lv
        li_ref = (idx_ref[r] // basis) % lv
        scaled = (np.tanh(lat[r]) + 1) / 2 * (lv - 1)
        margin = np.abs(np.abs(scaled - np.floor(scaled)) - 0.5)
        for d in np.nonzero(li != li_ref)[0]:
            single = single and abs(int(li[d]) - int(li_ref[d])) == 1
            max_margin = max(max_margin, float(margin[d]))
    # how close the batch's latents come to a boundary at all (context for the mismatch count)
    scaled = (np.tanh(lat) + 1) / 2 * (lv - 1)
    margins = np.abs(np.abs(scaled - np.floor(scaled)) - 0.5)
    return {"tokens": int(idx.size), "mismatches": int(bad.size), "mismatch_positions": [int(r) for r in bad[:8]],  # flat token numbers (clip * tokens + token)
            "max_margin_of_mismatches": max_margin,
            "single_step": bool(single), "decisions_within_1e-4": int((margins < 1e-4).sum()),
            "decisions_within_1e-5": int((margins < 1e-5).sum()), "min_margin": float(margins.min())}


def integration_snippet():
    """INTEGRATION.md §2's reference-side binding (`l3ac/_hip.py`), extracted from the document and executed as written — the only
    adaptation is where `ctypes.CDLL("libl3ac_hip.so")` finds the library (the in-tree build instead of the loader path).
    Returns the namespace the block defines (`_Cfg`, `_Tensor`, `folded_tensors`, `HipPath`, ...)."""
    import ctypes
    import re

    from l3ac_amd import _capi
    text = (Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    section = text[text.index("## 2."):text.index("## 3.")]
    code = re.search(r"```python\n(# l3ac/_hip\.py.*?)```", section, re.S).group(1)
    real_cdll = ctypes.CDLL

    def cdll(name, *a, **k):
        return real_cdll(str(_capi.LIB_PATH) if name == "libl3ac_hip.so" else name, *a, **k)
    ns = {"__name__": "l3ac._hip"}
    ctypes.CDLL = cdll
    try:
        exec(compile(code, "INTEGRATION.md#l3ac/_hip.py", "exec"), ns)
    finally:
        ctypes.CDLL = real_cdll
    return ns


def module_tree_from_state_dict(sd):
    """An nn.Module whose ``state_dict()`` has exactly the keys / shapes of `sd` (one of the reference's five per-module state dicts),
    with real ``torch.nn.utils.parametrizations.weight_norm`` parametrizations wherever `sd` holds ``<m>.parametrizations.weight.
    original{0,1}`` (reference layers.py:11-25), loaded with ``strict=True``.  Stands where a reference sub-module stands when the
    reference tree itself is absent (the GPU box): INTEGRATION.md's ``folded_tensors()`` walks it as it would walk the real one."""
    import torch.nn as nn
    from torch.nn.utils.parametrizations import weight_norm

    class Node(nn.Module):
        pass

    root = Node()

    def node_at(path):
        m = root
        for part in path:
            if not hasattr(m, part):
                m.add_module(part, Node())
            m = getattr(m, part)
        return m
    normed = sorted({k[:-len(".parametrizations.weight.original0")] for k in sd if k.endswith(".parametrizations.weight.original0")})
    for prefix in normed:
        leaf = node_at(prefix.split("."))
        leaf.weight = nn.Parameter(torch.zeros_like(sd[f"{prefix}.parametrizations.weight.original1"]) + 1.0)
        weight_norm(leaf)  # dim 0: the norm over all dims but the output channel, as the reference's wrapper
    for k, v in sd.items():
        if ".parametrizations." in k:
            continue
        *path, name = k.split(".")
        node_at(path).register_parameter(name, nn.Parameter(torch.zeros_like(v)))
    root.load_state_dict(sd, strict=True)
    return root.eval()


# ---- oracle outputs, computed once per session ------------------------------------------------------------------------------
ORACLE_CHUNK = 32  # clips per oracle call (host memory: the unfused CPU path holds ~25 MB per clip-second)
_ORACLE_ENCODE_CACHE = {}


def _digest(*tensors):
    import hashlib
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def oracle_indices(w, mc, audio, use_cache=True):
    """Oracle tokens + latents of `audio` [B, T] (CPU), in chunks of ORACLE_CHUNK clips.  Several tests compare the same seeded
    batches (or a prefix of one) with the oracle: each chunk of clips is encoded once per session, keyed by the bytes of the chunk's
    audio and a fingerprint of the weights (every tensor's sum and the bytes of the quantiser's projection).
    ``use_cache=False``: encode now, whatever the session already holds — for a caller that has hooked the oracle (a counting snake)
    and needs the evaluation to actually happen, in any test order; the results still go into the cache."""
    from oracle import l3ac_oracle as O
    wkey = (tuple(mc.levels), mc.hop_length, _digest(w["quantizer.project_in.weight"], w["quantizer.project_out.weight"]),
            round(float(sum(float(v.double().sum()) for v in w.values())), 6))
    idx, lat = [], []
    for b0 in range(0, audio.shape[0], ORACLE_CHUNK):
        chunk = audio[b0:b0 + ORACLE_CHUNK]
        key = (wkey, tuple(chunk.shape), _digest(chunk))
        if not use_cache or key not in _ORACLE_ENCODE_CACHE:
            taps = {}
            _, ind = O.encode_audio(w, mc, chunk, taps=taps)
            _ORACLE_ENCODE_CACHE[key] = (ind["indices"], taps["latents"])
        i, l = _ORACLE_ENCODE_CACHE[key]
        idx.append(i)
        lat.append(l)
    return torch.cat(idx), torch.cat(lat)
