"""l3ac_amd — MI355X-native encode -> quantize -> decode path behind L3AC's Python surface.

Drop-in for the reference package on this path (``import l3ac_amd as l3ac``):

    l3ac.list_models()                      reference l3ac/__init__.py:17-18
    codec = l3ac.get_model("1kbps")         reference l3ac/__init__.py:21-25
    codec.config.sample_rate                reference l3ac/__init__.py:54-81
    codec.network.to(device="cuda"); codec.network.eval()
    q_feature, indices = codec.encode_audio(audio)          reference l3ac/__init__.py:108-114
    audio = codec.decode_audio(q_feature)                   reference l3ac/__init__.py:116-121
    audio = codec.decode_audio(indices=indices["indices"])

All arithmetic runs in hand-written HIP kernels inside libl3ac_hip.so (include/l3ac_hip.h) on gfx950; torch
supplies device memory and streams only.  There is no CPU path: calling encode/decode with CPU tensors, or
without the built extension, raises.
"""
from __future__ import annotations

import logging
import math
import os
import weakref
from pathlib import Path
from typing import Optional

import torch

from . import _capi
from . import weights as _weights
from .chunking import ChunkData, plan as _chunk_plan
from .config import CONFIG_DIR, L3ACConfig, ModelConfig, list_models, resolve_config_file

__all__ = ["set_gemm_split", "get_gemm_split", "gemm_split_routes", "restore_gemm_split_routes", "list_models", "get_model", "get_model_info", "L3AC", "L3ACConfig", "ModelConfig", "Network",
           "bits_per_token", "pack_indices", "unpack_indices", "ChunkData"]
__version__ = "0.1.0"

log = logging.getLogger("L3AC")

# GEMM route: state of each Network's own HIP context.  The module-level setter below keeps the
# round-1/2 convenience (one call switches every live network and the default of later ones) without any process-wide
# state inside the library.
def _env_atoi_flag(name: str, default: bool) -> bool:
    """The library's own reading of a 0/1 environment switch (`std::atoi(value) != 0`, kernels/gemm_split.hip): leading
    whitespace, an optional sign, then digits; anything else ("off", "") counts as 0."""
    import re
    v = os.environ.get(name)
    if v is None:
        return default
    m = re.match(r"\s*([+-]?\d+)", v)
    return bool(m) and int(m.group(1)) != 0


_default_gemm_split = _env_atoi_flag("L3AC_GEMM_SPLIT", True)
_networks: "weakref.WeakSet[Network]" = weakref.WeakSet()


class Network:
    """Stands where the reference's ``EnCodec`` nn.Module stands (``codec.network``): holds the weights and the
    per-device HIP context.  Callers only move it (``.to`` / ``.cuda``) and switch it to eval mode."""

    def __init__(self, mc: ModelConfig):
        mc.check_supported()
        self.mc = mc
        self.training = True  # nn.Module default; the reference needs .eval() before inference (vq/fsq.py:31)
        # GRN (layers.py:112-115) normalises by g / (g + 1e-8), g = the clip's L2 norm over the whole hidden tensor: exactly 1.0f in
        # fp32 for g >= 0.25, which the kernels assume.  grn_exact = True (set BEFORE .to(device)) is the validation mode: the
        # literal two-pass formula is evaluated (correct for any input) and min_grn_norm() reports the smallest g seen, i.e.
        # whether the fast path would have been exact for the data that went through.
        self.grn_exact = False
        self._gemm_split = _default_gemm_split
        _networks.add(self)
        self._state_dicts = None
        self._folded = None
        self._ctx: Optional[_capi.Context] = None
        self.device = torch.device("cpu")

    # ---- weights ------------------------------------------------------------------------------------
    def load_state_dicts(self, state_dicts) -> "Network":
        _weights.check_state_dicts(state_dicts, self.mc)
        self._state_dicts = state_dicts
        self._folded = _weights.folded_weights(state_dicts)  # weight-norm folded once
        if self._ctx is not None:
            self._drop_ctx()
            self._make_ctx()
        return self

    def load_model(self, model_dir=None, model_path=None) -> "Network":
        """reference xtract/nn/module.py:43-54, but a missing file raises instead of keeping random weights."""
        model_path = Path(model_path) if model_path is not None else Path(model_dir)
        return self.load_state_dicts(_weights.load_state_dicts(model_path, self.mc))

    def state_dicts(self):
        return self._state_dicts

    @property
    def trainable_modules(self):
        return {name: self._state_dicts[name] for name in _weights.MODULE_NAMES} if self._state_dicts else {}

    # ---- nn.Module-like surface -----------------------------------------------------------------------
    def eval(self) -> "Network":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "Network":
        if mode:
            raise NotImplementedError("l3ac_amd implements the inference path only (FSQ noise / drop-path are training-side)")
        return self.eval()

    def to(self, device=None, dtype=None, **_ignored) -> "Network":
        if dtype is not None and dtype != torch.float32:
            raise NotImplementedError("the path computes in fp32, like the reference")
        if device is None:
            return self
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self.device:
            self._drop_ctx()
            self.device = device
            if device.type == "cuda":
                self._make_ctx()
        return self

    def cuda(self, device=None) -> "Network":
        return self.to(device="cuda" if device is None else device)

    def cpu(self) -> "Network":
        return self.to(device="cpu")

    def _make_ctx(self):
        if self._folded is None:
            raise RuntimeError("no weights loaded (get_model / load_state_dicts first)")
        self._ctx = _capi.Context(self.mc, self._folded, self.device.index, grn_exact=self.grn_exact)
        self._ctx.set_gemm_split(self._gemm_split)

    @property
    def gemm_split(self) -> bool:
        """The route of this network: its context's own state once it has one (whoever set it — `set_gemm_split`,
        `ctx.set_option("gemm_split", ...)`, `ctx.set_gemm_split`), else the route its context will be created on."""
        return self._ctx.get_gemm_split() if self._ctx is not None else self._gemm_split

    def set_gemm_split(self, enable: bool) -> "Network":
        """Route of THIS network's context: True = the large fp32 contractions as exact bf16x3 operand splits on the bf16 matrix
        cores (default), False = every product on the exact fp32 MFMA instruction.  Not to be called while another thread
        runs or captures a graph on this network."""
        self._gemm_split = bool(enable)
        if self._ctx is not None:
            self._ctx.set_gemm_split(self._gemm_split)
        return self

    def _drop_ctx(self):
        if self._ctx is not None:
            self._gemm_split = self._ctx.get_gemm_split()  # the route travels with the network to its next device
            self._ctx.close()
            self._ctx = None

    def min_grn_norm(self, reset: bool = False) -> float:
        """Validation mode only (grn_exact = True): the smallest per-clip GRN norm seen so far; the default fast path is exact
        for every input whose value here is >= 0.25."""
        if not self.grn_exact:
            raise RuntimeError("min_grn_norm() needs the validation mode: set network.grn_exact = True before .to(device)")
        return self.context().grn_min_norm(reset)

    def context(self) -> _capi.Context:
        if self._ctx is None:
            raise RuntimeError(
                "network is not on a GPU: call codec.network.to(device='cuda') first "
                "(l3ac_amd has no CPU path; the reference's PyTorch-CPU path is not part of this package)")
        return self._ctx

    # ---- reference Codec.preprocess (codec.py:79-84): kept for callers that use it -----------------
    def preprocess(self, audio_data: torch.Tensor):
        length = audio_data.shape[-1]
        hop = self.mc.hop_length
        pad_len = math.ceil(length / hop) * hop - length
        return torch.nn.functional.pad(audio_data, (0, pad_len)), length


class L3AC:
    """reference l3ac/__init__.py:84-121."""

    def __init__(self, config: L3ACConfig):
        self.config = config
        self.network = Network(config.network_config)

    def load_pretrained(self):
        """reference :104-106 minus the HTTP download (no network here): weights must already be on disk."""
        if not self.config.model_path.exists():
            raise FileNotFoundError(
                f"no weights at {self.config.model_path}: download "
                f"{self.config.weight_url.format('{encoder,quantizer,decoder,en_encoder,en_decoder}')} there, "
                "pass model_dir=..., or use get_model(..., synthetic_seed=N)")
        self.network.load_model(model_path=self.config.model_path)

    # ---- hot path -------------------------------------------------------------------------------------
    def _check_input(self, t: torch.Tensor, what: str):
        if self.network.training:
            raise RuntimeError("call codec.network.eval() first: the training-mode quantiser injects noise "
                               "(reference vq/fsq.py:31,40-43), which this inference path does not implement")
        ctx = self.network.context()
        if not t.is_cuda or t.device != self.network.device:
            raise RuntimeError(f"{what} is on {t.device} but the network is on {self.network.device}")
        return ctx

    @staticmethod
    def _coop_check_before(ctx, what: str):
        """validate=True, before the call: an EARLIER call's expired polls that nobody has been told about must not disappear into this
        call's baseline (they would: l3ac_coop_timeout_count acknowledges what it reports).  Synchronises."""
        earlier = ctx.coop_timeout_pending()
        if earlier:
            ctx.coop_timeout_count()  # delivered by the exception below: fall back, re-zero the arrival counters
            raise _capi.L3acError(
                f"{what}(validate=True): an EARLIER call on this context lost {earlier} arrival poll(s) of the cooperative transformer "
                "kernel to its time limit; that call's outputs are invalid (every call since the last validated one is suspect). "
                "Nothing was run. The context now runs the one-workgroup form (same bits): repeat those calls")

    @staticmethod
    def _raise_on_coop_timeout(ctx, what: str):
        lost = ctx.coop_timeout_pending()  # (synchronises)
        if lost:
            ctx.coop_timeout_count()  # delivered here: the context falls back to the one-workgroup form, counters re-zeroed
            raise _capi.L3acError(
                f"{what}: the cooperative transformer kernel lost {lost} arrival poll(s) to its time limit (its six workgroups per "
                "clip were not co-resident: another process or a CU mask on the device?); this call's outputs are invalid. The "
                "context now runs the one-workgroup form (same bits): repeat the call")

    @torch.no_grad()
    def encode_audio(self, audio_data: torch.Tensor, validate: bool = False):
        """audio (B, T) fp32 -> (q_feature (B, T_tok, C) fp32, {"indices": int32 (B, T_tok),
        "level_indices": fp32 (B, T_tok, D)}); the zero right-padding to a hop multiple happens in-kernel.
        ``validate=True`` synchronises before and after the call: it raises — without running anything — if an EARLIER, unvalidated
        call on the context lost a cooperative transformer launch to its time limit, and raises if a launch of THIS call did.  Without
        it a later call on the context returns L3AC_ECOOP once, after the fact and possibly several calls late (the entry check does
        not synchronise: include/l3ac_hip.h, L3AC_ECOOP / l3ac_coop_timeout_pending)."""
        ctx = self._check_input(audio_data, "audio_data")
        if audio_data.dim() != 2:
            raise ValueError(f"audio_data must be (batch, samples), got {tuple(audio_data.shape)}")
        audio = audio_data.to(torch.float32)
        if audio.stride(-1) != 1 or audio.stride(0) % 4 != 0 or audio.data_ptr() % 16 != 0:
            audio = audio.contiguous()
        b, t = audio.shape
        if t == 0 or b == 0:
            raise ValueError("empty audio")
        mc = self.network.mc
        n_tok = math.ceil(t / mc.hop_length)
        dev = audio.device
        q_feature = torch.empty((b, n_tok, mc.feature_dim), dtype=torch.float32, device=dev)
        indices = torch.empty((b, n_tok), dtype=torch.int32, device=dev)
        level_indices = torch.empty((b, n_tok, len(mc.levels)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if validate:
                self._coop_check_before(ctx, "encode_audio")
            _capi.check(ctx.lib.l3ac_encode(ctx.handle, audio.data_ptr(), b, t, audio.stride(0) if b > 1 else t,
                                            q_feature.data_ptr(), indices.data_ptr(), level_indices.data_ptr(), stream))
            if validate:
                self._raise_on_coop_timeout(ctx, "encode_audio")
        return q_feature, {"indices": indices, "level_indices": level_indices}

    @torch.no_grad()
    def decode_audio(self, audio_feature: torch.Tensor = None, indices: torch.Tensor = None, validate: bool = False) -> torch.Tensor:
        """(B, T_tok, C) features, or int indices (B, T_tok) -> audio (B, T_tok * hop), not trimmed.
        Indices outside [0, codebook_size) — a corrupted or truncated token stream — are clamped into range and counted on
        the device (``codec.network.context().bad_index_count()``); with ``validate=True`` the call synchronises and raises
        if this call met any — or if a cooperative transformer launch of this call timed out (as encode_audio)."""
        src = audio_feature if audio_feature is not None else indices
        if src is None:
            raise ValueError("decode_audio needs audio_feature or indices")
        ctx = self._check_input(src, "decode input")
        mc = self.network.mc
        if audio_feature is not None:
            if audio_feature.dim() != 3 or audio_feature.shape[-1] != mc.feature_dim:
                raise ValueError(f"audio_feature must be (batch, tokens, {mc.feature_dim})")
            feat = audio_feature.to(torch.float32).contiguous()
            b, n_tok = feat.shape[:2]
            f_ptr, i_ptr, keep = feat.data_ptr(), None, feat
        else:
            if indices.dim() != 2:
                raise ValueError("indices must be (batch, tokens)")
            idx = indices.to(torch.int32).contiguous()
            b, n_tok = idx.shape
            f_ptr, i_ptr, keep = None, idx.data_ptr(), idx
        if n_tok * mc.en_coder_compress_rate < 2:
            # reference behaviour: the first EnhanceBlock's InstanceNorm1d (tconv/__init__.py:36) raises on a single frame
            raise ValueError(f"Expected more than 1 spatial element when training, got input size torch.Size([{b}, 4, 1])")
        audio = torch.empty((b, n_tok * mc.hop_length), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            stream = torch.cuda.current_stream(src.device).cuda_stream
            before = ctx.bad_index_count() if validate and i_ptr is not None else 0  # cumulative counter: read, never reset here
            if validate:
                self._coop_check_before(ctx, "decode_audio")
            _capi.check(ctx.lib.l3ac_decode(ctx.handle, f_ptr, i_ptr, b, n_tok, audio.data_ptr(), stream))
            if validate:
                self._raise_on_coop_timeout(ctx, "decode_audio")
            if validate and i_ptr is not None:
                bad = ctx.bad_index_count() - before
                if bad:
                    raise ValueError(f"{bad} of {b * n_tok} indices lie outside [0, {mc.codebook_size}): corrupted token stream")
        del keep
        return audio


    # ---- long audio (reference l3ac/codec.py:124-156, corrected: see l3ac_amd/chunking.py) ---------------------------
    def _batched(self, fn, chunks):
        """Run `fn` on chunks grouped by length: equal-length chunks (all the middle ones) go through ONE call as a batch."""
        out = [None] * len(chunks)
        by_len = {}
        for i, c in enumerate(chunks):
            by_len.setdefault(c.shape[-1] if c.dim() == 1 else c.shape[0], []).append(i)
        for idxs in by_len.values():
            res = fn(torch.stack([chunks[i] for i in idxs]))
            for k, i in enumerate(idxs):
                out[i] = res[k]
        return out

    @torch.no_grad()
    def extract_unit(self, audio_data: torch.Tensor, process_window: int = 5 * 16000, prefix_tokens: Optional[int] = None):
        """Encode a clip of any length window by window: (1, T) audio -> (ChunkData of indices, ChunkData of q_feature), the
        return structure and the default ``process_window`` of reference ``Codec.extract_unit`` (codec.py:124-147).  Unlike
        the reference, every chunk goes through the whole encode path (``en_encoder`` included); equal-length chunks are
        batched through one ``encode_audio`` call.  ``.data`` of either result is the merged token stream.

        ``prefix_tokens`` is the overlap of a chunk with its predecessor, in tokens.  **The default differs from the
        reference's**: the local attention's window (its look-back, ``en_coder_window_size`` tokens) instead of ONE hop, so
        the returned ``ChunkData.prefix_len`` is that window, not 1.  ``prefix_tokens=1`` gives exactly the reference's chunk
        geometry (``chunk_len = process_window // hop`` tokens, ``prefix_len = 1``): such ChunkData can be merged / decoded by
        reference code with the same arguments, and the reference's by ``decode_unit`` here."""
        assert audio_data.dim() == 2 and len(audio_data) == 1, "Only support batch size 1"  # codec.py:133
        mc = self.network.mc
        hop = mc.hop_length
        prefix_tokens = mc.en_coder_window_size if prefix_tokens is None else int(prefix_tokens)
        audio, _ = self.network.preprocess(audio_data)  # right zero-pad to a hop multiple (codec.py:79-84)
        chunk_len, prefix_len = _chunk_plan(hop, process_window, prefix_tokens)
        chunks = ChunkData(chunk_len=chunk_len, prefix_len=prefix_len, original_data=audio[0]).chunk_data
        idx, feat = [None] * len(chunks), [None] * len(chunks)

        def run(batch):
            q, ind = self.encode_audio(batch)
            return list(zip(ind["indices"], q))
        for i, (ix, q) in enumerate(self._batched(run, chunks)):
            idx[i], feat[i] = ix, q
        return (ChunkData(chunk_len=chunk_len // hop, prefix_len=prefix_tokens, chunk_data=idx),
                ChunkData(chunk_len=chunk_len // hop, prefix_len=prefix_tokens, chunk_data=feat))

    @torch.no_grad()
    def decode_unit(self, chunk_indices: Optional[ChunkData] = None, chunk_q_feature: Optional[ChunkData] = None,
                    audio_length: Optional[int] = None) -> torch.Tensor:
        """Decode what ``extract_unit`` returned, chunk by chunk (each with its overlap tokens as left context, equal-length
        chunks batched), and merge the waveforms: reference ``Codec.decode_unit`` (codec.py:149-156) -> (1, T) audio,
        trimmed to ``audio_length`` when given."""
        src = chunk_q_feature if chunk_q_feature is not None else chunk_indices
        if src is None:
            raise ValueError("decode_unit needs chunk_indices or chunk_q_feature")
        hop = self.network.mc.hop_length
        if chunk_q_feature is not None:
            waves = self._batched(lambda b: list(self.decode_audio(b)), src.chunk_data)
        else:
            waves = self._batched(lambda b: list(self.decode_audio(indices=b)), src.chunk_data)
        merged = ChunkData(chunk_len=src.chunk_len * hop, prefix_len=src.prefix_len * hop, chunk_data=waves).data[None, :]
        return merged if audio_length is None else merged[:, :audio_length]


def set_gemm_split(enable: bool) -> None:
    """Route the large fp32 channel contractions of EVERY live network (and of networks created later) through the bf16x3
    split-operand kernels (default, fp32 accuracy on the bf16 matrix cores) or through the exact v_mfma_f32_32x32x2_f32
    kernel.  The state itself lives in each network's context (``Network.set_gemm_split``, l3ac_ctx_set_gemm_split): graphs
    already captured keep the route they were captured on."""
    global _default_gemm_split
    _default_gemm_split = bool(enable)
    for net in list(_networks):
        net.set_gemm_split(enable)


def get_gemm_split() -> bool:
    """The DEFAULT route: what networks created from now on start with (``L3AC_GEMM_SPLIT`` read the way the library reads it, or
    the last module-level ``set_gemm_split``).  It is not the state of any live network — a network whose route was changed on
    its own (``Network.set_gemm_split``, a context option) reports it in ``network.gemm_split``.  Code that switches routes
    temporarily should save and restore per network, or use ``gemm_split_routes()``."""
    return _default_gemm_split


def gemm_split_routes() -> dict:
    """{network: route} of every live network — the snapshot ``restore_gemm_split_routes`` takes back."""
    return {net: net.gemm_split for net in list(_networks)}


def restore_gemm_split_routes(routes: dict, default: Optional[bool] = None) -> None:
    """Undo a module-level ``set_gemm_split``: every network in `routes` gets ITS previous route back (a module-level
    ``set_gemm_split(before)`` would overwrite individually routed networks with the default)."""
    global _default_gemm_split
    if default is not None:
        _default_gemm_split = bool(default)
    for net, route in routes.items():
        net.set_gemm_split(route)


def bits_per_token(mc) -> int:
    """ceil(log2(codebook size)): 17 at 1kbps (117 649 codes), 18 at 3kbps (250 047)."""
    return max(1, (mc.codebook_size - 1).bit_length())


def pack_indices(indices: torch.Tensor, bits: int) -> torch.Tensor:
    """int indices (B, T_tok) on the GPU -> little-endian bit stream, one row of whole 32-bit words per clip, as
    uint8 (B, 4 * ceil(T_tok * bits / 32)).  The reference has no wire format (it keeps int32 tensors)."""
    if not indices.is_cuda or indices.dim() != 2:
        raise ValueError("indices must be a (batch, tokens) CUDA tensor")
    idx = indices.to(torch.int32).contiguous()
    b, n_tok = idx.shape
    words = -(-n_tok * bits // 32)
    out = torch.empty((b, words), dtype=torch.int32, device=idx.device)
    lib = _capi.load_library()
    with torch.cuda.device(idx.device):
        _capi.check(lib.l3ac_pack_indices(idx.data_ptr(), b, n_tok, bits, out.data_ptr(), words,
                                          torch.cuda.current_stream(idx.device).cuda_stream))
    return out.view(torch.uint8)


def unpack_indices(packed: torch.Tensor, n_tok: int, bits: int) -> torch.Tensor:
    """Inverse of `pack_indices`: uint8 (B, 4 * words) -> int32 (B, n_tok)."""
    if not packed.is_cuda or packed.dim() != 2 or packed.dtype != torch.uint8 or packed.shape[1] % 4:
        raise ValueError("packed must be a (batch, 4 * words) uint8 CUDA tensor")
    words = packed.shape[1] // 4
    if words * 32 < n_tok * bits:
        raise ValueError("packed stream too short for n_tok tokens")
    src = packed.contiguous().view(torch.int32)
    out = torch.empty((packed.shape[0], n_tok), dtype=torch.int32, device=packed.device)
    lib = _capi.load_library()
    with torch.cuda.device(packed.device):
        _capi.check(lib.l3ac_unpack_indices(src.data_ptr(), packed.shape[0], n_tok, bits, words, out.data_ptr(),
                                            torch.cuda.current_stream(packed.device).cuda_stream))
    return out


def get_model(config_name, model_dir=None, synthetic_seed: Optional[int] = None, synthetic_profile: str = "mild") -> L3AC:
    """reference l3ac/__init__.py:21-25.  ``config_name`` is a shipped model name (``list_models()``) or a path
    to a TOML file of the same schema.  Weights come from ``{model_dir}/{name}.{version}/*.pt`` (default
    ``~/.cache/l3ac``, the reference's cache), or — with ``synthetic_seed`` — from the seeded generator
    (``synthetic_profile``: "mild", or "stress" = the statistics of a trained network, see ``weights.synthetic_state_dicts``)."""
    overrides = {} if model_dir is None else {"model_dir": Path(model_dir)}
    codec = L3AC(L3ACConfig(config_file=resolve_config_file(config_name), **overrides))
    if synthetic_seed is not None:
        codec.network.load_state_dicts(_weights.synthetic_state_dicts(codec.config.network_config, seed=synthetic_seed,
                                                                      profile=synthetic_profile))
    else:
        codec.load_pretrained()
    return codec


def get_model_info(model, eval_flops_seconds=10, sample_rate: int = 16000) -> dict:
    """reference l3ac/__init__.py:28-51.  ``model`` is ``codec.network`` (as in example.py:11) or the codec.  The reference
    traces the model with ptflops on ``eval_flops_seconds`` of audio; here ``macs`` is the analytic multiply-accumulate
    count of the same input (l3ac_amd/macs.py: every conv / linear product plus the local-attention products, which
    ptflops does not see), ``macs_breakdown`` its parts, and ``params`` the exact parameter count of the five modules."""
    from .macs import path_macs
    mc = model.mc if hasattr(model, "mc") else model.network.mc
    compress_rate = mc.hop_length
    codebook_size = mc.codebook_size
    frame_rate = sample_rate / compress_rate
    params = sum(int(math.prod(shape)) for m in _weights.MODULE_NAMES for _, shape in _weights.raw_keys(mc, m))
    macs = path_macs(mc, int(eval_flops_seconds * sample_rate))
    return {
        "macs": macs["total"],
        "macs_breakdown": macs,
        "params": params,
        "codebook_size": codebook_size,
        "frame_rate": frame_rate,
        "bps": frame_rate * math.log2(codebook_size),
        "receptive_field": mc.en_coder_window_size / frame_rate,
    }
