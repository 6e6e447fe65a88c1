"""ctypes binding of libl3ac_hip.so (include/l3ac_hip.h).  No torch types cross this boundary: only raw
device pointers, sizes and the stream handle.  The library is REQUIRED: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# L3AC_LIB_PATH: load another build of the same library (experiment builds); default in-tree
LIB_PATH = Path(os.environ.get("L3AC_LIB_PATH") or Path(__file__).resolve().parent / "libl3ac_hip.so")
ABI_VERSION = 5
MAX_STAGES = 8
MAX_LEVELS = 8


class L3acError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("feature_dim", C.c_int32),
        ("n_enc", C.c_int32),
        ("enc_dims", C.c_int32 * MAX_STAGES),
        ("enc_depths", C.c_int32 * MAX_STAGES),
        ("compress_rates", C.c_int32 * MAX_STAGES),
        ("n_dec", C.c_int32),
        ("dec_dims", C.c_int32 * MAX_STAGES),
        ("dec_depths", C.c_int32 * MAX_STAGES),
        ("decode_rates", C.c_int32 * MAX_STAGES),
        ("n_levels", C.c_int32),
        ("levels", C.c_int32 * MAX_LEVELS),
        ("en_coder_depth", C.c_int32),
        ("en_coder_window_size", C.c_int32),
        ("en_coder_compress_rate", C.c_int32),
        ("grn_exact", C.c_int32),
    ]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


# name -> (restype, argtypes); must list every symbol include/l3ac_hip.h declares (tests check this)
_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "l3ac_last_error": (C.c_char_p, []),
    "l3ac_abi_version": (C.c_int, []),
    "l3ac_create": (C.c_int, [C.POINTER(Config), C.POINTER(Tensor), _I32, _I32, C.POINTER(_P)]),
    "l3ac_destroy": (None, [_P]),
    "l3ac_reserve": (C.c_int, [_P, _I32, _I32]),
    "l3ac_workspace_bytes": (_I64, [_P]),
    "l3ac_grn_min_norm": (C.c_int, [_P, _I32, C.POINTER(C.c_float)]),
    "l3ac_bad_index_count": (C.c_int, [_P, _I32, C.POINTER(_I64)]),
    "l3ac_coop_timeout_count": (C.c_int, [_P, _I32, C.POINTER(_I64)]),
    "l3ac_coop_timeout_pending": (C.c_int, [_P, C.POINTER(_I64)]),
    "l3ac_coop_claimed_cus": (_I32, [_I32]),
    "l3ac_hop_length": (_I32, [_P]),
    "l3ac_encode": (C.c_int, [_P, _P, _I32, _I32, _I64, _P, _P, _P, _P]),
    "l3ac_decode": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "l3ac_fsq_forward": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "l3ac_fsq_quantize_act": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), _I32, _P, _P, _P, _P, _P, _P]),
    "l3ac_fsq_decode": (C.c_int, [_P, _I64, _I32, C.POINTER(_I32), _I32, _P, _P, _P, _P]),
    "l3ac_fsq_copy_ceiling": (C.c_int, [_P, _I64, _P, _P, _P, _P]),
    "l3ac_fsq_copy_ceiling_at": (C.c_int, [_P, _I64, _P, _P, _P, _I32, _P]),
    "l3ac_vq_argmin_scratch_bytes": (_I64, [_I64, _I32, _I32]),
    "l3ac_vq_argmin": (C.c_int, [_P, _I64, _P, _I32, _I32, _P, _P, _I64, _I32, _P]),
    "l3ac_op_first_block": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_conv_unit": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_down_layer": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_conv_k3": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_enhance": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_up_layer": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_enhance_up": (C.c_int, [_P, C.c_char_p, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_last_block": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_local_trans": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, _P, _P]),
    "l3ac_op_encoder": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_en_encoder": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_en_decoder": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_decoder": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "l3ac_op_snake": (C.c_int, [_P, _P, _I64, _I32, _P, _I32, _P]),
    "l3ac_ctx_set_head_pretanh": (C.c_int, [_P, _I32]),
    "l3ac_gemm_f32": (C.c_int, [_P, _I64, _P, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "l3ac_split3_host": (None, [_P, _I64, _P]),
    "l3ac_ctx_set_gemm_split": (C.c_int, [_P, _I32]),
    "l3ac_ctx_set_option": (C.c_int, [_P, C.c_char_p, _I32]),
    "l3ac_ctx_get_gemm_split": (_I32, [_P]),
    "l3ac_gemm_split_image_bytes": (_I64, [_I32, _I32]),
    "l3ac_gemm_split_image": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "l3ac_gemm_split_f32": (C.c_int, [_P, _I64, _P, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "l3ac_pack_indices": (C.c_int, [_P, _I32, _I32, _I32, _P, _I32, _P]),
    "l3ac_unpack_indices": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "l3ac_profile_begin": (C.c_int, []),
    "l3ac_profile_end": (C.c_int, [_P, _I32, C.POINTER(_I32)]),
}


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int32), ("reserved", C.c_int32),
                ("ms_total", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class profile:
    """Context manager: per-kernel device time / algorithmic work of everything launched inside it."""

    def __enter__(self):
        check(load_library().l3ac_profile_begin())
        self.entries = []
        return self

    def __exit__(self, *exc):
        buf = (ProfileEntry * 64)()
        n = C.c_int32(0)
        check(load_library().l3ac_profile_end(buf, 64, C.byref(n)))
        self.entries = [dict(name=buf[i].name.decode(), launches=buf[i].launches, ms_total=buf[i].ms_total,
                             flops=buf[i].flops, bytes=buf[i].bytes) for i in range(n.value)]
        return False

_lib = None


def load_library() -> C.CDLL:
    """dlopen the extension; raises loudly if it has not been built (python -m l3ac_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise L3acError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
            "(run `python -m l3ac_amd.build`); this package has no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.l3ac_abi_version() != ABI_VERSION:
        raise L3acError(f"libl3ac_hip.so ABI {lib.l3ac_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().l3ac_last_error()
        raise L3acError(f"libl3ac_hip error {rc}: {msg.decode() if msg else '?'}")


def make_config(mc, grn_exact: bool = False) -> Config:
    cfg = Config()
    cfg.abi_version = ABI_VERSION
    cfg.feature_dim = mc.feature_dim
    cfg.n_enc = len(mc.encoder_dims)
    cfg.n_dec = len(mc.decoder_dims)
    if cfg.n_enc > MAX_STAGES or cfg.n_dec > MAX_STAGES or len(mc.levels) > MAX_LEVELS:
        raise L3acError("network too deep for the C ABI (L3AC_MAX_STAGES / L3AC_MAX_LEVELS)")
    for i, v in enumerate(mc.encoder_dims):
        cfg.enc_dims[i] = v
    for i, v in enumerate(mc.encoder_depths):
        cfg.enc_depths[i] = v
    for i, v in enumerate(mc.compress_rates):
        cfg.compress_rates[i] = v
    for i, v in enumerate(mc.decoder_dims):
        cfg.dec_dims[i] = v
    for i, v in enumerate(mc.decoder_depths):
        cfg.dec_depths[i] = v
    for i, v in enumerate(mc.decode_rates):
        cfg.decode_rates[i] = v
    cfg.n_levels = len(mc.levels)
    for i, v in enumerate(mc.levels):
        cfg.levels[i] = v
    cfg.en_coder_depth = mc.en_coder_depth
    cfg.en_coder_window_size = mc.en_coder_window_size
    cfg.en_coder_compress_rate = mc.en_coder_compress_rate
    cfg.grn_exact = int(bool(grn_exact))
    return cfg


class Context:
    """Owns one l3ac_ctx (device-resident folded weights + workspace) on one GPU."""

    def __init__(self, mc, folded: dict, device_index: int, grn_exact: bool = False):
        self.lib = load_library()
        self.mc = mc
        self.device_index = device_index
        cfg = make_config(mc, grn_exact)
        names = sorted(folded)
        arr = (Tensor * len(names))()
        self._keep = []
        for i, name in enumerate(names):
            t = folded[name].detach().to("cpu").float().contiguous()
            self._keep.append(t)
            arr[i].name = name.encode()
            arr[i].data = t.data_ptr()
            arr[i].numel = t.numel()
        handle = C.c_void_p()
        check(self.lib.l3ac_create(C.byref(cfg), arr, len(names), device_index, C.byref(handle)))
        self._keep = None  # weights now live on the device
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.l3ac_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def hop_length(self) -> int:
        return self.lib.l3ac_hop_length(self.handle)

    def bad_index_count(self, reset: bool = False) -> int:
        """Indices outside [0, codebook size) that decode calls of this context have met (and clamped).  Synchronises."""
        out = _I64(0)
        check(self.lib.l3ac_bad_index_count(self.handle, int(reset), C.byref(out)))
        return int(out.value)

    def coop_timeout_count(self, reset: bool = False) -> int:
        """Arrival polls of the cooperative transformer kernel that expired since the last reset (calls whose outputs are invalid;
        the context has then fallen back to the one-workgroup form).  Synchronises."""
        out = _I64(0)
        check(self.lib.l3ac_coop_timeout_count(self.handle, int(reset), C.byref(out)))
        return int(out.value)

    def coop_timeout_pending(self) -> int:
        """Expired arrival polls that have NOT been reported yet (no L3AC_ECOOP returned for them, no coop_timeout_count call since).
        Synchronises; changes nothing: the next call on the context still reports them."""
        out = _I64(0)
        check(self.lib.l3ac_coop_timeout_pending(self.handle, C.byref(out)))
        return int(out.value)

    def grn_min_norm(self, reset: bool = False) -> float:
        """Smallest per-clip GRN norm this context has seen (grn_exact contexts; +inf otherwise).  Synchronises."""
        out = C.c_float(0.0)
        check(self.lib.l3ac_grn_min_norm(self.handle, int(reset), C.byref(out)))
        return float(out.value)

    @property
    def workspace_bytes(self) -> int:
        return self.lib.l3ac_workspace_bytes(self.handle)

    def reserve(self, batch: int, samples: int) -> None:
        check(self.lib.l3ac_reserve(self.handle, batch, samples))

    # ---- route switches: state of THIS context only (include/l3ac_hip.h) -----------------------------------
    def set_gemm_split(self, enable: bool) -> None:
        check(self.lib.l3ac_ctx_set_gemm_split(self.handle, int(bool(enable))))

    def get_gemm_split(self) -> bool:
        return bool(self.lib.l3ac_ctx_get_gemm_split(self.handle))

    def set_option(self, name: str, value: int) -> None:
        check(self.lib.l3ac_ctx_set_option(self.handle, name.encode(), int(value)))

    def set_head_pretanh(self, enable: bool) -> None:
        check(self.lib.l3ac_ctx_set_head_pretanh(self.handle, int(bool(enable))))
