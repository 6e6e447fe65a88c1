"""Model-zoo configuration for the L3AC hot path.

Parses the same TOML schema the reference ships (reference l3ac/configs/*.toml) without
pydantic-settings: one `L3ACConfig` (reference l3ac/__init__.py:54-81) holding a nested
`ModelConfig` (reference l3ac/codec.py:13-36 + l3ac/en_codec.py:9-19).  Unknown keys are
rejected, as the reference's ``extra='forbid'`` does (reference l3ac/xtract/config.py:8).
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Optional

import tomli
from pydantic import BaseModel, ConfigDict, Field, computed_field, model_validator

CONFIG_DIR = Path(__file__).parent / "configs"


class ModelConfig(BaseModel):
    """Network geometry.  Field names/defaults follow reference codec.py:13-25, en_codec.py:10-14."""

    model_config = ConfigDict(extra="forbid")

    feature_dim: int = 256
    compress_rates: tuple[int, ...] = (9, 5)
    encoder_dims: tuple[int, ...] = (24, 96, 192)
    encoder_depths: tuple[int, ...] = (1, 1, 2)
    decode_rates: tuple[int, ...] = (5, 3, 3)
    decoder_dims: tuple[int, ...] = (256, 128, 64, 32)
    decoder_depths: tuple[int, ...] = (3, 2, 1, 1)
    base_unit: str = "normal"
    use_norm: bool = True
    use_snake_act: bool = True
    decoder_last_layer: Optional[str] = None
    vq_config: dict = Field(default_factory=lambda: dict(name="super_fsq", levels=[7] * 6, noise_rate=0.5))
    en_coder_depth: int = 2
    en_coder_window_size: int = 500
    en_coder_dynamic_pos: bool = False
    en_coder_compress_rate: int = 1
    en_coder_cache_size: int = 0

    @model_validator(mode="after")
    def _check(self):
        # arity asserts: reference codec.py:32-36
        assert len(self.compress_rates) + 1 == len(self.encoder_dims) == len(self.encoder_depths)
        assert len(self.decode_rates) + 1 == len(self.decoder_dims) == len(self.decoder_depths)
        return self

    # ---- derived geometry -------------------------------------------------------------
    @computed_field
    @property
    def hop_length(self) -> int:
        """Samples per token: prod(compress_rates) * en_coder_compress_rate (en_codec.py:16-19)."""
        return math.prod(self.compress_rates) * self.en_coder_compress_rate

    @property
    def levels(self) -> tuple[int, ...]:
        return tuple(int(v) for v in self.vq_config["levels"])

    @property
    def codebook_size(self) -> int:
        return math.prod(self.levels)

    @property
    def compressed(self) -> bool:
        """True → CompressedLocal{En,De}coderWithCache, else LocalEncoder/LocalDecoder (en_codec.py:25-44)."""
        return not (self.en_coder_compress_rate == 1 and self.en_coder_cache_size == 0)

    def check_supported(self) -> None:
        """The hot path implements exactly what the shipped configs select; anything else fails loudly."""
        if self.base_unit != "normal":
            raise NotImplementedError(f"base_unit={self.base_unit!r} (reference asserts 'normal', codec.py:42)")
        if not (self.use_norm and self.use_snake_act):
            raise NotImplementedError("only use_norm=true, use_snake_act=true is implemented")
        if self.decoder_last_layer != "legacy":
            raise NotImplementedError(f"decoder_last_layer={self.decoder_last_layer!r}: only 'legacy' is implemented")
        if not self.en_coder_dynamic_pos:
            raise NotImplementedError("rotary position embedding (en_coder_dynamic_pos=false) is not implemented")
        if self.en_coder_cache_size != 0:
            raise NotImplementedError("en_coder_cache_size must be 0 (reference asserts it, local_trans.py:151,174)")
        if self.vq_config.get("name", "vq") != "super_fsq" or self.vq_config.get("codebook_num", 1) != 1:
            raise ValueError(f"Unknown vq config: {self.vq_config}")  # reference vq/__init__.py:37-47
        if self.compressed and self.en_coder_depth < 2:
            raise ValueError("compressed en_decoder needs en_coder_depth >= 2 (local_trans.py:176-180)")
        if self.codebook_size >= 2 ** 24:
            raise ValueError("codebook too large for the exact fp32 index sum (vq/fsq.py:67-68)")


class L3ACConfig(BaseModel):
    """Top-level model config; mirrors reference l3ac/__init__.py:54-81 field for field."""

    model_config = ConfigDict(extra="forbid", protected_namespaces=())

    config_file: Optional[Path] = None
    model_name: str = "debug"
    sample_rate: int = 16000
    model_version: str = "v0.0"
    model_dir: Path = Path.home() / ".cache" / "l3ac"
    weight_url: Optional[str] = None
    network_config: Optional[ModelConfig] = None

    def __init__(self, config_file=None, **overrides):
        data = {}
        if config_file is not None:
            with open(config_file, "rb") as fh:
                data = tomli.load(fh)
        data.update(overrides)  # init arguments take priority over the file (xtract/config.py:16-31)
        super().__init__(config_file=config_file, **data)
        if self.weight_url is None:  # reference __init__.py:74-81
            self.weight_url = (
                "https://huggingface.co/zhai-lw/L3AC/resolve/main/weights/"
                f"{self.model_name}.{self.model_version}/" + "{}.pt"
            )

    @property
    def model_tag(self) -> str:
        return f"{self.model_name}.{self.model_version}"

    @property
    def model_path(self) -> Path:
        return self.model_dir / self.model_tag


def list_models() -> list[str]:
    """Stems of the shipped config files (reference l3ac/__init__.py:17-18)."""
    return sorted(p.stem for p in CONFIG_DIR.glob("*.toml"))


def resolve_config_file(config_name) -> Path:
    p = Path(str(config_name))
    if p.suffix == ".toml" and p.exists():
        return p
    cand = CONFIG_DIR / f"{config_name}.toml"
    if not cand.exists():
        raise FileNotFoundError(f"no such model config: {config_name!r} (known: {list_models()})")
    return cand
