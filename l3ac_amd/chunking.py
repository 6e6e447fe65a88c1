"""Long-audio chunking: the split / merge bookkeeping of the reference's ``ChunkData`` (l3ac/codec.py:159-188) and the
chunk plan of the corrected long-audio path (``L3AC.extract_unit`` / ``L3AC.decode_unit`` in this package).

The reference's ``extract_unit`` / ``decode_unit`` (codec.py:124-156) cut a clip into windows that overlap their predecessor
by ONE hop, run ``Codec.compress`` on each — which skips ``en_encoder`` / ``en_decoder`` entirely — and glue the pieces with
``ChunkData``.  Here the same bookkeeping is kept (same class, same ``data`` / ``chunk_data`` semantics, along ``dim`` —
default 0, the reference's: a waveform ``(T,)``, indices ``(T_tok,)`` or token features ``(T_tok, C)`` are all cut along their
first dimension), but every chunk goes through
the full path (encoder -> en_encoder -> quantizer, en_decoder -> decoder) and the overlap is a parameter whose default is the
local attention's look-back (one window of tokens), since the transformer — not the one-hop conv halo — is what carries
context across a cut.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


class ChunkData:
    """reference l3ac/codec.py:159-188, generalised from dim 0 to a chosen dim.  Either ``original_data`` (to be cut:
    chunk i > 0 is ``data[i*chunk_len - prefix_len : (i+1)*chunk_len]``, chunk 0 has no prefix) or ``chunk_data`` (to be merged:
    every chunk after the first drops its first ``prefix_len`` elements)."""

    def __init__(self, chunk_len: int, prefix_len: int, original_data: Optional[torch.Tensor] = None,
                 chunk_data: Optional[Sequence[torch.Tensor]] = None, dim: int = 0):
        assert chunk_len > prefix_len
        self.chunk_len = chunk_len
        self.prefix_len = prefix_len
        self.dim = dim
        self._original_data = original_data
        self._chunk_data = None if chunk_data is None else list(chunk_data)

    @property
    def data(self) -> torch.Tensor:
        if self._original_data is not None:
            return self._original_data
        parts = [self._chunk_data[0]]
        for x in self._chunk_data[1:]:
            parts.append(x.narrow(self.dim, self.prefix_len, x.shape[self.dim] - self.prefix_len))
        return torch.cat(parts, dim=self.dim)

    @property
    def chunk_data(self) -> List[torch.Tensor]:
        if self._chunk_data is not None:
            return self._chunk_data
        n = self._original_data.shape[self.dim]
        chunks = []
        for i in range(0, n, self.chunk_len):
            start = 0 if i == 0 else i - self.prefix_len
            stop = min(n, i + self.chunk_len)
            chunks.append(self._original_data.narrow(self.dim, start, stop - start))
        return chunks


def plan(hop: int, process_window: int, prefix_tokens: int):
    """(chunk_len, prefix_len) in samples: the window rounded down to whole hops (codec.py:135), the overlap in whole hops."""
    chunk_len = process_window // hop * hop
    prefix_len = prefix_tokens * hop
    if chunk_len <= prefix_len:
        raise ValueError(f"process_window ({process_window} samples) must exceed the overlap ({prefix_len} samples)")
    return chunk_len, prefix_len
