"""CPU restatement of the reference's long-audio chunk bookkeeping (TEST INFRASTRUCTURE ONLY; nothing under ``l3ac_amd/``
imports this file).

``ChunkData`` follows reference l3ac/codec.py:159-188 line by line (dim 0, as the reference).  ``extract_unit`` /
``decode_unit`` follow the control flow of codec.py:124-156 (pad, window rounded to whole hops, one call per chunk, token-domain
ChunkData with chunk_len / prefix_len divided by the hop, merge) with the two corrections the product makes and documents
(l3ac_amd/chunking.py): each chunk runs the oracle's FULL encode / decode path (``en_encoder`` / ``en_decoder`` included, which
reference ``Codec.compress`` / ``decompress`` :113-122 skip), and the overlap is ``prefix_tokens`` tokens instead of one hop.
With ``prefix_tokens=1`` the split / merge indices are exactly the reference's.

PINNED by tests/golden/chunk_kat.npz (tests/test_oracle_golden.py::test_chunk_oracle_*): ``ChunkData`` against the reference's own
class on integer sequences, and ``conv_extract_unit`` / ``conv_decode_unit`` — the same control flow with the reference's
``Codec.compress`` / ``decompress`` (conv codec only, hop = prod(compress_rates), overlap one hop) in place of the full path —
against the reference's ``Codec.extract_unit`` / ``decode_unit`` run on the tiny model.
"""
import math

import torch

from . import l3ac_oracle as O


class ChunkData:
    def __init__(self, chunk_len, prefix_len, original_data=None, chunk_data=None):
        assert chunk_len > prefix_len  # codec.py:161
        self.chunk_len = chunk_len
        self.prefix_len = prefix_len
        self._original_data = original_data
        self._chunk_data = chunk_data

    @property
    def data(self):  # codec.py:167-175
        if self._original_data is not None:
            return self._original_data
        original_data = [self._chunk_data[0]]
        for x in self._chunk_data[1:]:
            original_data.append(x[self.prefix_len:])
        return torch.cat(original_data, dim=0)

    @property
    def chunk_data(self):  # codec.py:177-188
        if self._chunk_data is not None:
            return self._chunk_data
        chunk_data = []
        for i in range(0, len(self._original_data), self.chunk_len):
            if i == 0:
                chunk_data.append(self._original_data[:self.chunk_len])
            else:
                chunk_data.append(self._original_data[i - self.prefix_len:i + self.chunk_len])
        return chunk_data


@torch.no_grad()
def extract_unit(w, mc, audio_data, process_window, prefix_tokens):
    assert len(audio_data) == 1  # codec.py:133
    audio_data, _ = O.preprocess(mc, audio_data)  # :134
    hop = mc.hop_length
    process_window = process_window // hop * hop  # :135
    chunk_audio = ChunkData(chunk_len=process_window, prefix_len=prefix_tokens * hop, original_data=audio_data[0])  # :137
    chunk_indices, chunk_q_feature, chunk_latents = [], [], []
    for x in chunk_audio.chunk_data:  # :139-142
        taps = {}
        q_feature, ind = O.encode_audio(w, mc, x[None, :], taps=taps)
        chunk_indices.append(ind["indices"][0])
        chunk_q_feature.append(q_feature[0])
        chunk_latents.append(taps["latents"][0])
    n, p = process_window // hop, prefix_tokens  # :144
    return (ChunkData(chunk_len=n, prefix_len=p, chunk_data=chunk_indices), ChunkData(chunk_len=n, prefix_len=p, chunk_data=chunk_q_feature),
            ChunkData(chunk_len=n, prefix_len=p, chunk_data=chunk_latents))


@torch.no_grad()
def decode_unit(w, mc, chunk_indices):
    hop = mc.hop_length
    chunk_audio = [O.decode_audio(w, mc, indices=x[None, :])[0] for x in chunk_indices.chunk_data]  # :150-151
    return ChunkData(chunk_len=chunk_indices.chunk_len * hop, prefix_len=chunk_indices.prefix_len * hop, chunk_data=chunk_audio).data[None, :]  # :155-156


# ---- the reference's own variant (conv codec only), for pinning the bookkeeping against reference-run fixtures -------------
def _conv_hop(mc):
    hop = 1
    for r in mc.compress_rates:
        hop *= r
    return hop  # codec.py:27-30 (base ModelConfig.hop_length)


@torch.no_grad()
def conv_extract_unit(w, mc, audio_data, process_window=5 * 16000):
    """reference Codec.extract_unit (codec.py:124-147) on the base ``Codec``: compress = encoder -> quantizer (:113-116)."""
    assert len(audio_data) == 1  # :133
    fill = _conv_hop(mc)  # fill_length (:75-77)
    length = audio_data.shape[-1]
    audio_data = torch.nn.functional.pad(audio_data, (0, math.ceil(length / fill) * fill - length))  # preprocess (:79-84)
    process_window = process_window // fill * fill  # :135
    chunk_audio = ChunkData(chunk_len=process_window, prefix_len=fill, original_data=audio_data[0])  # :137
    chunk_indices, chunk_q_feature = [], []
    for x in chunk_audio.chunk_data:  # :139-142
        feature = O.encoder(w, mc, x[None, None, :]).permute(0, 2, 1)
        q_feature, ind, _ = O.quantizer(w, mc, feature)
        chunk_indices.append(ind["indices"][0])
        chunk_q_feature.append(q_feature[0])
    n, p = process_window // fill, fill // fill  # :144
    return ChunkData(chunk_len=n, prefix_len=p, chunk_data=chunk_indices), ChunkData(chunk_len=n, prefix_len=p, chunk_data=chunk_q_feature)


@torch.no_grad()
def conv_decode_unit(w, mc, chunk_indices):
    """reference Codec.decode_unit (codec.py:149-156): decompress = to_features -> decoder (:118-122)."""
    chunk_audio = [O.decoder(w, mc, O.to_features(w, mc, x[None, :]).permute(0, 2, 1))[0, 0] for x in chunk_indices.chunk_data]  # :151
    return ChunkData(chunk_len=len(chunk_audio[0]), prefix_len=_conv_hop(mc), chunk_data=chunk_audio).data[None, :]  # :155-156
