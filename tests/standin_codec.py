"""Stand-in codec for `bench.py --dry-run-cpu` (test infrastructure): the product's method surface — `codec.network.mc`,
`codec.config.sample_rate`, `encode_audio`, `decode_audio` — computed by the oracle on CPU with the tiny golden config.  It exists
so that the N-rank code of bench.py (self-launch, rendezvous, barriers, gathers, max-over-ranks timing, JSON relay) can run with
two processes in the CPU test suite; it is never a measurement and never part of the product path."""
import types


def make_codec():
    from oracle import l3ac_oracle as O
    from tests.helpers import load_case
    mc, w, _, _ = load_case("tiny")

    class Codec:
        network = types.SimpleNamespace(mc=mc)
        config = types.SimpleNamespace(sample_rate=16000)

        def encode_audio(self, audio):
            return O.encode_audio(w, mc, audio)

        def decode_audio(self, audio_feature=None, indices=None):
            return O.decode_audio(w, mc, audio_feature=audio_feature, indices=indices)

    return Codec()


def make_shape_codec():
    """BASELINE config 4 at its REAL shape on CPU ranks: the 1kbps geometry (hop 270 -> 60 tokens and 16 200 output samples per 1 s clip,
    128 features, the shipped levels) with trivial arithmetic in place of the network — a deterministic token per hop window and a
    waveform that repeats it — so that 8 gloo ranks x 256 clips run in seconds.  What it exercises is bench.py's N-rank plumbing at the
    sizes the 8-GPU run will have (gathers of [2048, 60] int32 and [2048, 16200] fp32); no number it produces means anything."""
    import math

    import torch

    from l3ac_amd.config import L3ACConfig, resolve_config_file
    mc = L3ACConfig(config_file=resolve_config_file("1kbps")).network_config
    hop, feat, k = mc.hop_length, mc.feature_dim, mc.codebook_size

    class Codec:
        network = types.SimpleNamespace(mc=mc)
        config = types.SimpleNamespace(sample_rate=16000)

        def encode_audio(self, audio):
            b, t = audio.shape
            n_tok = math.ceil(t / hop)
            x = torch.nn.functional.pad(audio, (0, n_tok * hop - t)).view(b, n_tok, hop)
            idx = (x.abs().sum(-1) * 1000.0).to(torch.int64).remainder(k).to(torch.int32)
            q = (idx.to(torch.float32) / k).unsqueeze(-1).expand(b, n_tok, feat).contiguous()
            return q, {"indices": idx, "level_indices": torch.zeros(b, n_tok, len(mc.levels))}

        def decode_audio(self, audio_feature=None, indices=None):
            src = audio_feature[..., 0] if audio_feature is not None else indices.to(torch.float32) / k
            return src.repeat_interleave(hop, dim=1).contiguous()

    return Codec()
