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
