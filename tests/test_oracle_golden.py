"""The oracle against the committed reference vectors (tests/golden, made by make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import l3ac_oracle as O
from tests.helpers import GOLDEN, index_mismatch_report, load_case, seeded_audio, strided

ATOL, RTOL = 2e-5, 2e-5  # fp32 conv stacks on a different host CPU may pick other mkldnn kernels
TAU = 1e-4  # level units; a flipped index must come from a latent this close to its rounding boundary


def _check(name, got, fx, full):
    got = got.detach()
    if full:
        np.testing.assert_allclose(got.numpy(), fx[name], atol=ATOL, rtol=RTOL, err_msg=name)
    else:
        assert tuple(got.shape) == tuple(fx[name + "_shape"]), name
        np.testing.assert_allclose(strided(got).numpy(), fx[name + "_strided"], atol=ATOL, rtol=RTOL, err_msg=name)
        assert abs(got.double().abs().sum().item() - float(fx[name + "_abssum"])) <= 1e-5 * float(fx[name + "_abssum"]) + 1e-3


@pytest.mark.parametrize("tag", ["tiny", "1kbps", "3kbps", "stress_1kbps", "stress_3kbps"])
def test_conv_stacks_and_quantizer_match_reference(tag):
    mc, w, conv, _ = load_case(tag)
    full = tag == "tiny"
    audio = seeded_audio(int(conv["batch"]), int(conv["samples"]))
    with torch.inference_mode():
        x, length = O.preprocess(mc, audio)
        assert x.shape[-1] == int(conv["padded_len"]) and length == int(conv["orig_len"])
        feature = O.encoder(w, mc, x.unsqueeze(1))
        _check("feature", feature, conv, full)
        q_feat, ind, lat = O.quantizer(w, mc, feature.permute(0, 2, 1))
        _check("latents", lat, conv, full)
        n_bad, ok = index_mismatch_report(ind["indices"], conv["indices"], conv["latents"] if full else lat, mc.levels, TAU)
        assert ok and n_bad <= 1, f"{n_bad} index mismatches"  # observed in the build container: 0
        if n_bad == 0:
            np.testing.assert_array_equal(ind["level_indices"].numpy(), conv["level_indices"])
            _check("q_feat", q_feat, conv, full)
        # decode from the REFERENCE indices so that the decoder check does not depend on boundary flips
        q_ref = O.to_features(w, mc, torch.from_numpy(conv["indices"]))
        wave = O.decoder(w, mc, q_ref.permute(0, 2, 1)).squeeze(1)
        _check("wave", wave, conv, full)


def test_tiny_per_block_outputs_match_reference():
    mc, w, conv, _ = load_case("tiny")
    audio = seeded_audio(int(conv["batch"]), int(conv["samples"]))
    with torch.inference_mode():
        x, _ = O.preprocess(mc, audio)
        taps = {}
        O.encoder(w, mc, x.unsqueeze(1), taps=taps)
        np.testing.assert_allclose(taps["enc.first"].numpy(), conv["enc_block0"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(taps["enc.stage0"].numpy(), conv["enc_block1"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(taps["enc.down0"].numpy(), conv["enc_block2"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(taps["enc.down1"].numpy(), conv["enc_block4"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(taps["enc.tail"].numpy(), conv["enc_block5"], atol=ATOL, rtol=RTOL)
        dt = {}
        O.decoder(w, mc, torch.from_numpy(conv["q_feat"]).permute(0, 2, 1), taps=dt)
        np.testing.assert_allclose(dt["dec.in"].numpy(), conv["dec_block0"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(dt["dec.stage0"].numpy(), conv["dec_block1"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(dt["dec.enh0"].numpy(), conv["dec_block2"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(dt["dec.up0"].numpy(), conv["dec_block3"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(dt["dec.up1"].numpy(), conv["dec_block6"], atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("tag", ["tiny", "1kbps", "3kbps", "stress_1kbps", "stress_3kbps"])
def test_end_to_end_wiring_matches_reference(tag):
    """Reference EnCodec wiring (local_trans.py) with the stand-in attention: pins wiring, not attention maths."""
    mc, w, conv, e2e = load_case(tag)
    full = tag == "tiny"
    audio = seeded_audio(int(e2e["batch"]), int(e2e["samples"]))
    taps = {}
    q_feat, ind = O.encode_audio(w, mc, audio, taps=taps)
    _check("trans", taps["en_encoder.out"], e2e, full)
    _check("latents", taps["latents"], e2e, full)
    n_bad, ok = index_mismatch_report(ind["indices"], e2e["indices"], taps["latents"], mc.levels, TAU)
    assert ok and n_bad <= 1  # observed in the build container: 0
    wave = O.decode_audio(w, mc, indices=torch.from_numpy(e2e["indices"]))
    _check("wave", wave, e2e, full)
    if n_bad == 0:
        _check("q_trans", q_feat, e2e, full)
        wave2 = O.decode_audio(w, mc, audio_feature=q_feat)
        assert torch.equal(wave2, wave)  # reference: to_features(indices) == q_feat exactly


@pytest.mark.parametrize("window,n", [(8, 21), (8, 8), (8, 5), (16, 40), (250, 60)])
def test_bucketed_attention_equals_dense_form(window, n):
    """the bucketed look-back-1 algorithm == causal attention restricted to own/previous window."""
    mc, w, _, _ = load_case("tiny")
    g = torch.Generator().manual_seed(window * 1000 + n)
    x = torch.randn(2, n, mc.feature_dim, generator=g)
    with torch.inference_mode():
        a = O.local_trans(w, "en_decoder.local_trans", x, window, 2)
        b = O.local_trans_dense(w, "en_decoder.local_trans", x, window, 2)
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-6, rtol=1e-5)


def test_fsq_known_answers():
    kat = np.load(GOLDEN / "fsq_kat.npz")
    for tag in ("l7", "l9977", "even", "tiny"):
        levels = kat[f"{tag}_levels"].tolist()
        z = torch.from_numpy(kat[f"{tag}_z"])
        q, idx, li = O.fsq_quantize(z, levels)
        np.testing.assert_array_equal(idx.numpy(), kat[f"{tag}_indices"])
        np.testing.assert_array_equal(li.numpy(), kat[f"{tag}_level_indices"])
        np.testing.assert_array_equal(q.numpy(), kat[f"{tag}_q"])
        codes = O.fsq_indices_to_codes(torch.from_numpy(kat[f"{tag}_dec_idx"]), levels)
        np.testing.assert_array_equal(codes.numpy(), kat[f"{tag}_dec_codes"])
        # closed form == nearest neighbour over the explicit codebook (ties aside; none in these vectors)
        _, idx_q, _ = O.fsq_quantize(torch.atanh(torch.from_numpy(kat[f"{tag}_nn_query"]).double()).float(), levels)
        cb = O.codebook(levels)
        nn = torch.cdist(torch.from_numpy(kat[f"{tag}_nn_query"]), cb).argmin(dim=1).to(torch.int32)
        np.testing.assert_array_equal(nn.numpy(), kat[f"{tag}_nn_idx"])
    # half-to-even at exact ties: tanh(0) = 0 -> act = 0.5 -> 0.5, 1.5, 2.5, 3.5 -> 0, 2, 2, 4
    _, _, li = O.fsq_quantize(torch.zeros(1, 4), [2, 4, 6, 8])
    assert li.tolist() == [[0.0, 2.0, 2.0, 4.0]]


def test_fsq_rounding_boundaries():
    """Exact k + 0.5 products and their +-1 / +-2 ulp neighbours (reference answers: SuperFSQ.quantize_act_value)."""
    kat = np.load(GOLDEN / "fsq_boundary_kat.npz")
    for tag in ("l7", "l9977", "even", "tiny"):
        levels = kat[f"{tag}_levels"].tolist()
        q, idx, li = O.fsq_quantize_act(torch.from_numpy(kat[f"{tag}_act"]), levels)
        np.testing.assert_array_equal(li.numpy(), kat[f"{tag}_level_indices"])
        np.testing.assert_array_equal(idx.numpy(), kat[f"{tag}_indices"])
        np.testing.assert_array_equal(q.numpy(), kat[f"{tag}_q"])
        # the fixture really sits on the boundaries: some products are exactly k + 0.5, and both roundings occur
        prod = kat[f"{tag}_act"] * (np.asarray(levels, dtype=np.float32) - 1)
        assert ((prod - np.floor(prod)) == 0.5).any()


def test_chunk_oracle_chunkdata_matches_reference():
    """oracle/chunk_oracle.py::ChunkData and the product's l3ac_amd.chunking.ChunkData against the reference's own class
    (tests/golden/chunk_kat.npz, written by make_golden.py running l3ac.codec.ChunkData): cut lengths and contents, merge of the
    cut, merge of unrelated chunks (ragged tails, a single chunk, prefix = chunk_len - 1)."""
    from l3ac_amd.chunking import ChunkData as ProductChunkData
    from oracle.chunk_oracle import ChunkData
    kat = np.load(GOLDEN / "chunk_kat.npz")
    for i, (n, cl, pl) in enumerate(kat["cd_cases"]):
        data = torch.arange(int(n), dtype=torch.int64) * 3 + 1
        for cls in (ChunkData, ProductChunkData):
            chunks = cls(chunk_len=int(cl), prefix_len=int(pl), original_data=data).chunk_data
            assert [len(c) for c in chunks] == kat[f"cd{i}_lens"].tolist()
            np.testing.assert_array_equal(torch.cat(chunks).numpy(), kat[f"cd{i}_cat"])
            np.testing.assert_array_equal(cls(chunk_len=int(cl), prefix_len=int(pl), chunk_data=chunks).data.numpy(), kat[f"cd{i}_merged"])
            other = torch.from_numpy(kat[f"cd{i}_other_cat"]).split(kat[f"cd{i}_lens"].tolist())
            np.testing.assert_array_equal(cls(chunk_len=int(cl), prefix_len=int(pl), chunk_data=list(other)).data.numpy(),
                                          kat[f"cd{i}_other_merged"])


def test_chunk_oracle_extract_unit_matches_reference():
    """The oracle's restatement of the reference's chunk plan (window rounded to whole hops, one-hop overlap, token-domain
    ChunkData, merge) against Codec.extract_unit / decode_unit run by make_golden.py on the tiny conv codec: chunk geometry
    exactly, tokens exactly, features and waveform to fp32 tolerance."""
    from oracle import chunk_oracle as CO
    kat = np.load(GOLDEN / "chunk_kat.npz")
    assert kat["extract_unit_errors_as_written"].tolist() == ["KeyError", "AttributeError"]  # the method as shipped cannot run
    mc, w, _, _ = load_case("tiny")
    assert int(kat["eu_seed"]) == 3
    for j, (samples, window) in enumerate(kat["eu_cases"]):
        audio = seeded_audio(1, int(samples), seed=77 + j)
        ci, cq = CO.conv_extract_unit(w, mc, audio, process_window=int(window))
        assert ci.chunk_len == int(kat[f"eu{j}_chunk_len"]) and ci.prefix_len == int(kat[f"eu{j}_prefix_len"])
        assert [len(c) for c in ci.chunk_data] == kat[f"eu{j}_chunk_tokens"].tolist()
        np.testing.assert_array_equal(ci.data.numpy(), kat[f"eu{j}_indices"])
        np.testing.assert_allclose(cq.data.numpy(), kat[f"eu{j}_q_feature"], atol=ATOL, rtol=RTOL)
        wave = CO.conv_decode_unit(w, mc, ci)
        np.testing.assert_allclose(wave.numpy(), kat[f"eu{j}_wave"], atol=ATOL, rtol=RTOL)
    # prefix_tokens = 1 on the full-path variant gives the same split points as the reference's plan (one hop of overlap)
    from l3ac_amd.chunking import plan
    assert plan(12, 500, 1) == (492, 12)
