"""End-to-end parity of the drop-in surface (encode_audio / decode_audio) against the oracle and the committed
reference vectors, plus size-independent properties at the BASELINE batch size (`pytest -m gpu`)."""
import numpy as np
import pytest
import torch

import l3ac_amd
from l3ac_amd import weights as W
from oracle import l3ac_oracle as O
from tests import gpu_ops as G
from tests.helpers import GOLDEN, index_agreement, index_mismatch_report, load_case, seeded_audio, strided

pytestmark = pytest.mark.gpu

TAU = 1e-4        # a flipped index must come from a latent within TAU of a rounding boundary (in level units)
# Tolerances: MAX gates = the largest error observed on the MI355X for the cases of this file x ~2 (the values come from one box, one
# hipcc and one seed set, and a different summation order — a new kernel geometry, a compiler update — moves the worst sample by tens
# of % without any defect; the decoder amplifies the ~1e-7 rounding noise of ANY evaluation order to a few 1e-4 at the tanh output),
# each backed by an RMS gate at ~2 x the observed rms: a real regression moves the rms of every clip, not one sample.
# Observed = profiles/r05/pytest_gpu.log (every test prints it).
WAVE_ATOL = 1.1e-3      # 1 s clips, decoder given identical indices (observed max <= 6.1e-4: 1kbps exact route vs the reference vector)
WAVE_RMS = 9e-5         # ... and their rms error (observed <= 4.4e-5)
WAVE_ATOL_LONG = 3e-3   # 6.5 s clips and chunked long audio (observed <= 1.47e-3)
WAVE_RMS_LONG = 1.5e-4  # (observed <= 6.9e-5)
WAVE_ATOL_ROUTES = 1.5e-3 # bf16x3 route against exact-fp32 route, same tokens (observed <= 7.0e-4)
FEAT_ATOL = 1.3e-5      # encoder / transformer features, activations O(1) (observed <= 6.0e-6)
FEAT_RMS = 2.5e-6       # (observed <= 1.2e-6)


def _max_err(name, got, ref):
    """max |got - ref| over numpy arrays, printed (the observed value the tolerances are set from)."""
    e = float(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)).max())
    print(f"[{name}] max|err|={e:.3e}")
    return e


def _codec(tag, seed, grn_exact=False):
    profile = "stress" if tag.startswith("stress_") else "mild"  # stress_<config>: trained-weight statistics (weights.py)
    tag = tag[len("stress_"):] if profile == "stress" else tag
    cfg = GOLDEN / f"{tag}.toml" if tag in ("tiny", "refdefault") else tag  # refdefault: the reference ModelConfig's default geometry
    codec = l3ac_amd.get_model(cfg, synthetic_seed=seed, synthetic_profile=profile)
    codec.network.grn_exact = grn_exact
    codec.network.to(device="cuda").eval()
    return codec


def _err(name, got, ref, rms_tol=None):
    """max |got - ref| (printed with the rms); with `rms_tol` the rms error is gated here as well."""
    e = (got.detach().cpu().double() - ref.detach().cpu().double()).abs()
    rms = e.pow(2).mean().sqrt().item()
    print(f"[{name}] max|err|={e.max().item():.3e} rms={rms:.3e} max|ref|={ref.abs().max().item():.3e}")
    if rms_tol is not None:
        assert rms < rms_tol, f"{name}: rms error {rms:.3e} >= {rms_tol:.1e}"
    return e.max().item()


@pytest.mark.parametrize("tag,seed,batch,samples", [("tiny", 3, 3, 250), ("tiny", 3, 2, 1201), ("1kbps", 0, 2, 16000),
                                                    ("3kbps", 0, 2, 16000), ("1kbps", 0, 1, 5000), ("refdefault", 5, 2, 9000)])
def test_submodules_against_oracle(tag, seed, batch, samples):
    codec = _codec(tag, seed)
    ctx = codec.network.context()
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(batch, samples)
    x, _ = O.preprocess(mc, audio)
    frames = x.shape[-1]
    enc_rate = mc.hop_length // mc.en_coder_compress_rate
    with torch.inference_mode():
        feat_ref = O.encoder(w, mc, x.unsqueeze(1))
        tok_ref = O.en_encoder(w, mc, feat_ref)
        q_ref, ind_ref, lat_ref = O.quantizer(w, mc, tok_ref)
        dec_in_ref = O.en_decoder(w, mc, q_ref)
        wave_ref = O.decoder(w, mc, dec_in_ref).squeeze(1)
    feat = G.op_plain(ctx, "l3ac_op_encoder", x.cuda(), batch, frames, (batch, frames // enc_rate, mc.feature_dim))
    assert _err(f"{tag} encoder", G.from_frames(feat), feat_ref, FEAT_RMS) < FEAT_ATOL
    tok = G.op_plain(ctx, "l3ac_op_en_encoder", G.to_frames(feat_ref), batch, frames // enc_rate,
                     (batch, frames // mc.hop_length, mc.feature_dim))
    assert _err(f"{tag} en_encoder", tok.cpu(), tok_ref, FEAT_RMS) < FEAT_ATOL
    dec_in = G.op_plain(ctx, "l3ac_op_en_decoder", q_ref.cuda(), batch, frames // mc.hop_length,
                        (batch, frames // enc_rate, mc.feature_dim))
    assert _err(f"{tag} en_decoder", G.from_frames(dec_in), dec_in_ref, FEAT_RMS) < FEAT_ATOL
    wave = G.op_plain(ctx, "l3ac_op_decoder", G.to_frames(dec_in_ref), batch, frames // enc_rate, (batch, frames))
    assert _err(f"{tag} decoder", wave.cpu(), wave_ref, WAVE_RMS) < WAVE_ATOL


@pytest.mark.parametrize("tag,seed,batch,samples", [("tiny", 3, 4, 250), ("1kbps", 0, 4, 16000), ("3kbps", 0, 3, 16000),
                                                    ("1kbps", 0, 2, 16001), ("1kbps", 0, 3, 100), ("0k75bps", 1, 2, 8000),
                                                    ("1k5bps", 1, 2, 8000),
                                                    # ragged sizes: one sample, one short of / one past a hop, a tile-unfriendly length
                                                    ("1kbps", 0, 1, 1), ("1kbps", 0, 2, 269), ("1kbps", 0, 1, 271), ("1kbps", 0, 5, 8191),
                                                    ("3kbps", 0, 1, 97),
                                                    # either side of the longest clip the one-workgroup transformer stack takes (192 frames)
                                                    ("1kbps", 0, 2, 17280), ("1kbps", 0, 2, 17281), ("3kbps", 0, 2, 18432), ("3kbps", 0, 2, 18433),
                                                    # the reference's default geometry: 128-channel fused stage, 64 / 32-channel units
                                                    ("refdefault", 5, 3, 9000), ("refdefault", 5, 1, 46)])
def test_encode_decode_against_oracle(tag, seed, batch, samples):
    codec = _codec(tag, seed)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(batch, samples)
    taps = {}
    q_ref, ind_ref = O.encode_audio(w, mc, audio, taps=taps)
    q, ind = codec.encode_audio(audio.cuda())
    n_tok = -(-samples // mc.hop_length)
    assert q.shape == (batch, n_tok, mc.feature_dim) and q.dtype == torch.float32
    assert ind["indices"].shape == (batch, n_tok) and ind["indices"].dtype == torch.int32
    assert ind["level_indices"].shape == (batch, n_tok, len(mc.levels)) and ind["level_indices"].dtype == torch.float32
    n_bad, ok = index_mismatch_report(ind["indices"].cpu().numpy(), ind_ref["indices"].numpy(), taps["latents"].numpy(),
                                      mc.levels, TAU)
    total = batch * n_tok
    print(f"[{tag} B={batch} T={samples}] index mismatches vs oracle: {n_bad}/{total}")
    assert ok, "an index differs by more than one level or away from a rounding boundary"
    assert n_bad <= 1  # observed on every case: 0
    # decoder, given the ORACLE's indices: waveform within tolerance
    wave_ref = O.decode_audio(w, mc, indices=ind_ref["indices"])
    wave = codec.decode_audio(indices=ind_ref["indices"].cuda())
    assert wave.shape == (batch, n_tok * mc.hop_length)
    assert _err(f"{tag} wave(from oracle indices)", wave, wave_ref, WAVE_RMS) < WAVE_ATOL
    # decode_audio(q_feature) and decode_audio(indices=...) agree bit for bit on the device's own outputs
    wave_a = codec.decode_audio(q)
    wave_b = codec.decode_audio(indices=ind["indices"])
    assert torch.equal(wave_a, wave_b)


@pytest.fixture
def gemm_mode(request):
    """Runs a test under one GEMM route: "split" (default: bf16x3 operands on the bf16 matrix cores) or "exact"
    (every product on v_mfma_f32_32x32x2_f32)."""
    default, routes = l3ac_amd.get_gemm_split(), l3ac_amd.gemm_split_routes()
    l3ac_amd.set_gemm_split(request.param == "split")
    yield request.param
    l3ac_amd.restore_gemm_split_routes(routes, default=default)  # every live network gets ITS route back, not the default


@pytest.mark.parametrize("gemm_mode", ["split", "exact"], indirect=True)
@pytest.mark.parametrize("tag", ["tiny", "1kbps", "3kbps", "stress_1kbps", "stress_3kbps"])
def test_against_committed_reference_vectors(tag, gemm_mode):
    """tests/golden/*_e2e.npz were produced by the reference's own EnCodec wiring (see make_golden.py); stress_*: the same with
    the trained-statistics weight profile."""
    mc, w, conv, e2e = load_case(tag)
    codec = _codec(tag, int(e2e["seed"]))
    audio = seeded_audio(int(e2e["batch"]), int(e2e["samples"]))
    q, ind = codec.encode_audio(audio.cuda())
    lat = O.encode_audio(w, mc, audio, taps=(taps := {})) and taps["latents"]
    n_bad, ok = index_mismatch_report(ind["indices"].cpu().numpy(), e2e["indices"], lat.numpy(), mc.levels, TAU)
    print(f"[{tag}] index mismatches vs reference vectors: {n_bad}/{e2e['indices'].size}")
    assert ok and n_bad == 0  # the metric says "indices bit-exact": against the COMMITTED reference vectors a flipped token is a regression (observed: 0)
    wave = codec.decode_audio(indices=torch.from_numpy(e2e["indices"]).cuda()).cpu()
    if tag == "tiny":
        assert _max_err(f"{tag} {gemm_mode} wave vs reference e2e vector", wave.numpy(), e2e["wave"]) < WAVE_ATOL
    else:
        assert _max_err(f"{tag} {gemm_mode} wave vs reference e2e vector", strided(wave).numpy(), e2e["wave_strided"]) < WAVE_ATOL
    # conv-stack-only vectors (reference code only): encoder feature and decoder waveform
    ctx = codec.network.context()
    x, _ = O.preprocess(mc, audio)
    enc_rate = mc.hop_length // mc.en_coder_compress_rate
    feat = G.op_plain(ctx, "l3ac_op_encoder", x.cuda(), x.shape[0], x.shape[1], (x.shape[0], x.shape[1] // enc_rate, mc.feature_dim))
    feat = G.from_frames(feat)
    if tag == "tiny":
        assert _max_err(f"{tag} {gemm_mode} encoder feature vs reference vector", feat.numpy(), conv["feature"]) < FEAT_ATOL
        q_feat = torch.from_numpy(conv["q_feat"])
    else:
        # FEAT_ATOL is for activations of O(1) (mild profile: max |feature| ~ 3); the stress profile's reach 10
        tol = FEAT_ATOL * max(1.0, float(np.abs(conv["feature_strided"]).max()) / 3.0)
        assert _max_err(f"{tag} {gemm_mode} encoder feature vs reference vector", strided(feat).numpy(), conv["feature_strided"]) < tol
        q_feat = O.to_features(w, mc, torch.from_numpy(conv["indices"]))
    wave = G.op_plain(ctx, "l3ac_op_decoder", q_feat.cuda().contiguous(), q_feat.shape[0], q_feat.shape[1],
                      (q_feat.shape[0], q_feat.shape[1] * enc_rate)).cpu()
    if tag == "tiny":
        assert _max_err(f"{tag} {gemm_mode} decoder wave vs reference vector", wave.numpy(), conv["wave"]) < WAVE_ATOL
    else:
        assert _max_err(f"{tag} {gemm_mode} decoder wave vs reference vector", strided(wave).numpy(), conv["wave_strided"]) < WAVE_ATOL


def test_split_and_exact_gemm_routes_agree():
    """Same model, same clips, both GEMM routes: identical tokens; the waveforms differ by no more than either differs
    from the CPU oracle (the decoder amplifies fp32 rounding noise of ANY evaluation order to a few 1e-4)."""
    codec = _codec("1kbps", 0)
    audio = seeded_audio(8, 16000).cuda()
    out = {}
    before = codec.network.gemm_split  # this network's own route (its context's state)
    try:
        for mode in (True, False):
            codec.network.set_gemm_split(mode)
            q, ind = codec.encode_audio(audio)
            out[mode] = (ind["indices"].cpu(), codec.decode_audio(indices=ind["indices"]).cpu())
    finally:
        codec.network.set_gemm_split(before)
    n_diff = int((out[True][0] != out[False][0]).sum())
    err = (out[True][1] - out[False][1]).abs().max().item()
    print(f"[split vs exact] token differences {n_diff}/{out[True][0].numel()}, waveform max |diff| {err:.3e}")
    assert n_diff == 0
    assert err <= WAVE_ATOL_ROUTES


def test_full_batch_properties_1kbps():
    """BASELINE config 2 (1kbps, 256 x 1 s): properties that need no oracle at this size."""
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    audio = seeded_audio(256, 16000).cuda()
    q, ind = codec.encode_audio(audio)
    idx = ind["indices"]
    assert idx.shape == (256, 60) and int(idx.min()) >= 0 and int(idx.max()) < mc.codebook_size
    # clips are independent: any clip processed alone gives bit-identical tokens and waveform
    for b in (0, 101, 255):
        q1, ind1 = codec.encode_audio(audio[b:b + 1])
        assert torch.equal(ind1["indices"], idx[b:b + 1]) and torch.equal(q1, q[b:b + 1])
    wave = codec.decode_audio(indices=idx)
    assert wave.shape == (256, 16200) and torch.isfinite(wave).all() and float(wave.abs().max()) <= 1.0
    for b in (3, 200):
        assert torch.equal(codec.decode_audio(indices=idx[b:b + 1]), wave[b:b + 1])
    # ... and so do the small batches in between, which take other launch forms stage by stage (round 5: the wide ConvUnits' sliced form
    # up to 256 frame tiles — 4 clips at C = 256, 23 at C = 192 —, the streamed GEMMs, the stem's split form, the cooperative stacks)
    for lo, n in ((10, 2), (40, 4), (90, 5), (120, 13), (200, 24)):
        qn, indn = codec.encode_audio(audio[lo:lo + n])
        assert torch.equal(indn["indices"], idx[lo:lo + n]) and torch.equal(qn, q[lo:lo + n]), f"{n} clips from {lo}"
        assert torch.equal(codec.decode_audio(indices=idx[lo:lo + n]), wave[lo:lo + n]), f"{n} clips from {lo}: waveform"
    # deterministic
    q2, ind2 = codec.encode_audio(audio)
    assert torch.equal(ind2["indices"], idx) and torch.equal(q2, q)
    # right zero-padding is what the kernels assume (codec.py:79-84)
    padded = torch.nn.functional.pad(audio[:8], (0, 200))
    q3, ind3 = codec.encode_audio(padded)
    assert torch.equal(ind3["indices"], idx[:8]) and torch.equal(q3, q[:8])
    # level_indices <-> indices
    basis = torch.cumprod(torch.tensor([1] + list(mc.levels)[:-1]), 0).cuda()
    assert torch.equal((ind["level_indices"].long() * basis).sum(-1).int(), idx)
    # a sample of the batch against the oracle
    w = W.folded_weights(codec.network.state_dicts())
    sel = [0, 77, 255]
    taps = {}
    _, ind_ref = O.encode_audio(w, mc, audio[sel].cpu(), taps=taps)
    n_bad, ok = index_mismatch_report(idx[sel].cpu().numpy(), ind_ref["indices"].numpy(), taps["latents"].numpy(), mc.levels, TAU)
    assert ok and n_bad == 0  # the BASELINE batch: strict (observed: 0)


from tests.helpers import ORACLE_CHUNK, oracle_indices as _oracle_indices  # (each chunk of clips encoded once per session)


# mismatches observed on the MI355X for these exact batches (seed 1234, synthetic weights seed 0), both GEMM routes
OBSERVED_FULL_BATCH_MISMATCHES = {"1kbps": 0, "3kbps": 0}


@pytest.mark.parametrize("tag", ["1kbps", "3kbps"])
def test_index_agreement_full_batch(tag):
    """Every token of the BASELINE batches (config 2: 1kbps 256 x 1 s = 15 360 tokens; config 3: 3kbps 256 x 1 s = 42 752
    tokens) against the oracle, on both GEMM routes.  These are the batches the headline metric ("indices bit-exact") is quoted on: the
    count must EQUAL the observed one — 0 — and a flipped token fails the test (round 5 allowed observed + 1)."""
    codec = _codec(tag, 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(256, 16000)
    idx_ref, lat_ref = _oracle_indices(w, mc, audio)
    before = codec.network.gemm_split  # this network's own route (its context's state)
    try:
        for route in (True, False):
            codec.network.set_gemm_split(route)
            _, ind = codec.encode_audio(audio.cuda())
            rep = index_agreement(ind["indices"].cpu().numpy(), idx_ref.numpy(), lat_ref.numpy(), mc.levels)
            print(f"[index agreement {tag} {'split' if route else 'exact'}] {rep}")
            assert rep["tokens"] == 256 * (-(-16000 // mc.hop_length))
            assert rep["single_step"] and rep["max_margin_of_mismatches"] < TAU
            assert rep["mismatches"] == OBSERVED_FULL_BATCH_MISMATCHES[tag]
    finally:
        codec.network.set_gemm_split(before)


# observed on the MI355X with the stress profile (round 4; printed by the test): index mismatches per (config, route) and the
# largest waveform error given identical indices
OBSERVED_STRESS_MISMATCHES = {}
STRESS_WAVE_ATOL = 3.7e-3  # observed <= 1.84e-3 (stress_1kbps, split route, 48 clips, unsaturated output; round 5: 1.22e-3)
STRESS_WAVE_RMS = 8e-5     # rms of the same error (observed <= 3.9e-5)


@pytest.mark.parametrize("tag", ["stress_1kbps", "stress_3kbps"])
def test_parity_under_trained_weight_statistics(tag):
    """VERDICT r3 item 7.  Every other parity number is on the mild synthetic weights (trunc-normal 0.02, parameters near their
    init).  A trained network has what that profile lacks — weight-norm gains with a heavy tail, snake alpha far from 1 (large sine
    arguments, large 1/alpha), GRN gamma / beta of O(1), heavy-tailed biases, latents in tanh's saturation: `profile="stress"`
    (l3ac_amd/weights.py; oracle pinned to the reference on these weights by tests/golden/stress_*.npz).  32 noise clips + 16
    structured clips, both GEMM routes: index agreement with the oracle (flips reported with their margins), waveform error given
    identical indices, how far the snake arguments go (the +-1e5 clamp of the kernels' sine), and the GRN fast path's guard."""
    from tests.helpers import structured_audio
    codec = _codec(tag, 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = torch.cat([seeded_audio(32, 16000, seed=11), structured_audio(2, 16000, seed=12)[0]])
    seen = {"max_arg": 0.0, "over_clamp": 0, "n": 0}
    orig_snake = O.snake

    def counting_snake(x, alpha):
        a = (x * alpha).abs()
        seen["max_arg"] = max(seen["max_arg"], float(a.max()))
        seen["over_clamp"] += int((a > 1e5).sum())
        seen["n"] += a.numel()
        return orig_snake(x, alpha)
    O.snake = counting_snake
    try:
        idx_ref, lat_ref = _oracle_indices(w, mc, audio, use_cache=False)  # (the counting hook must see the ENCODER run, in any test order)
        enc_evals = seen["n"]
        assert enc_evals > 0, "the oracle's encoder did not run under the counting snake"
        wave_ref = torch.cat([O.decode_audio(w, mc, indices=idx_ref[b0:b0 + ORACLE_CHUNK]) for b0 in range(0, len(idx_ref), ORACLE_CHUNK)])
    finally:
        O.snake = orig_snake
    print(f"[{tag}] snake arguments seen by the oracle: max |alpha x| = {seen['max_arg']:.1f} over {seen['n']} evaluations, "
          f"{seen['over_clamp']} beyond the kernels' 1e5 clamp")
    assert seen["over_clamp"] == 0  # were it ever non-zero the clamp's 1e-5 relative bound would be what is compared
    lv = torch.tensor(mc.levels)
    li = (idx_ref.unsqueeze(-1) // torch.tensor(np.concatenate([[1], np.cumprod(mc.levels[:-1])]))) % lv
    sat = float(((li == 0) | (li == lv - 1)).float().mean())
    print(f"[{tag}] latents: std {float(lat_ref.std()):.2f}, max |z| {float(lat_ref.abs().max()):.2f}, outermost-level fraction {sat:.3f}")
    assert sat >= 0.10
    before = codec.network.gemm_split
    try:
        for route in (True, False):
            name = "split" if route else "exact"
            codec.network.set_gemm_split(route)
            _, ind = codec.encode_audio(audio.cuda())
            rep = index_agreement(ind["indices"].cpu().numpy(), idx_ref.numpy(), lat_ref.numpy(), mc.levels)
            print(f"[index agreement {tag} {name}] {rep}")
            assert rep["single_step"] and rep["max_margin_of_mismatches"] < TAU
            # strict since round 6 (the arithmetic is deterministic on this hardware: the count is a property of the build): the bf16x3 down
            # layers (L3AC_DOWN_FUSED=1), which move ONE stress_3kbps token across its boundary, fail here — profiles/r06/strict_gates.txt
            assert rep["mismatches"] == OBSERVED_STRESS_MISMATCHES.get((tag, name), 0)
            wave = codec.decode_audio(indices=idx_ref.cuda()).cpu()
            err = (wave - wave_ref).abs()
            print(f"[{tag} {name}] waveform given identical indices: max|err| {float(err.max()):.3e} rms {float(err.pow(2).mean().sqrt()):.3e} "
                  f"(|wave| max {float(wave_ref.abs().max()):.3f}, {float((wave_ref.abs() > 0.999).float().mean()):.3f} saturated)")
            assert float(err.max()) < STRESS_WAVE_ATOL and float(err.pow(2).mean().sqrt()) < STRESS_WAVE_RMS
    finally:
        codec.network.set_gemm_split(before)
    # the GRN fast path (n = g / (g + 1e-8) taken as 1) under gamma / beta of O(1): the validation mode evaluates the literal formula,
    # must give the same tokens, and its smallest per-clip norm must stay above the 0.25 at which the two are bit-identical
    exact = _codec(tag, 0, grn_exact=True)
    _, ind_x = exact.encode_audio(audio.cuda())
    exact.decode_audio(indices=ind_x["indices"])
    g_min = exact.network.min_grn_norm()
    _, ind_f = codec.encode_audio(audio.cuda())
    print(f"[{tag}] smallest GRN norm {g_min:.3f}; tokens of the fast path equal the literal formula's: {bool(torch.equal(ind_x['indices'], ind_f['indices']))}")
    assert g_min >= 0.25 and torch.equal(ind_x["indices"], ind_f["indices"])


# (round 5 needed this test to run after the stress test, which had to be the one to fill the session cache; round 6: the stress test
# bypasses the cache for its hooked evaluation, so the order no longer matters)
# option "down_fused": observed mismatches with the narrow encoder down layers in their one-kernel bf16x3 form (value 1)
OBSERVED_DOWN_FUSED_MISMATCHES = {"1kbps": 0, "stress_3kbps": 1}


@pytest.mark.parametrize("tag", ["1kbps", "stress_3kbps"])
def test_index_agreement_with_fused_down_layers(tag):
    """The encoder down layers 24 -> 48 and 48 -> 96 (Conv1d(k = stride) + ChannelNorm) have two one-kernel forms (context option
    "down_fused").  Value 2, the DEFAULT since round 6 (down_exact_kernel), evaluates the arithmetic of the GEMM + row kernel it replaces
    bit for bit: tokens AND q_feature of a whole encode equal those of value 0.  Value 1 (the bf16x3 form, round 4) is a different —
    equally accurate — rounding of the encoder's first layers: the index contract holds with it (single-level flips within TAU of a
    rounding boundary only) and what it costs is printed: on the stress weights the one decision that lies 1.9e-6 level units from its
    boundary falls on the other side — which is why it never became the default, and what a strict gate must catch (profiles/r06/strict_gates.txt)."""
    from tests.helpers import structured_audio
    codec = _codec(tag, 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = (torch.cat([seeded_audio(32, 16000, seed=11), structured_audio(2, 16000, seed=12)[0]]) if tag.startswith("stress")
             else seeded_audio(64, 16000))
    idx_ref, lat_ref = _oracle_indices(w, mc, audio)
    ctx = codec.network.context()
    q_default, default = codec.encode_audio(audio.cuda())  # value 2
    try:
        ctx.set_option("down_fused", 0)
        q_plain, plain = codec.encode_audio(audio.cuda())
        ctx.set_option("down_fused", 1)
        _, ind = codec.encode_audio(audio.cuda())
    finally:
        ctx.set_option("down_fused", 2)
    assert torch.equal(default["indices"], plain["indices"]) and torch.equal(q_default, q_plain), "down_exact_kernel changed the encoder's bits"
    rep = index_agreement(ind["indices"].cpu().numpy(), idx_ref.numpy(), lat_ref.numpy(), mc.levels)
    differ = int((ind["indices"] != plain["indices"]).sum())
    print(f"[index agreement {tag} down_fused=1] {rep}; tokens that differ from the default form's: {differ}")
    assert rep["single_step"] and rep["max_margin_of_mismatches"] < TAU
    assert rep["mismatches"] <= OBSERVED_DOWN_FUSED_MISMATCHES[tag] + 1


# mismatches observed on the MI355X per (config, input set), both GEMM routes (round 3; every one a +-1 flip within TAU).  The one entry
# is an exact tie: the latent lies 9.4e-9 level units from its rounding boundary, below fp32 resolution of the scaled value; its flat token
# number (clip * tokens per clip + token) is asserted below, so a DIFFERENT flipped token cannot hide behind the count.
OBSERVED_WIDE_MISMATCHES = {("3kbps", "noise seed 2"): 1}
KNOWN_TIES = {("3kbps", "noise seed 2"): 9694}  # clip 58, token 8 (167 tokens per clip), both routes
OBSERVED_WIDE_WAVE_ERR = 1.4e-3  # structured set, 64 clips, both routes: observed <= 7.2e-4 (x ~2, see WAVE_ATOL)


@pytest.mark.parametrize("tag", ["1kbps", "3kbps"])
def test_index_agreement_other_seeds_and_structured_inputs(tag):
    """Round-2 verdict: one draw of white noise says little.  Three further noise seeds (64 clips each) and a structured set
    (chirps, sines, harmonic stacks, bursts after digital silence, DC offsets, hard-clipped +-1.0, 0.005 and 1.0 amplitudes; 8
    clips each) are encoded on both GEMM routes and compared token for token with the oracle; the decoder is compared on the
    structured set given the oracle's tokens.  Reported per set: flips, their margins, how close the set's latents come to a
    boundary."""
    from tests.helpers import structured_audio
    codec = _codec(tag, 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    sets = {f"noise seed {sd}": seeded_audio(64, 16000, seed=sd) for sd in (1, 2, 3)}
    sets["structured"], kinds = structured_audio(8, 16000)
    before = codec.network.gemm_split  # this network's own route (its context's state)
    try:
        for name, audio in sets.items():
            idx_ref, lat_ref = _oracle_indices(w, mc, audio)
            for route in (True, False):
                codec.network.set_gemm_split(route)
                _, ind = codec.encode_audio(audio.cuda())
                rep = index_agreement(ind["indices"].cpu().numpy(), idx_ref.numpy(), lat_ref.numpy(), mc.levels)
                print(f"[index agreement {tag} {name} {'split' if route else 'exact'}] {rep}")
                assert rep["single_step"] and rep["max_margin_of_mismatches"] < TAU
                # random-seed / structured sweeps: the observed count + 1 (a tie may fall either way on another summation order) ...
                assert rep["mismatches"] <= OBSERVED_WIDE_MISMATCHES.get((tag, name), 0) + 1
                # ... but the known tie must be THE flipped token when it is the only one
                known = KNOWN_TIES.get((tag, name))
                if known is not None and rep["mismatches"] == 1:
                    assert rep["mismatch_positions"] == [known], f"a different token flipped: {rep['mismatch_positions']} (known tie: {known})"
                if name == "structured":
                    wave = codec.decode_audio(indices=idx_ref.cuda()).cpu()
                    wave_ref = torch.cat([O.decode_audio(w, mc, indices=idx_ref[b0:b0 + ORACLE_CHUNK]) for b0 in range(0, len(idx_ref), ORACLE_CHUNK)])
                    err = (wave - wave_ref).abs()
                    per_kind = {k: float(err[[i for i, kk in enumerate(kinds) if kk == k]].max()) for k in dict.fromkeys(kinds)}
                    print(f"[structured wave {tag} {'split' if route else 'exact'}] max err per kind: "
                          + ", ".join(f"{k} {v:.2e}" for k, v in per_kind.items()))
                    assert float(err.max()) < OBSERVED_WIDE_WAVE_ERR
    finally:
        codec.network.set_gemm_split(before)


def test_batch_invariance_3kbps_256():
    """3kbps at the BASELINE batch: the clip-group scheduling of the wide stages depends on the geometry (T = 167 / 668 /
    2672 frames), so batch invariance is asserted here as well as for 1kbps."""
    codec = _codec("3kbps", 0)
    mc = codec.network.mc
    audio = seeded_audio(256, 16000).cuda()
    q, ind = codec.encode_audio(audio)
    idx = ind["indices"]
    assert idx.shape == (256, 167) and int(idx.min()) >= 0 and int(idx.max()) < mc.codebook_size
    wave = codec.decode_audio(q)
    assert wave.shape == (256, 16032) and torch.isfinite(wave).all() and float(wave.abs().max()) <= 1.0
    for b in (0, 51, 52, 130, 255):  # clips on both sides of the clip-group boundaries
        q1, ind1 = codec.encode_audio(audio[b:b + 1])
        assert torch.equal(ind1["indices"], idx[b:b + 1]) and torch.equal(q1, q[b:b + 1])
        assert torch.equal(codec.decode_audio(q1), wave[b:b + 1])
    q64, ind64 = codec.encode_audio(audio[64:128])
    assert torch.equal(ind64["indices"], idx[64:128]) and torch.equal(codec.decode_audio(q64), wave[64:128])
    for lo, n in ((7, 2), (33, 5), (140, 12)):  # the few-clip launch forms of round 5 (sliced wide units, streamed GEMMs) at this geometry
        qn, indn = codec.encode_audio(audio[lo:lo + n])
        assert torch.equal(indn["indices"], idx[lo:lo + n]) and torch.equal(codec.decode_audio(qn), wave[lo:lo + n]), f"{n} clips from {lo}"


@pytest.mark.parametrize("tag", ["0k75bps", "1k5bps"])
def test_batch_invariance_and_agreement_other_configs_256(tag):
    """SURVEY §8 f2 at the headline batch: the other two shipped configs (hop 360: 135 frames / 45 tokens per second; hop 180:
    178 frames / 89 tokens) cut their clip groups and attention windows differently from 1kbps / 3kbps.  A clip of the 256-batch
    must be bit-identical alone, and the first 32 clips' tokens are compared with the oracle."""
    codec = _codec(tag, 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(256, 16000)
    q, ind = codec.encode_audio(audio.cuda())
    idx = ind["indices"]
    n_tok = -(-16000 // mc.hop_length)
    assert idx.shape == (256, n_tok) and int(idx.min()) >= 0 and int(idx.max()) < mc.codebook_size
    wave = codec.decode_audio(q)
    assert wave.shape == (256, n_tok * mc.hop_length) and torch.isfinite(wave).all() and float(wave.abs().max()) <= 1.0
    for b in (0, 77, 130, 255):
        q1, ind1 = codec.encode_audio(audio[b:b + 1].cuda())
        assert torch.equal(ind1["indices"], idx[b:b + 1]) and torch.equal(q1, q[b:b + 1])
        assert torch.equal(codec.decode_audio(q1), wave[b:b + 1])
    for lo, n in ((20, 3), (100, 9)):
        qn, indn = codec.encode_audio(audio[lo:lo + n].cuda())
        assert torch.equal(indn["indices"], idx[lo:lo + n]) and torch.equal(codec.decode_audio(qn), wave[lo:lo + n]), f"{n} clips from {lo}"
    idx_ref, lat_ref = _oracle_indices(w, mc, audio[:32])
    rep = index_agreement(idx[:32].cpu().numpy(), idx_ref.numpy(), lat_ref.numpy(), mc.levels)
    print(f"[index agreement {tag} b256, first 32 clips] {rep}")
    assert rep["single_step"] and rep["max_margin_of_mismatches"] < TAU and rep["mismatches"] <= 1  # observed: 0
    wave_ref = O.decode_audio(w, mc, indices=idx_ref[:8])
    err = float((codec.decode_audio(indices=idx_ref[:8].cuda()).cpu() - wave_ref).abs().max())
    print(f"[{tag} wave given the oracle's tokens, 8 clips] max err {err:.3e}")
    assert err < WAVE_ATOL


def test_decoder_before_tanh_full_size():
    """The tanh output saturates with the synthetic weights (|wave| reaches 1.0), which hides pre-tanh error.  With the
    l3ac_ctx_set_head_pretanh validation switch the head stores the Conv1d(24 -> 1, k7) result itself: compared here at full clip size
    against the oracle's pre-tanh value, relative to the size of the signal."""
    from l3ac_amd import _capi
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(4, 16000)
    _, ind_ref = O.encode_audio(w, mc, audio)
    wave_ref = O.decode_audio(w, mc, indices=ind_ref["indices"]).double()
    clipped = wave_ref.abs() > 0.999
    pre_ref = torch.atanh(wave_ref.clamp(-0.999, 0.999))  # where the oracle's tanh has not saturated, atanh recovers its input
    other = _codec("1kbps", 0)  # a second context: the switch is per context and must not reach it
    ctx = codec.network.context()
    other_before = other.decode_audio(indices=ind_ref["indices"].cuda())
    ctx.set_head_pretanh(True)
    try:
        pre = codec.decode_audio(indices=ind_ref["indices"].cuda()).cpu().double()
        # isolation: the other context's output is what it was before the switch, bit for bit (|wave| <= 1 alone could hide a leak)
        assert torch.equal(other.decode_audio(indices=ind_ref["indices"].cuda()), other_before)
    finally:
        ctx.set_head_pretanh(False)
    wave = codec.decode_audio(indices=ind_ref["indices"].cuda()).cpu().double()
    assert torch.equal(torch.tanh(pre.float()).double(), wave) or (torch.tanh(pre) - wave).abs().max() < 1e-6
    ok = ~clipped
    err = (pre - pre_ref)[ok].abs()
    scale = pre_ref[ok].abs().clamp_min(1.0)
    print(f"[pre-tanh] unsaturated samples {int(ok.sum())}/{ok.numel()}, max|pre|={float(pre.abs().max()):.2f}, "
          f"max err {float(err.max()):.3e}, max err/scale {float((err / scale).max()):.3e}, rms {float(err.pow(2).mean().sqrt()):.3e}")
    assert ok.float().mean() > 0.5
    assert float((err / scale).max()) < 2e-3
    # saturated samples: the pre-tanh value must be large with the right sign
    assert (pre[clipped].sign() == wave_ref[clipped].sign()).all() and (pre[clipped].abs() > 3.0).all()


def test_batch_2048_on_one_gpu():
    """BASELINE config 4 moved onto one device (2048 x 1 s): workspace sizing, clip-group scheduling of the wide stages and
    batch invariance at 8x the headline batch."""
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    audio = seeded_audio(2048, 16000).cuda()
    q, ind = codec.encode_audio(audio)
    idx = ind["indices"]
    assert idx.shape == (2048, 60) and int(idx.min()) >= 0 and int(idx.max()) < mc.codebook_size
    wave = codec.decode_audio(q)
    assert wave.shape == (2048, 16200) and torch.isfinite(wave).all()
    for b in (0, 1023, 2047):  # any clip alone: identical tokens and samples
        q1, ind1 = codec.encode_audio(audio[b:b + 1])
        assert torch.equal(ind1["indices"], idx[b:b + 1])
        assert torch.equal(codec.decode_audio(q1), wave[b:b + 1])
    # the first 256 clips as their own batch (the headline shape) give the same results
    q256, ind256 = codec.encode_audio(audio[:256])
    assert torch.equal(ind256["indices"], idx[:256]) and torch.equal(codec.decode_audio(q256), wave[:256])


def test_edge_cases():
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    # digital silence: biases keep the activations alive; GRN normaliser must still be 1
    silent = torch.zeros(2, 4000)
    taps = {}
    _, ind_ref = O.encode_audio(w, mc, silent, taps=taps)
    q, ind = codec.encode_audio(silent.cuda())
    n_bad, ok = index_mismatch_report(ind["indices"].cpu().numpy(), ind_ref["indices"].numpy(), taps["latents"].numpy(), mc.levels, TAU)
    assert ok and n_bad <= 1  # observed: 0
    wave = codec.decode_audio(indices=ind_ref["indices"].cuda())
    assert _err("silence wave", wave, O.decode_audio(w, mc, indices=ind_ref["indices"])) < WAVE_ATOL
    # full-scale input, and a non-contiguous view
    loud = seeded_audio(2, 3000) * 2.0
    q, ind = codec.encode_audio(loud.cuda())
    taps = {}
    _, ind_ref = O.encode_audio(w, mc, loud, taps=taps)
    n_bad, ok = index_mismatch_report(ind["indices"].cpu().numpy(), ind_ref["indices"].numpy(), taps["latents"].numpy(), mc.levels, TAU)
    assert ok and n_bad <= 1  # observed: 0
    big = seeded_audio(4, 6001).cuda()
    view = big[::2, 1:6001]
    qa, ia = codec.encode_audio(view)
    qb, ib = codec.encode_audio(view.contiguous())
    assert torch.equal(ia["indices"], ib["indices"]) and torch.equal(qa, qb)
    # exact GRN (two-pass g / (g + eps)) agrees with the fast path on ordinary input
    exact = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    exact.network.grn_exact = True
    exact.network.to(device="cuda").eval()
    x = seeded_audio(2, 8000).cuda()
    q1, i1 = codec.encode_audio(x)
    q2, i2 = exact.encode_audio(x)
    assert torch.equal(i1["indices"], i2["indices"])
    assert _err("grn exact vs fast (features)", q1, q2) < 1e-5
    # two GPU summation orders (fused units vs GEMM route) of the same decoder, both checked against the oracle elsewhere: observed
    # 0.8e-4 .. 1.03e-4 depending on the build (any change upstream re-rolls the last bits), an order below either one's oracle error
    assert _err("grn exact vs fast (wave)", codec.decode_audio(q1), exact.decode_audio(q1)) < 2e-4
    # the guard of the fast path: in validation mode the context reports the smallest GRN norm it has seen — far above the 0.25
    # below which g / (g + 1e-8) stops being exactly 1.0f, for ordinary audio and for digital silence alike
    assert exact.network.min_grn_norm() > 0.25
    exact.network.min_grn_norm(reset=True)
    exact.decode_audio(exact.encode_audio(torch.zeros(1, 4000).cuda())[0])
    g_silence = exact.network.min_grn_norm()
    print(f"[grn guard] smallest ||x|| over all GRN layers on digital silence: {g_silence:.3e}")
    assert 0.25 < g_silence < float("inf")
    with pytest.raises(RuntimeError):
        codec.network.min_grn_norm()  # only the validation mode tracks it
    # errors are loud
    with pytest.raises(RuntimeError):
        codec.encode_audio(torch.zeros(1, 1000))  # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        codec.encode_audio(torch.zeros(1000).cuda())
    with pytest.raises(ValueError):
        codec.decode_audio()
    # indices from outside: out-of-range values are counted and clamped, never decomposed into wrapped levels
    good = codec.encode_audio(seeded_audio(1, 4000).cuda())[1]["indices"]
    bad = good.clone()
    bad[0, 3] = mc.codebook_size + 5
    bad[0, 7] = -1
    ctx = codec.network.context()
    ctx.bad_index_count(reset=True)
    wave_bad = codec.decode_audio(indices=bad)
    assert ctx.bad_index_count(reset=True) == 2 and torch.isfinite(wave_bad).all()
    clamped = good.clone()
    clamped[0, 3] = mc.codebook_size - 1
    clamped[0, 7] = 0
    assert torch.equal(wave_bad, codec.decode_audio(indices=clamped))
    with pytest.raises(ValueError, match="outside"):
        codec.decode_audio(indices=bad, validate=True)
    codec.decode_audio(indices=good, validate=True)


def test_long_clip_multi_window_attention():
    """A 6.5 s clip has more frames (1 170) than the 1kbps attention windows (750 / 250): exercises the look-back-one-
    window masking end to end, and a ragged length (not a multiple of the hop)."""
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(2, 104001)
    taps = {}
    q_ref, ind_ref = O.encode_audio(w, mc, audio, taps=taps)
    q, ind = codec.encode_audio(audio.cuda())
    n_bad, ok = index_mismatch_report(ind["indices"].cpu().numpy(), ind_ref["indices"].numpy(), taps["latents"].numpy(),
                                      mc.levels, TAU)
    print(f"[long clip] index mismatches vs oracle: {n_bad}/{ind_ref['indices'].numel()}")
    assert ok and n_bad <= 1  # observed: 0
    wave = codec.decode_audio(indices=ind_ref["indices"].cuda())
    assert _err("long clip wave", wave, O.decode_audio(w, mc, indices=ind_ref["indices"]), WAVE_RMS_LONG) < WAVE_ATOL_LONG


def test_weights_from_disk_and_example_flow(tmp_path):
    """SURVEY f4.  Five ``.pt`` state dicts in the reference's cache layout ({model_dir}/{name}.{version}/{module}.pt with the
    weight-norm parametrisation keys, l3ac/__init__.py:70-72, xtract/nn/module.py:43-54) are written to a temporary directory and
    loaded with ``get_model(name, model_dir=...)``: the results must be bit-identical to the ``synthetic_seed`` route that
    every other test uses.  Then the flow of reference example.py:7-30 / README.md:36-67, line by line."""
    from l3ac_amd.config import L3ACConfig, resolve_config_file
    cfg = L3ACConfig(config_file=resolve_config_file("1kbps"), model_dir=tmp_path)
    W.save_state_dicts(W.synthetic_state_dicts(cfg.network_config, seed=0), cfg.model_path)
    assert sorted(f.name for f in cfg.model_path.iterdir()) == sorted(f"{m}.pt" for m in W.MODULE_NAMES)
    # ---- example.py -------------------------------------------------------------------------------------------------
    assert "1kbps" in l3ac_amd.list_models() and "3kbps" in l3ac_amd.list_models()       # example.py:8
    codec = l3ac_amd.get_model("1kbps", model_dir=tmp_path)                               # :9
    assert codec.config.sample_rate == 16000                                              # :10
    info = l3ac_amd.get_model_info(codec.network)                                         # :11
    assert info["macs"] > 3.8e10 and info["params"] > 1.1e7 and abs(info["bps"] - 998.2) < 0.1
    sample_audio = seeded_audio(1, 40000).numpy()                                         # :13-17 (librosa sample -> synthetic clip)
    codec.network.to(device="cuda")                                                       # :19
    codec.network.eval()                                                                  # :20
    with torch.inference_mode():                                                          # :21
        audio_in = torch.tensor(sample_audio, dtype=torch.float32, device="cuda")         # :22
        _, audio_length = audio_in.shape                                                  # :23
        q_feature, indices = codec.encode_audio(audio_in)                                 # :25
        audio_out = codec.decode_audio(q_feature)                                         # :26
        audio_out2 = codec.decode_audio(indices=indices["indices"])                       # :27
        generated_audio = audio_out[:, :audio_length].detach().cpu().numpy()              # :28
    assert generated_audio.shape == sample_audio.shape and np.isfinite(generated_audio).all()
    mse = float(((sample_audio - generated_audio) ** 2).mean())                           # :30
    assert np.isfinite(mse)
    assert torch.equal(audio_out, audio_out2)
    # ---- same weights through the seeded generator: bit-identical tokens, features and waveform -------------------------
    ref = _codec("1kbps", 0)
    q2, ind2 = ref.encode_audio(torch.tensor(sample_audio, device="cuda"))
    assert torch.equal(ind2["indices"], indices["indices"]) and torch.equal(q2, q_feature)
    assert torch.equal(ref.decode_audio(q2), audio_out)
    # a missing file is an error, not silently random weights (the reference only logs, module.py:52-54)
    (cfg.model_path / "en_decoder.pt").unlink()
    with pytest.raises(FileNotFoundError):
        l3ac_amd.get_model("1kbps", model_dir=tmp_path)


@pytest.mark.parametrize("tag,seed", [("tiny", 3), ("1kbps", 0)])
def test_integration_snippet_runs_as_written(tag, seed):
    """INTEGRATION.md §2 (the ctypes stub a reference maintainer would add as l3ac/_hip.py), extracted from the document and run as
    written: `folded_tensors()` walks module trees with real weight-norm parametrizations carrying the reference's state-dict keys
    (strict load; tests/helpers.py::module_tree_from_state_dict — the names were compared with the reference's own EnCodec in the
    build container, tests/test_host.py), `HipPath` creates a context through `l3ac_create` with the snippet's own `_Cfg` / `_Tensor`
    and its `encode` / `decode` must return what the package's drop-in surface returns, bit for bit."""
    import types

    from tests.helpers import integration_snippet, module_tree_from_state_dict
    codec = _codec(tag, seed)
    mc = codec.network.mc
    sds = codec.network.state_dicts()
    net = types.SimpleNamespace(mc=mc, trainable_modules={m: module_tree_from_state_dict(sds[m]) for m in W.MODULE_NAMES})
    ns = integration_snippet()
    path = ns["HipPath"](net, torch.cuda.current_device())
    audio = seeded_audio(3, 250 if tag == "tiny" else 16000).cuda()
    q, ind = path.encode(audio)
    q0, ind0 = codec.encode_audio(audio)
    assert torch.equal(ind["indices"], ind0["indices"]) and torch.equal(q, q0) and torch.equal(ind["level_indices"], ind0["level_indices"])
    assert torch.equal(path.decode(feature=q), codec.decode_audio(q0))
    assert torch.equal(path.decode(indices=ind["indices"]), codec.decode_audio(indices=ind0["indices"]))
    # a wrong ABI version is refused with a message, through the snippet's own error path
    cfg = ns["_Cfg"](abi_version=2)
    out = ns["C"].c_void_p()
    with pytest.raises(RuntimeError, match="(?i)abi"):
        ns["_check"](ns["_lib"].l3ac_create(ns["C"].byref(cfg), None, 0, 0, ns["C"].byref(out)))
    ns["_lib"].l3ac_destroy(path.ctx)


def test_long_audio_chunker():
    """extract_unit / decode_unit (reference l3ac/codec.py:124-156, corrected to run en_encoder / en_decoder on every
    chunk and to overlap by the attention look-back) against (i) the oracle run chunk by chunk with the reference's ChunkData
    bookkeeping, (ii) plain encode_audio when one window covers the clip, (iii) the whole-clip oracle in the valid region."""
    from oracle import chunk_oracle as CO
    codec = _codec("1kbps", 0)
    mc = codec.network.mc
    hop = mc.hop_length
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(1, 6 * 16000 + 123)  # 6 s: several windows of 2 s, ragged tail
    for window, prefix in ((2 * 16000, 30), (2 * 16000, 1), (3 * 16000 + 77, 60)):
        ci, cq = codec.extract_unit(audio.cuda(), process_window=window, prefix_tokens=prefix)
        ri, rq, rl = CO.extract_unit(w, mc, audio, window, prefix)
        assert len(ci.chunk_data) == len(ri.chunk_data) and ci.chunk_len == ri.chunk_len and ci.prefix_len == ri.prefix_len
        assert [tuple(x.shape) for x in ci.chunk_data] == [tuple(x.shape) for x in ri.chunk_data]
        n_bad = 0
        for a, b, lat in zip(ci.chunk_data, ri.chunk_data, rl.chunk_data):
            nb, ok = index_mismatch_report(a.cpu().numpy(), b.numpy(), lat.numpy(), mc.levels, TAU)
            assert ok
            n_bad += nb
        assert n_bad <= 1  # observed: 0
        tokens = ci.data
        assert tokens.shape == (-(-audio.shape[1] // hop),) and torch.equal(tokens.cpu(), ri.data) or n_bad > 0
        assert _err(f"chunked q_feature w={window} p={prefix}", cq.data, rq.data) < FEAT_ATOL * 4
        # decode the ORACLE's chunks on the GPU and compare the stitched waveform with the oracle's
        wave = codec.decode_unit(chunk_indices=l3ac_amd.ChunkData(ri.chunk_len, ri.prefix_len, chunk_data=[x.cuda() for x in ri.chunk_data]),
                                 audio_length=audio.shape[1])
        ref = CO.decode_unit(w, mc, ri)[:, :audio.shape[1]]
        assert wave.shape == (1, audio.shape[1])
        assert _err(f"chunked wave w={window} p={prefix}", wave, ref, WAVE_RMS_LONG) < WAVE_ATOL_LONG
        # from q_feature chunks == from index chunks, bit for bit
        assert torch.equal(codec.decode_unit(chunk_q_feature=cq), codec.decode_unit(chunk_indices=ci))
    # one window covers the clip: identical to the plain call
    ci, cq = codec.extract_unit(audio.cuda(), process_window=8 * 16000)
    q, ind = codec.encode_audio(audio.cuda())
    assert len(ci.chunk_data) == 1 and torch.equal(ci.data, ind["indices"][0]) and torch.equal(cq.data, q[0])
    assert torch.equal(codec.decode_unit(chunk_indices=ci), codec.decode_audio(indices=ind["indices"]))
    # valid region vs the whole-clip oracle: a chunk that sees one attention window of left context reproduces most of the
    # whole-clip tokens (not all: the stacked layers reach further back than one window, and the clip-wide statistics differ)
    _, ind_ref = O.encode_audio(w, mc, audio)
    ci, _ = codec.extract_unit(audio.cuda(), process_window=2 * 16000, prefix_tokens=60)
    agree = float((ci.data.cpu() == ind_ref["indices"][0]).float().mean())
    ci1, _ = codec.extract_unit(audio.cuda(), process_window=2 * 16000, prefix_tokens=1)
    agree1 = float((ci1.data.cpu() == ind_ref["indices"][0]).float().mean())
    print(f"[chunker] tokens equal to the whole-clip oracle: {agree:.3f} with the look-back overlap, {agree1:.3f} with the reference's one-hop overlap")
    assert agree >= agree1 and agree > 0.5


def test_streaming_chunks_and_graph_capture():
    """BASELINE config 5: 1 s chunks through a captured graph give the same tokens as eager calls."""
    codec = _codec("1kbps", 0)
    ctx = codec.network.context()
    ctx.reserve(1, 16000)
    chunks = seeded_audio(6, 16000).cuda()
    eager = [codec.encode_audio(chunks[i:i + 1]) for i in range(6)]
    eager_wave = [codec.decode_audio(indices=e[1]["indices"]) for e in eager]
    static_in = torch.zeros(1, 16000, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        codec.encode_audio(static_in)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q, ind = codec.encode_audio(static_in)
        wave = codec.decode_audio(indices=ind["indices"])
    for i in range(6):
        static_in.copy_(chunks[i:i + 1])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ind["indices"], eager[i][1]["indices"])
        assert torch.equal(wave, eager_wave[i])
    # the decode-from-features entry copies its input into the workspace (a device-to-device copy node): replayed repeatedly too
    static_q = eager[0][0].clone()
    eager_wave_q = [codec.decode_audio(e[0]) for e in eager]
    with torch.cuda.stream(s):
        codec.decode_audio(static_q)
    torch.cuda.current_stream().wait_stream(s)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        wave_q = codec.decode_audio(static_q)
    for i in (3, 1, 4, 1):
        static_q.copy_(eager[i][0])
        graph2.replay()
        torch.cuda.synchronize()
        assert torch.equal(wave_q, eager_wave_q[i])


def test_batch_graph_replays_with_the_unit_counters():
    """Round 6: at a batch the C = 96 ConvUnits and the LegacyUnits take their tiles from device counters that every launch leaves zeroed
    (no memset node).  A captured graph of encode + decode of 32 clips must therefore replay — again and again, on new inputs — to the bits
    of the eager calls, and to those of a context that uses static shares."""
    codec = _codec("1kbps", 0)
    ctx = codec.network.context()
    ctx.reserve(32, 16000)
    clips = seeded_audio(96, 16000).cuda()
    ctx.set_option("unit_counter", 0)
    try:
        static_shares = [codec.decode_audio(indices=codec.encode_audio(clips[32 * i:32 * i + 32])[1]["indices"]) for i in range(3)]
    finally:
        ctx.set_option("unit_counter", 1)
    eager = [codec.encode_audio(clips[32 * i:32 * i + 32]) for i in range(3)]
    eager_wave = [codec.decode_audio(indices=e[1]["indices"]) for e in eager]
    for i in range(3):
        assert torch.equal(eager_wave[i], static_shares[i]), f"batch {i}: tiles by counter differ from static shares"
    static_in = torch.zeros(32, 16000, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q, ind = codec.encode_audio(static_in)
        wave = codec.decode_audio(indices=ind["indices"])
    for i in (0, 1, 2, 1, 0):
        static_in.copy_(clips[32 * i:32 * i + 32])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ind["indices"], eager[i][1]["indices"]), f"replay of batch {i}: tokens differ from the eager call"
        assert torch.equal(wave, eager_wave[i]), f"replay of batch {i}: waveform differs from the eager call"


def test_token_bit_packing_roundtrip():
    """Wire format: little-endian bit stream, ceil(log2 K) bits per token, checked against numpy."""
    for tag, bits in (("1kbps", 17), ("3kbps", 18)):
        codec = _codec(tag, 0)
        mc = codec.network.mc
        assert l3ac_amd.bits_per_token(mc) == bits
        g = torch.Generator().manual_seed(5)
        for n_tok in (1, 7, 60, 167, 1000):
            idx = torch.randint(0, mc.codebook_size, (5, n_tok), generator=g, dtype=torch.int32)
            idx[0, 0] = mc.codebook_size - 1
            packed = l3ac_amd.pack_indices(idx.cuda(), bits)
            assert packed.dtype == torch.uint8 and packed.shape == (5, 4 * (-(-n_tok * bits // 32)))
            # numpy reference: token t occupies stream bits [t*bits, (t+1)*bits), bit b = bit b%8 of byte b//8
            ref = np.zeros((5, packed.shape[1] * 8), dtype=np.uint8)
            for t in range(n_tok):
                for k in range(bits):
                    ref[:, t * bits + k] = (idx[:, t].numpy() >> k) & 1
            ref_bytes = np.packbits(ref, axis=1, bitorder="little")
            np.testing.assert_array_equal(packed.cpu().numpy(), ref_bytes)
            back = l3ac_amd.unpack_indices(packed, n_tok, bits)
            assert torch.equal(back.cpu(), idx)
    # end to end: audio -> tokens -> bytes -> tokens -> audio is bit-identical to decoding the tokens directly
    codec = _codec("1kbps", 0)
    q, ind = codec.encode_audio(seeded_audio(3, 16000).cuda())
    packed = l3ac_amd.pack_indices(ind["indices"], 17)
    assert packed.shape[1] * 8 / 1.0125 <= 1024 + 8  # 60 tokens * 17 bits = 1020 bits for 1.0125 s ~ 998 bps + padding
    wave = codec.decode_audio(indices=l3ac_amd.unpack_indices(packed, 60, 17))
    assert torch.equal(wave, codec.decode_audio(indices=ind["indices"]))


def test_two_streams_alternate_on_one_context():
    """include/l3ac_hip.h: a context belongs to one call at a time, and a call on another stream than the previous one first waits
    for it (hipStreamWaitEvent) — so two torch streams may alternate on one codec without host synchronisation in between and without
    racing on the shared workspace: every result equals the single-stream one."""
    codec = _codec("1kbps", 0)
    xs = [seeded_audio(3, 12000 + 500 * i).cuda() for i in range(4)]
    ref = []
    for x in xs:
        q, ind = codec.encode_audio(x)
        ref.append((ind["indices"].clone(), codec.decode_audio(indices=ind["indices"]).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for rep in range(3):
        for i, x in enumerate(xs):
            st = streams[i & 1]
            with torch.cuda.stream(st):
                q, ind = codec.encode_audio(x)
            other = streams[1 - (i & 1)]
            other.wait_stream(st)  # (the decode below consumes tensors produced on `st`: ordinary producer/consumer ordering)
            with torch.cuda.stream(other):
                wave = codec.decode_audio(indices=ind["indices"])
            got.append((i, ind["indices"], wave))
    torch.cuda.synchronize()
    for i, idx, wave in got:
        assert torch.equal(idx, ref[i][0]) and torch.equal(wave, ref[i][1]), f"clip set {i}"


def test_graph_capture_in_a_cold_process():
    """reserve -> capture with NO eager call before it, in a fresh process (in this one every kernel has long been configured):
    the replayed graph returns the tokens and the waveform of the eager path."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "capture_cold.py"
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "captured without warm-up; tokens equal True wave equal True" in r.stdout, r.stdout[-2000:]
