"""HIP kernels vs the CPU oracle, block by block, through the C ABI (needs an MI355X: `pytest -m gpu`)."""
import time

import numpy as np
from pathlib import Path
import pytest
import torch

import l3ac_amd
from l3ac_amd import _capi, weights as W
from oracle import l3ac_oracle as O
from tests import gpu_ops as G
from tests.helpers import GOLDEN, index_mismatch_report, load_case, seeded_audio

pytestmark = pytest.mark.gpu

ATOL_BLOCK = 2e-5   # one block, activations O(1), fp32 with a different summation order
RTOL_BLOCK = 2e-5


def _close(name, got, ref, atol=ATOL_BLOCK, rtol=RTOL_BLOCK):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    print(f"[{name}] max|err|={err.max().item():.3e} max|ref|={ref.abs().max().item():.3e}")
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    assert (err <= bound).all(), f"{name}: max err {err.max().item():.3e} (worst excess {(err - bound).max().item():.3e})"


@pytest.fixture(scope="module")
def tiny():
    codec = l3ac_amd.get_model(GOLDEN / "tiny.toml", synthetic_seed=3)
    codec.network.to(device="cuda").eval()
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    return codec, mc, w


@pytest.fixture(scope="module")
def full():
    codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    codec.network.to(device="cuda").eval()
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    return codec, mc, w


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------------
def test_gemm_against_fp64():
    for (m, n, k) in [(300, 24, 96), (257, 96, 24), (1000, 130, 344), (129, 512, 2048), (64, 48, 144)]:
        a, w, b = _rand((m, k), 1), _rand((n, k), 2, 0.1), _rand((n,), 3)
        got = G.gemm(a.cuda(), w.cuda(), b.cuda()).cpu()
        ref = (a.double() @ w.double().T + b.double())
        scale = (a.abs().double() @ w.abs().double().T).max().item()
        err = (got.double() - ref).abs().max().item()
        print(f"[gemm {m}x{n}x{k}] max err {err:.3e} (sum|a||w| {scale:.3e})")
        assert err <= 4e-7 * scale + 1e-6


def test_gemm_split_accuracy():
    """bf16x3 split GEMM (gemm_split.hip) against fp64: its error must stay within the fp32-MFMA kernel's own error
    budget and, in rms, not exceed the exact-fp32 kernel's error on the same operands."""
    torch.manual_seed(0)
    # (grids of <= 256 blocks of 128 x 128 run the few-blocks variant of the kernel, larger ones the full-grid kernel: both are
    # covered, each with a ragged N and a K tail)
    for (m, n, k, heavy) in [(1000, 192, 32, False), (257, 256, 1024, False), (129, 512, 2048, True), (1000, 200, 344, False),
                             (384, 704, 128, True), (4096, 1024, 256, False), (20000, 200, 344, False), (17000, 512, 256, True)]:
        a, w, b = _rand((m, k), 1), _rand((n, k), 2, 0.1), _rand((n,), 3)
        if heavy:  # wide dynamic range inside one dot product
            a = a * torch.exp(3.0 * _rand((m, k), 4))
        ref = a.double() @ w.double().T + b.double()
        mag = a.abs().double() @ w.abs().double().T + b.abs().double()
        got_s = G.gemm_split(a.cuda(), w.cuda(), b.cuda()).cpu()
        got_f = G.gemm(a.cuda(), w.cuda(), b.cuda()).cpu()
        es = ((got_s.double() - ref).abs() / mag)
        ef = ((got_f.double() - ref).abs() / mag)
        print(f"[gemm_split {m}x{n}x{k}] err/sum|a.w|: split max {es.max():.3e} rms {es.pow(2).mean().sqrt():.3e} | "
              f"fp32 mfma max {ef.max():.3e} rms {ef.pow(2).mean().sqrt():.3e}")
        assert es.max().item() <= max(4e-7, 1.05 * ef.max().item())
        assert es.pow(2).mean().sqrt().item() <= 1.05 * ef.pow(2).mean().sqrt().item() + 1e-9


def test_gemm_split_forms_return_the_same_bits():
    """The bf16x3 GEMM has four launch forms by grid size and shape — 128 x 128 blocks, 64 x 128 blocks with W two k tiles ahead (few
    blocks), and for a single clip's rows 32 x 64 blocks whose operands stream through an LDS-DMA ring (round 5: whole k tiles, at
    least eight of them; seven or eight in flight by the parity of their number) or 64 x 32 column slices with W and A four k tiles
    ahead in registers (the rest) — and a clip's results must not depend on the batch it arrives in: the same rows through a large
    call and through calls of 180 / 60 / 1 rows, bit for bit (also k % 32 != 0, n not a multiple of 128 or 32, odd and even numbers of
    k tiles, exactly eight k tiles)."""
    g = torch.Generator().manual_seed(41)
    for n, k in ((512, 2048), (2048, 512), (256, 512), (192, 96), (200, 40), (128 * 3, 1032), (192, 288), (320, 352), (200, 256), (256, 544)):
        m_big = 128 * 2 * (256 // max(1, n // 128) + 1)   # more 128 x 128 blocks than the chip has CUs
        a = torch.randn(m_big, k, generator=g)
        w = torch.randn(n, k, generator=g) * 0.1
        b = torch.randn(n, generator=g)
        big = G.gemm_split(a.cuda(), w.cuda(), b.cuda()).cpu()
        for rows in (180, 60, 1, 400):
            small = G.gemm_split(a[:rows].contiguous().cuda(), w.cuda(), b.cuda()).cpu()
            assert torch.equal(small, big[:rows]), f"n={n} k={k}: {rows} rows alone differ from the same rows of a {m_big}-row call"


def test_gemm_split_w256_returns_the_same_bits():
    """Round 6: gemm_split_kernel_w256 (one wave per SIMD, 192 x 256 per workgroup) takes the long-K light-epilogue batch products by default
    (L3AC_GEMM_W256=1) and every eligible shape with =2 — among them the snake + GRN epilogue of the C = 512 stage's first product, which the
    default never sends there.  The switch is read once per process: one subprocess per value, and every digest (ragged row counts, both
    weight shapes, a 256-column weight, the whole C = 512 ConvUnit) must equal the old kernel's (=0)."""
    import json
    import os
    import subprocess
    import sys
    outs = {}
    for mode in ("0", "1", "2"):
        env = dict(os.environ, L3AC_GEMM_W256=mode)
        r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "w256_digest.py")], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"[gemm_split_w256] digests: {outs['0']}")
    assert outs["1"] == outs["0"] and outs["2"] == outs["0"], {k: (outs["0"][k], outs["1"][k], outs["2"][k]) for k in outs["0"] if len({outs[m][k] for m in outs}) > 1}


def test_gemm_f32_forms_return_the_same_bits():
    """The exact-fp32 GEMM has two instruction forms — v_mfma_f32_32x32x2_f32 on 128-row blocks, and for a single clip's rows
    v_mfma_f32_16x16x4_f32 on 64-row blocks with the k tiles four ahead (round 5) — that must return the same bits: both are per output
    the k-ordered chain of fused multiply-adds, fed the same k sequence (tools/probes/mfma_f32_order_probe.hip).  The same rows through a
    large call and through calls of a single clip's size, for the shapes of the streaming chunk's down convs (k % 32 != 0 among them),
    with mixed magnitudes so that every rounding step shows; and against the fp64 product."""
    g = torch.Generator().manual_seed(43)
    for n, k in ((128, 576), (48, 144), (96, 240), (128, 384), (32, 36), (100, 72)):
        m_big = 128 * (1100 // max(1, -(-n // 32)))   # more blocks than a quarter of the chip's CUs: the large-grid form
        a = torch.randn(m_big, k, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (m_big, k), generator=g).float())
        w = torch.randn(n, k, generator=g) * 0.1
        b = torch.randn(n, generator=g)
        big = G.gemm(a.cuda(), w.cuda(), b.cuda()).cpu()
        for rows in (180, 60, 1, 540, 65):
            small = G.gemm(a[:rows].contiguous().cuda(), w.cuda(), b.cuda()).cpu()
            assert torch.equal(small, big[:rows]), f"n={n} k={k}: {rows} rows alone differ from the same rows of a {m_big}-row call"
        ref = a[:540].double() @ w.double().T + b.double()
        mag = a[:540].abs().double() @ w.abs().double().T + b.abs().double()
        assert float(((big[:540].double() - ref).abs() / mag).max()) < 2.0 ** -20


def test_gemm_split_cancellation_and_underflow():
    """Adversarial operands for the bf16x3 split GEMM, with an ABSOLUTE bound per output: |err| <= 2^-20 * sum|a.w|.  (2^-23 was
    asked for; measured on these K = 1024 shapes the k-ordered fp32 MFMA chain itself — the kernel the split route replaces —
    reaches 0.99 x 2^-22 on the cancellation case and 2.3 x 2^-22 on the mixed-magnitude one, the split kernel 0.71 x and
    2.0 x: both ratios are printed, and the split kernel must ALSO stay within 1.05 x of the exact kernel's worst case), plus — only where operands sit within 2^16 of the bottom of the fp32 range, so that their lower planes
    leave the bf16 normal range — K * 2^-126 * max|w|.
      * cancellation: every dot product is ~0 while its terms are O(1): pairs (k, k + K/2) cancel to the last bit or two;
      * mixed magnitudes: a few O(1e4) terms over a floor of O(1e-4) ones;
      * small operands: a ~ 1e-30 (all three planes still normal bf16 numbers: exact) and a ~ 1e-37 (x1 / x2 planes are
        bf16 subnormals or zero)."""
    g = torch.Generator().manual_seed(21)
    cases = []
    m, n, k = 512, 256, 1024
    w = torch.randn(n, k, generator=g) * 0.1
    a = torch.randn(m, k, generator=g)
    # cancellation is per (row, column) pair, so build it on the weights: w[:, k + K/2] = -w[:, k] and a[:, k + K/2] ~ a[:, k]
    wc = w.clone()
    wc[:, k // 2:] = -wc[:, :k // 2]
    ac = a.clone()
    ac[:, k // 2:] = ac[:, :k // 2] * (1 + 2.0 ** -20 * torch.randn(m, k // 2, generator=g))
    cases.append(("cancellation", ac, wc, 0.0))
    am = torch.randn(m, k, generator=g) * 1e-4
    am[:, ::97] = torch.randn(m, len(range(0, k, 97)), generator=g) * 1e4
    cases.append(("mixed magnitudes", am, w, 0.0))
    cases.append(("a ~ 1e-30", a * 1e-30, w, 0.0))
    cases.append(("a ~ 1e-37", a * 1e-37, w, k * 2.0 ** -126 * float(w.abs().max())))
    cases.append(("w ~ 1e-30", a, w * 1e-30, 0.0))
    for name, aa, ww, floor in cases:
        ref = aa.double() @ ww.double().T
        mag = aa.abs().double() @ ww.abs().double().T
        got = G.gemm_split(aa.cuda(), ww.cuda(), None).cpu().double()
        got_f = G.gemm(aa.cuda(), ww.cuda(), None).cpu().double()
        err, err_f = (got - ref).abs(), (got_f - ref).abs()
        bound = 2.0 ** -20 * mag + floor
        worst, worst_f = float((err / bound.clamp_min(1e-300)).max()), float((err_f / bound.clamp_min(1e-300)).max())
        print(f"[gemm_split {name}] max err/bound {worst:.3f} (fp32 mfma: {worst_f:.3f}); max|ref|/max mag {float(ref.abs().max() / mag.max()):.2e}")
        assert torch.isfinite(got).all()
        assert (err <= bound).all(), name
        assert worst <= 1.05 * worst_f + 0.05, name


def test_sin_squared_range():
    """The kernels' own sin^2 (device_math.hpp) against fp64: |err| <= 2e-7 for |u| <= 1e5 in both the scalar and the
    packed form; beyond that the argument is clamped (documented guard): the value stays in [0, 1] and snake(x) is within
    1e-5 relative of the exact activation; NaN stays NaN, no integer overflow garbage."""
    g = torch.Generator().manual_seed(31)
    u = torch.cat([torch.linspace(-40, 40, 20000), torch.logspace(-6, 5, 40000), -torch.logspace(-6, 5, 40000),
                   (torch.rand(100000, generator=g) * 2 - 1) * 1e5, torch.tensor([0.0, 1e5, -1e5, 99999.99, 3.14159274, 1.57079637])])
    u = u[: u.numel() // 4 * 4].reshape(-1, 4).float().contiguous()
    alpha = torch.ones(4).cuda()
    ref = torch.sin(u.double()).pow(2)
    for mode in (2, 3):
        got = G.snake(u.cuda(), alpha, mode).cpu().double()
        err = (got - ref).abs().max().item()
        print(f"[sin^2 mode {mode}] max |err| vs fp64 over |u| <= 1e5: {err:.3e}")
        assert err <= 2e-7
    # beyond the range: bounded, finite, and harmless for the activation
    big = torch.tensor([1.0001e5, 3e5, 1e7, 2.6e7, 1e9, 3e9, 1e20, 3.4e38, -1e6, -5e9, -3.4e38, 123456.789]).reshape(-1, 4).float()
    for mode in (2, 3):
        got = G.snake(big.cuda(), alpha, mode).cpu()
        assert torch.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
    a = torch.tensor([0.5, 1.0, 2.7, 10.0])
    big = big.clamp(-1e30, 1e30)  # keep alpha * x finite
    for mode in (0, 1):
        got = G.snake(big.cuda(), a.cuda(), mode).cpu().double()
        exact = big.double() + (a + 1e-8).reciprocal().double() * torch.sin((a * big).double()).pow(2)
        rel = ((got - exact).abs() / exact.abs()).max().item()
        print(f"[snake mode {mode}] max relative error beyond the guard: {rel:.3e}")
        assert rel <= 1e-5
    nan = torch.tensor([float("nan"), float("inf"), -float("inf"), 1.0]).reshape(1, 4)
    for mode in (0, 1):
        got = G.snake(nan.cuda(), alpha, mode).cpu()
        assert torch.isnan(got[0, 0]) and torch.isinf(got[0, 1]) and torch.isinf(got[0, 2]) and torch.isfinite(got[0, 3])
    # ordinary activations: snake itself against the oracle's formula, both forms
    x = _rand((4096, 64), 77, 3.0)
    al = (torch.rand(64, generator=g) * 3 + 0.05)
    ref = O.snake(x, al)
    for mode in (0, 1):
        _close(f"snake mode {mode}", G.snake(x.cuda(), al.cuda(), mode).cpu(), ref, atol=1e-6, rtol=1e-6)


def test_gelu_known_answers():
    """The kernels' exact GELU (device_math.hpp gelu_erf: 0.5 x (1 + erf(x / sqrt 2)) with the branch-free fp32 erf_f32; the encoder
    stem evaluates 80 per sample, the transformer's GEGLU 341 per token) against fp64 on 2^20 points — a dense sweep of [-8, 8], normal
    draws of the stem's argument scale, the fits' switch-over point and its one-ulp neighbours, large and tiny arguments — and against
    torch.nn.functional.gelu in fp32 on the same points.  GELU's 1 + erf cancels for negative arguments, so the bound is on the TERMS:
    |err| <= 0.5 |x| * 1.2e-7 (erf to 2 ulp of 1) + 1.5 ulp of the result; and the worst error may not exceed twice torch's own fp32 worst."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(77)
    sw = 0.927734375 * 2 ** 0.5
    edge = torch.tensor([sw, -sw, 0.0, -0.0, 1e-30, -1e-30, 1e-6, 40.0, -40.0, 1e20, -1e20], dtype=torch.float64).float()
    edge = torch.cat([edge, torch.nextafter(edge, torch.full_like(edge, 9e9)), torch.nextafter(edge, torch.full_like(edge, -9e9))])
    n = (1 << 20) - edge.numel()
    x = torch.cat([torch.linspace(-8, 8, n // 2), torch.randn(n - n // 2, generator=g) * 1.5, edge]).float().reshape(-1, 4).contiguous()
    alpha = torch.ones(4).cuda()
    got = G.snake(x.cuda(), alpha, 4).cpu().double()
    ref = F.gelu(x.double())
    tor = F.gelu(x).double()
    ulp = torch.from_numpy(np.spacing(np.abs(ref.float().numpy()))).double()
    e_got, e_tor = (got - ref).abs(), (tor - ref).abs()
    print(f"[gelu] kernels: max |err| {float(e_got.max()):.3e} = {float((e_got / (ulp + 1e-38)).max()):.2f} ulp; torch fp32: "
          f"max |err| {float(e_tor.max()):.3e} = {float((e_tor / (ulp + 1e-38)).max()):.2f} ulp ({x.numel()} points)")
    assert torch.isfinite(got).all()
    bound = 0.5 * x.double().abs().clamp(max=10.0) * 1.2e-7 + 1.5 * ulp + 1e-38
    worst = float((e_got / bound).max())
    print(f"[gelu] worst err / bound {worst:.3f}")
    assert (e_got <= bound).all()
    assert float(e_got.max()) <= 2.0 * float(e_tor.max())


def test_first_block(tiny, full):
    for codec, mc, w in (tiny, full):
        x = seeded_audio(2, 1000)
        ref = O.first_block(w, "encoder.blocks.0", x.unsqueeze(1))
        got = G.op_plain(codec.network.context(), "l3ac_op_first_block", x.cuda(), 2, 1000, (2, 1000, mc.encoder_dims[0]))
        _close("first_block", G.from_frames(got), ref)


def test_conv_units(tiny, full):
    codec, mc, w = tiny
    for block, c in (("encoder.blocks.1.0.module", 8), ("encoder.blocks.5.1.module", 24), ("decoder.blocks.1.1.module", 32)):
        x = _rand((2, c, 70), 10 + c)
        ref = O.conv_unit(w, block, x)
        got = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (2, 70, c))
        _close(block, G.from_frames(got), ref)
    codec, mc, w = full
    for block, c, t in (("encoder.blocks.1.0.module", 24, 400), ("encoder.blocks.7.1.module", 192, 60),
                        ("decoder.blocks.1.2.module", 512, 40), ("decoder.blocks.4.0.module", 256, 50),
                        ("decoder.blocks.10.0.module", 48, 300), ("encoder.blocks.5.0.module", 96, 130),
                        ("decoder.blocks.7.1.module", 96, 77), ("encoder.blocks.3.0.module", 48, 31),
                        ("encoder.blocks.1.0.module", 24, 1)):
        x = _rand((2, c, t), 20 + c)
        ref = O.conv_unit(w, block, x)
        got = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (2, t, c))
        _close(block, G.from_frames(got), ref, atol=5e-5, rtol=5e-5)


def test_conv_units_narrow_ring_and_split_forms(full):
    """The two fused forms of the narrow ConvUnits (C = 24 / 48; C = 96 runs on conv_unit_wide_kernel<96>: test_conv_units_wide_fused) —
    conv_unit_ring_kernel (16 frames per wave, weights resident in LDS; default at C = 48) and conv_unit_split_kernel (32 frames per
    wave; default at C = 24) — against the oracle on shapes that exercise tile and clip boundaries (frames = 1, 15, not a multiple of
    16, enough tiles for several passes of the persistent grid), against each other through an fp64 evaluation (neither may be the
    less accurate one by more than rounding noise), and the ring kernel's small-grid geometry (four waves per workgroup: a single
    clip) against its large-grid one bit for bit."""
    codec, mc, w = full
    ctx = codec.network.context()
    cases = (("encoder.blocks.1.0.module", 24, 3, 1), ("encoder.blocks.1.0.module", 24, 2, 47), ("encoder.blocks.1.0.module", 24, 40, 16200),
             ("encoder.blocks.3.0.module", 48, 5, 15), ("decoder.blocks.10.0.module", 48, 3, 8100), ("decoder.blocks.10.0.module", 48, 300, 97))
    for block, c, b, t in cases:
        x = _rand((b, c, t), 900 + c + t)
        ref = O.conv_unit(w, block, x[:2])
        outs = {}
        for name, ring in (("ring", 2), ("split", 0)):
            ctx.set_option("narrow_ring", ring)
            try:
                outs[name] = G.from_frames(G.op_block(ctx, "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, c)))
                if name == "ring" and b > 1:  # a clip alone (small grid: four-wave workgroups at C = 48) == the same clip inside the batch
                    alone = G.from_frames(G.op_block(ctx, "l3ac_op_conv_unit", block, G.to_frames(x[:1]), (1, t, c)))
                    assert torch.equal(alone, outs[name][:1]), f"ring kernel: clip alone differs from the batch's on {block} B={b} T={t}"
            finally:
                ctx.set_option("narrow_ring", 1)
            _close(f"{name} {block} B={b} T={t}", outs[name][:2], ref, atol=5e-5, rtol=5e-5)
        d = float((outs["ring"] - outs["split"]).abs().max())
        print(f"[ring vs split {block} B={b} T={t}] max difference {d:.3e}")
        assert d < 5e-5
    block, c = "decoder.blocks.10.0.module", 48
    x = _rand((2, c, 2000), 4242)
    ref64 = O.conv_unit({k: v.double() for k, v in w.items() if k.startswith(block)}, block, x.double())
    errs = {}
    for name, ring in (("ring", 2), ("split", 0)):
        ctx.set_option("narrow_ring", ring)
        try:
            got = G.from_frames(G.op_block(ctx, "l3ac_op_conv_unit", block, G.to_frames(x), (2, 2000, c))).double()
        finally:
            ctx.set_option("narrow_ring", 1)
        e = (got - ref64).abs()
        errs[name] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    print(f"[C = 48 forms vs fp64] ring max {errs['ring'][0]:.3e} rms {errs['ring'][1]:.3e} | split max {errs['split'][0]:.3e} rms {errs['split'][1]:.3e}")
    assert errs["ring"][1] <= 1.5 * errs["split"][1] + 1e-9 and errs["split"][1] <= 1.5 * errs["ring"][1] + 1e-9


def test_conv_units_wide_fused(full):
    """conv_unit_wide_kernel (C = 96 / 192 / 256: hidden tensor in registers, weights streamed through the LDS ring) against the
    oracle on shapes that exercise what the small cases above do not: clip boundaries inside a 32-row tile (frames % 32 != 0),
    a ragged last tile, one frame per clip, and enough rows for several passes per workgroup (the weight ring wraps around
    and is re-entered: 36 000 rows > 256 workgroups x 128 rows)."""
    codec, mc, w = full
    for block, c, b, t in (("decoder.blocks.4.1.module", 256, 3, 900), ("encoder.blocks.7.0.module", 192, 5, 180),
                           ("decoder.blocks.4.2.module", 256, 7, 1), ("encoder.blocks.7.1.module", 192, 2, 33),
                           ("decoder.blocks.4.0.module", 256, 40, 900), ("encoder.blocks.7.1.module", 192, 200, 180),
                           # C = 96 (round 4): ring slots of three pieces copied by three of the four waves
                           ("decoder.blocks.7.0.module", 96, 3, 2700), ("encoder.blocks.5.0.module", 96, 5, 534),
                           ("decoder.blocks.7.1.module", 96, 7, 1), ("encoder.blocks.5.0.module", 96, 2, 33),
                           ("decoder.blocks.7.1.module", 96, 16, 2700)):
        x = _rand((b, c, t), 200 + c + t)
        ref = O.conv_unit(w, block, x)
        got = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, c))
        _close(f"{block} B={b} T={t}", G.from_frames(got), ref, atol=5e-5, rtol=5e-5)
    # the two tile forms of the kernel — 32 frames per wave, and 16 frames per wave for grids that leave most of the chip idle (a
    # single clip: the streaming chunk) — return the same bits: the first clips of a large batch against the same clips run alone
    for block, c, t, b_big in (("decoder.blocks.4.1.module", 256, 900, 40), ("encoder.blocks.7.0.module", 192, 180, 200),
                               ("decoder.blocks.4.0.module", 256, 901, 40), ("encoder.blocks.7.1.module", 192, 37, 300),
                               ("decoder.blocks.7.0.module", 96, 2700, 14), ("encoder.blocks.5.0.module", 96, 535, 70)):
        x = _rand((b_big, c, t), 400 + c + t)
        xf = G.to_frames(x)
        big = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, xf, (b_big, t, c))
        for b_small in (1, 3):
            small = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, xf[:b_small].contiguous(), (b_small, t, c))
            assert torch.equal(small, big[:b_small]), f"{block} T={t}: {b_small} clip(s) alone differ from the same clips of a batch of {b_big}"
    # C = 128 (no shipped model has such a stage; the reference's default geometry does): 2 ring slots per product
    codec128 = l3ac_amd.get_model(GOLDEN / "refdefault.toml", synthetic_seed=5)
    codec128.network.to(device="cuda").eval()
    w128 = W.folded_weights(codec128.network.state_dicts())
    for block, b, t in (("decoder.blocks.4.0.module", 3, 1000), ("decoder.blocks.4.1.module", 40, 1000), ("decoder.blocks.4.1.module", 2, 31)):
        x = _rand((b, 128, t), 300 + t)
        ref = O.conv_unit(w128, block, x)
        got = G.op_block(codec128.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, 128))
        _close(f"refdefault {block} B={b} T={t}", G.from_frames(got), ref, atol=5e-5, rtol=5e-5)
    # the same unit through the unfused route (dwconv+LN, two split GEMMs): the fused kernel may not be the less accurate one
    x = _rand((4, 256, 450), 999)
    block = "decoder.blocks.4.1.module"
    ref64 = O.conv_unit({k: v.double() for k, v in w.items() if k.startswith(block)}, block, x.double())
    fused = G.from_frames(G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (4, 450, 256))).double()
    xin = G.to_frames(x)
    y = xin.clone()  # x == y: the in-place call takes the unfused route
    from l3ac_amd import _capi
    _capi.check(codec.network.context().lib.l3ac_op_conv_unit(codec.network.context().handle, block.encode(), y.data_ptr(), 4, 450,
                                                                y.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    unfused = G.from_frames(y).double()
    e_f, e_u = (fused - ref64).abs(), (unfused - ref64).abs()
    print(f"[wide fused vs unfused, fp64 reference] fused max {float(e_f.max()):.3e} rms {float(e_f.pow(2).mean().sqrt()):.3e} | "
          f"unfused max {float(e_u.max()):.3e} rms {float(e_u.pow(2).mean().sqrt()):.3e}")
    assert float(e_f.pow(2).mean().sqrt()) <= 1.5 * float(e_u.pow(2).mean().sqrt()) + 1e-9
    assert float(e_f.max()) <= 2.0 * float(e_u.max()) + 1e-7


def test_conv_units_wide_sliced_form_returns_the_same_bits(full):
    """The SLICED form of the wide ConvUnits (round 5: two launches over frame tiles x channel slices for few frames — the streaming
    chunk) against the fused kernel on the same input: every output bit equal, for whole and ragged tiles, odd tile counts (an idle
    wave in the last workgroup of either launch), one frame, several clips (the front end's clip boundaries), at C = 256 / 192 and
    at C = 128 (the reference's default geometry); then against the oracle."""
    codec, mc, w = full
    codec128 = l3ac_amd.get_model(GOLDEN / "refdefault.toml", synthetic_seed=5)
    codec128.network.to(device="cuda").eval()
    w128 = W.folded_weights(codec128.network.state_dicts())
    cases = [(codec, w, "decoder.blocks.4.1.module", 256, 1, 900), (codec, w, "decoder.blocks.4.0.module", 256, 1, 901),
             (codec, w, "decoder.blocks.4.2.module", 256, 3, 333), (codec, w, "decoder.blocks.4.2.module", 256, 1, 1),
             (codec, w, "decoder.blocks.4.0.module", 256, 4, 1024), (codec, w, "encoder.blocks.7.0.module", 192, 1, 178),
             (codec, w, "encoder.blocks.7.1.module", 192, 5, 37), (codec, w, "encoder.blocks.7.1.module", 192, 2, 16),
             (codec, w, "encoder.blocks.7.0.module", 192, 22, 178), (codec128, w128, "decoder.blocks.4.0.module", 128, 1, 1000),
             (codec128, w128, "decoder.blocks.4.1.module", 128, 3, 47),
             # C = 96: the front end (dw-conv + LayerNorm + split) runs inside the first launch, as in the fused kernel's pass prologue
             (codec, w, "decoder.blocks.7.0.module", 96, 1, 2700), (codec, w, "decoder.blocks.7.1.module", 96, 1, 2701),
             (codec, w, "encoder.blocks.5.0.module", 96, 1, 534), (codec, w, "encoder.blocks.5.0.module", 96, 5, 37),
             (codec, w, "decoder.blocks.7.1.module", 96, 3, 1), (codec, w, "decoder.blocks.7.0.module", 96, 7, 585)]
    for cdc, ww, block, c, b, t in cases:
        ctx = cdc.network.context()
        x = _rand((b, c, t), 4000 + c + t)
        xf = G.to_frames(x)
        outs = {}
        for mode in (0, 2):
            ctx.set_option("wide_sliced", mode)
            try:
                with _capi.profile() as prof:
                    outs[mode] = G.op_block(ctx, "l3ac_op_conv_unit", block, xf, (b, t, c))
            finally:
                ctx.set_option("wide_sliced", 1)
            names = [e["name"] for e in prof.entries]
            assert any(n.startswith("wide_sliced_out_kernel") for n in names) == (mode == 2), names
            assert any(n.startswith("conv_unit_wide_kernel") for n in names) == (mode == 0), names
        assert torch.equal(outs[0], outs[2]), f"{block} C={c} B={b} T={t}: the sliced form differs from the fused kernel"
        _close(f"sliced {block} B={b} T={t}", G.from_frames(outs[2]), O.conv_unit(ww, block, x), atol=5e-5, rtol=5e-5)
    # by default a single clip takes the sliced form, a large batch the fused one
    ctx = codec.network.context()
    for b, want in ((1, "wide_sliced_out_kernel<256>"), (40, "conv_unit_wide_kernel<256>")):
        with _capi.profile() as prof:
            G.op_block(ctx, "l3ac_op_conv_unit", "decoder.blocks.4.1.module", G.to_frames(_rand((b, 256, 900), 7)), (b, 900, 256))
        assert want in [e["name"] for e in prof.entries]


def test_conv_unit_wide_unit_counter_returns_the_same_bits(full):
    """Round 6: at two workgroups per CU (C = 96) the batch form hands its units — groups of four 32-frame tiles, then groups of four half
    tiles — out by a device counter (option "unit_counter", default 1) instead of equal static shares.  Which workgroup computes a tile must
    not show: the same bits as the static form, on consecutive launches (the last workgroup to leave zeroes the counters for the next
    launch), for a ragged last tile, and for a row count just above the grid's first pass."""
    codec, mc, w = full
    ctx = codec.network.context()
    for block, b, t in (("decoder.blocks.7.0.module", 31, 2700), ("decoder.blocks.7.1.module", 64, 2699), ("decoder.blocks.7.0.module", 25, 2700)):
        xf = G.to_frames(_rand((b, 96, t), 9100 + b))
        outs = {}
        for mode in (1, 0, 1, 1):
            ctx.set_option("unit_counter", mode)
            try:
                with _capi.profile() as prof:
                    y = G.op_block(ctx, "l3ac_op_conv_unit", block, xf, (b, t, 96))
            finally:
                ctx.set_option("unit_counter", 1)
            assert "conv_unit_wide_kernel<96>" in [e["name"] for e in prof.entries]
            if mode in outs:
                assert torch.equal(outs[mode], y), f"{block} B={b} T={t}: two launches with unit_counter={mode} differ"
            outs[mode] = y
        assert torch.equal(outs[0], outs[1]), f"{block} B={b} T={t}: units by counter differ from static shares"
    _close("unit counter, C = 96", G.from_frames(outs[1]), O.conv_unit(w, "decoder.blocks.7.0.module", G.from_frames(xf)), atol=5e-5, rtol=5e-5)


def test_conv_unit_wide_sliced_tail_returns_the_same_bits(full):
    """Round 6: at C >= 128 a remainder of at most 256 frame tiles behind the full passes runs in the sliced form (two launches from the same
    plane image) instead of as half tiles inside the fused kernel.  40 x 900 frames at C = 256 are one full pass + 101 tiles: both kernels
    must appear, and every clip — those of the passes, those of the tail, the one the boundary falls into — must equal the clip alone."""
    codec, mc, w = full
    ctx = codec.network.context()
    block, c, b, t = "decoder.blocks.4.1.module", 256, 40, 900
    xf = G.to_frames(_rand((b, c, t), 9300))
    with _capi.profile() as prof:
        y = G.op_block(ctx, "l3ac_op_conv_unit", block, xf, (b, t, c))
    names = [e["name"] for e in prof.entries]
    assert "conv_unit_wide_kernel<256>" in names and "wide_sliced_out_kernel<256>" in names and "wide_sliced_hidden_kernel<256>" in names, names
    for i in (0, 35, 36, 37, 39):  # rows 32 768 .. fall into clip 36
        y1 = G.op_block(ctx, "l3ac_op_conv_unit", block, xf[i:i + 1].contiguous(), (1, t, c))
        assert torch.equal(y1, y[i:i + 1]), f"clip {i}: differs from the clip alone"
    _close("sliced tail, C = 256", G.from_frames(y), O.conv_unit(w, block, G.from_frames(xf)), atol=5e-5, rtol=5e-5)


def test_conv_units_wide_scratch_on_a_fresh_context():
    """The wide ConvUnit's front end writes bf16x3 planes of WHOLE 32-frame tiles (conv_unit_wide_scratch_bytes) into the
    hidden scratch: more than the 4C floats per row that scratch is otherwise sized by when batch * frames < 12.  On a
    context that has never run anything larger (round-2 advisor finding: the write ran past the buffer) a few-row unit and a
    one-token clip of the reference-default geometry (C = 128 / 256 stages, hop 45) must work and match the oracle."""
    for cfg, seed, block, c in (("1kbps", 0, "decoder.blocks.4.0.module", 256), ("1kbps", 0, "encoder.blocks.7.0.module", 192),
                                (GOLDEN / "refdefault.toml", 5, "decoder.blocks.4.0.module", 128)):
        for b, t in ((1, 3), (2, 4), (1, 11)):
            codec = l3ac_amd.get_model(cfg, synthetic_seed=seed)  # fresh context every time: its workspace starts empty
            codec.network.to(device="cuda").eval()
            w = W.folded_weights(codec.network.state_dicts())
            x = _rand((b, c, t), 700 + c + t)
            ref = O.conv_unit(w, block, x)
            got = G.op_block(codec.network.context(), "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, c))
            _close(f"fresh ctx {block} B={b} T={t}", G.from_frames(got), ref, atol=5e-5, rtol=5e-5)
    codec = l3ac_amd.get_model(GOLDEN / "refdefault.toml", synthetic_seed=5)
    codec.network.to(device="cuda").eval()
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    audio = seeded_audio(1, 2 * mc.hop_length)  # two tokens (with ONE the reference itself raises: InstanceNorm over a single frame)
    q, ind = codec.encode_audio(audio.cuda())
    wave = codec.decode_audio(q)
    torch.cuda.synchronize()
    _, ind_ref = O.encode_audio(w, mc, audio)
    assert torch.equal(ind["indices"].cpu(), ind_ref["indices"])
    wave_ref = O.decode_audio(w, mc, indices=ind_ref["indices"])
    assert float((wave.cpu() - wave_ref).abs().max()) < 2e-3


def test_down_and_k3_layers(tiny, full):
    import torch.nn.functional as F
    for (codec, mc, w), cases in ((tiny, [("encoder.blocks.2", 8, 16, 2, 66), ("encoder.blocks.4", 16, 24, 3, 66)]),
                                  (full, [("encoder.blocks.2", 24, 48, 6, 600), ("encoder.blocks.6", 96, 192, 3, 90),
                                          # the DOWN form of up_fused_kernel (conv + ChannelNorm in one kernel): both widths it takes at
                                          # 1kbps, output lengths around its 16-frame tiles (1, 15, 16, 17, 33) and whole clips
                                          ("encoder.blocks.4", 48, 96, 5, 5 * 33), ("encoder.blocks.4", 48, 96, 5, 2700),
                                          ("encoder.blocks.2", 24, 48, 6, 6), ("encoder.blocks.2", 24, 48, 6, 6 * 15),
                                          ("encoder.blocks.2", 24, 48, 6, 6 * 16), ("encoder.blocks.2", 24, 48, 6, 6 * 17),
                                          ("encoder.blocks.2", 24, 48, 6, 16200)])):
        for block, ci, co, s, t in cases:
            x = _rand((2, ci, t), 30 + ci + t)
            ref = F.conv1d(x, w[f"{block}.0.weight"], w[f"{block}.0.bias"], stride=s)
            ref = O.channel_norm_first(ref, w[f"{block}.1.weight"], w[f"{block}.1.bias"])
            ctx = codec.network.context()
            # option "down_fused": 0 = the GEMM + row kernel, 1 = the bf16x3 one-kernel form, 2 (default, round 6) = down_exact_kernel, whose
            # results must be those of 0 BIT FOR BIT (the exact fp32 MFMA chain in gemm_f32_kernel's k order, row_kernel's ChannelNorm tree)
            outs = {}
            for fused in (0, 1, 2):
                ctx.set_option("down_fused", fused)
                try:
                    got = G.op_block(ctx, "l3ac_op_down_layer", block, G.to_frames(x), (2, t // s, co))
                finally:
                    ctx.set_option("down_fused", 2)
                _close(f"{block} T={t} down_fused={fused}", G.from_frames(got), ref)
                outs[fused] = got.cpu()
            assert torch.equal(outs[2], outs[0]), f"{block} T={t}: the exact one-kernel form differs from the GEMM + row kernel route"
    for (codec, mc, w), cases in ((tiny, [("encoder.blocks.6", 24, 16, 50), ("decoder.blocks.0", 16, 32, 50)]),
                                  (full, [("encoder.blocks.8", 192, 128, 180), ("decoder.blocks.0", 128, 512, 180)])):
        for block, ci, co, t in cases:
            x = _rand((3, ci, t), 40 + ci)
            ref = F.conv1d(x, w[f"{block}.weight"], w[f"{block}.bias"], padding=1)
            got = G.op_block(codec.network.context(), "l3ac_op_conv_k3", block, G.to_frames(x), (3, t, co))
            _close(block, G.from_frames(got), ref)


def test_enhance_and_up_layers(tiny, full):
    import torch.nn.functional as F
    for (codec, mc, w), cases in ((tiny, [("decoder.blocks.2", "decoder.blocks.3", 32, 16, 3, 41)]),
                                  (full, [("decoder.blocks.2", "decoder.blocks.3", 512, 256, 5, 180),
                                          ("decoder.blocks.11", "decoder.blocks.12", 48, 24, 2, 700),
                                          # up_fused_kernel (gate + conv + upsample + ChannelNorm in one kernel): every width it takes,
                                          # lengths around its 14-frame tiles (1, 13, 14, 15, 29 frames) and a long clip
                                          ("decoder.blocks.5", "decoder.blocks.6", 256, 96, 3, 900), ("decoder.blocks.8", "decoder.blocks.9", 96, 48, 3, 2700),
                                          ("decoder.blocks.5", "decoder.blocks.6", 256, 96, 3, 29), ("decoder.blocks.8", "decoder.blocks.9", 96, 48, 3, 15),
                                          ("decoder.blocks.11", "decoder.blocks.12", 48, 24, 2, 14), ("decoder.blocks.11", "decoder.blocks.12", 48, 24, 2, 13),
                                          ("decoder.blocks.8", "decoder.blocks.9", 96, 48, 3, 2)])):
        for eb, ub, ci, co, s, t in cases:
            x = _rand((2, ci, t), 50 + ci)
            ref = O.enhance_block(w, eb, x)
            got = G.op_block(codec.network.context(), "l3ac_op_enhance", eb, G.to_frames(x), (2, t, ci))
            _close(eb, G.from_frames(got), ref)
            r = F.conv1d(x, w[f"{ub}.0.weight"], w[f"{ub}.0.bias"])
            r = F.interpolate(r, scale_factor=s, mode="linear", align_corners=False)
            r = O.channel_norm_first(r, w[f"{ub}.2.weight"], w[f"{ub}.2.bias"])
            got = G.op_block(codec.network.context(), "l3ac_op_up_layer", ub, G.to_frames(x), (2, t * s, co))
            _close(ub, G.from_frames(got), r)
            # the pipeline's fused pair: gate folded into the up conv's A operand
            r = F.conv1d(ref, w[f"{ub}.0.weight"], w[f"{ub}.0.bias"])
            r = F.interpolate(r, scale_factor=s, mode="linear", align_corners=False)
            r = O.channel_norm_first(r, w[f"{ub}.2.weight"], w[f"{ub}.2.bias"])
            xin = G.to_frames(x)
            keep = xin.clone()
            got = G.op_block2(codec.network.context(), "l3ac_op_enhance_up", eb, ub, xin, (2, t * s, co))
            _close(eb + "+" + ub, G.from_frames(got), r)
            assert torch.equal(xin, keep), f"{eb}+{ub}: l3ac_op_enhance_up wrote to its input (x is only read)"
            with pytest.raises(_capi.L3acError, match="alias"):  # in place is refused, not silently another rounding
                G.op_block2(codec.network.context(), "l3ac_op_enhance_up", eb, ub, xin, (2, t * s, co), out=xin)


def test_last_block(tiny, full):
    for (codec, mc, w), blk, t in ((tiny, "decoder.blocks.7.block", 300), (full, "decoder.blocks.13.block", 1500)):
        c = mc.decoder_dims[-1]
        x = _rand((2, c, t), 60)
        r = x
        for u, d in enumerate((1, 3, 9)):
            r = O.legacy_unit(w, f"{blk}.0.{u}.module", r, d)
        r = O.snake(r, w[f"{blk}.1.alpha"])
        r = torch.tanh(torch.nn.functional.conv1d(r, w[f"{blk}.2.weight"], w[f"{blk}.2.bias"], padding=3)).squeeze(1)
        got = G.op_plain(codec.network.context(), "l3ac_op_last_block", G.to_frames(x), 2, t, (2, t))
        _close("last_block", got.cpu(), r, atol=5e-5, rtol=5e-5)


def test_last_block_tile_counter_returns_the_same_bits(full):
    """Round 6: the LegacyUnits' persistent workgroups take their tiles from a device counter (option "unit_counter") once there are more
    than two per workgroup — the same bits as static shares, launch after launch (three units per call, each leaving the counters zeroed),
    with a ragged last tile per clip."""
    codec, mc, w = full
    ctx = codec.network.context()
    c = mc.decoder_dims[-1]
    for b, t in ((20, 16200), (33, 9001)):
        xf = G.to_frames(_rand((b, c, t), 61 + b))
        outs = {}
        for mode in (1, 0, 1):
            ctx.set_option("unit_counter", mode)
            try:
                y = G.op_plain(ctx, "l3ac_op_last_block", xf, b, t, (b, t))
            finally:
                ctx.set_option("unit_counter", 1)
            if mode in outs:
                assert torch.equal(outs[mode], y), f"B={b} T={t}: two calls with unit_counter={mode} differ"
            outs[mode] = y
        assert torch.equal(outs[0], outs[1]), f"B={b} T={t}: tiles by counter differ from static shares"


def test_local_trans_single_and_multi_window(tiny, full):
    codec, mc, w = tiny
    for block, window, depth, t in (("en_encoder.down_trans.trans", 16, 1, 42), ("en_decoder.local_trans", 8, 2, 21),
                                    ("en_decoder.up_trans.trans", 16, 2, 70), ("en_encoder.local_trans", 8, 2, 5)):
        x = _rand((3, t, mc.feature_dim), 70 + t)
        ref = O.local_trans(w, block, x, window, depth)
        got = G.op_block(codec.network.context(), "l3ac_op_local_trans", block, x.cuda(), (3, t, mc.feature_dim))
        _close(f"{block} T={t} W={window}", got.cpu(), ref, atol=5e-5, rtol=5e-5)
    codec, mc, w = full
    for block, window, depth, t in (("en_encoder.down_trans.trans", 750, 1, 180), ("en_decoder.local_trans", 250, 3, 60),
                                    ("en_decoder.local_trans", 250, 3, 600)):
        x = _rand((2, t, 128), 80 + t)
        ref = O.local_trans(w, block, x, window, depth)
        got = G.op_block(codec.network.context(), "l3ac_op_local_trans", block, x.cuda(), (2, t, 128))
        _close(f"{block} T={t} W={window}", got.cpu(), ref, atol=1e-4, rtol=1e-4)


def test_local_trans_stack_kernel(full):
    """trans_stack_kernel (one launch per LocalTrans stack, one workgroup per clip: LayerNorm, qkv, causal attention + distance
    bias, out projection, GEGLU FeedForward, all on bf16x3 MFMAs with the clip resident on the CU) against the oracle for every
    stack of the 1kbps and 3kbps models: full-length clips, lengths that are not multiples of 16 / 32 (padding frames inside the
    last wave's tile, an odd number of key tiles), a single frame, and a batch larger than the chip holds workgroups at once.  The
    fp64 evaluation of the same stack bounds the error: the fused kernel may not be less accurate than the unfused route."""
    codec, mc, w = full
    ctx = codec.network.context()
    cases = [("en_encoder.down_trans.trans", 750, 1, 180, 3), ("en_encoder.local_trans", 250, 2, 60, 3),
             ("en_decoder.local_trans", 250, 3, 60, 2), ("en_decoder.up_trans.trans", 750, 2, 180, 2),
             ("en_decoder.up_trans.trans", 750, 2, 177, 2), ("en_decoder.local_trans", 250, 3, 45, 2),
             ("en_decoder.local_trans", 250, 3, 1, 2), ("en_decoder.up_trans.trans", 750, 2, 33, 1),
             ("en_decoder.up_trans.trans", 750, 2, 192, 1), ("en_encoder.local_trans", 250, 2, 17, 300)]
    for block, window, depth, t, bsz in cases:
        x = _rand((bsz, t, 128), 500 + t + bsz)
        ref = O.local_trans(w, block, x[:4], window, depth)
        got = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (bsz, t, 128)).cpu()
        _close(f"stack {block} T={t} B={bsz}", got[:4], ref, atol=1e-4, rtol=1e-4)
        if bsz > 4:  # clips are independent: every clip of the big batch equals the same clip run alone
            alone = G.op_block(ctx, "l3ac_op_local_trans", block, x[[0, 255, 299]].cuda(), (3, t, 128)).cpu()
            assert torch.equal(alone, got[[0, 255, 299]])
    codec3 = l3ac_amd.get_model("3kbps", synthetic_seed=0)
    codec3.network.to(device="cuda").eval()
    w3 = W.folded_weights(codec3.network.state_dicts())
    for block, depth, t in (("en_encoder.local_trans", 1, 167), ("en_decoder.local_trans", 3, 167), ("en_decoder.local_trans", 3, 100)):
        x = _rand((2, t, 128), 600 + t)
        ref = O.local_trans(w3, block, x, 400, depth)
        got = G.op_block(codec3.network.context(), "l3ac_op_local_trans", block, x.cuda(), (2, t, 128)).cpu()
        _close(f"3kbps stack {block} T={t}", got, ref, atol=1e-4, rtol=1e-4)
    # accuracy against fp64, fused vs unfused (the exact-fp32 route of the same context runs the unfused kernels)
    block, window, depth, t = "en_decoder.up_trans.trans", 750, 2, 180
    x = _rand((4, t, 128), 777)
    ref64 = O.local_trans({k: v.double() for k, v in w.items() if k.startswith(block)}, block, x.double(), window, depth)
    fused = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (4, t, 128)).cpu().double()
    ctx.set_gemm_split(False)
    try:
        unfused = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (4, t, 128)).cpu().double()
    finally:
        ctx.set_gemm_split(True)
    e_f, e_u = (fused - ref64).abs(), (unfused - ref64).abs()
    print(f"[trans stack vs fp64] fused max {float(e_f.max()):.3e} rms {float(e_f.pow(2).mean().sqrt()):.3e} | "
          f"exact-fp32 unfused max {float(e_u.max()):.3e} rms {float(e_u.pow(2).mean().sqrt()):.3e}")
    assert float(e_f.pow(2).mean().sqrt()) <= 2.0 * float(e_u.pow(2).mean().sqrt()) + 1e-9


def test_local_trans_stack_cooperative_form(full):
    """The cooperative form of trans_stack_kernel (batches of at most 32 clips — a streaming chunk is one: six co-resident workgroups
    per clip, one head / one pair of FeedForward chunks each, partial tiles exchanged through write-through slabs behind an arrival
    counter and added in a fixed order) must return the SAME BITS as the one-workgroup form, which sums its heads and chunk pairs in
    that order too: checked through the context option that switches the form off, for every wave-count instantiation (<= 64, <=
    128, <= 192 frames), ragged lengths, 1 .. 32 clips, a single frame; against clips of a large batch (batch invariance across the
    two forms); and launch after launch on one context (the arrival counters are left zeroed by every launch)."""
    codec, mc, w = full
    ctx = codec.network.context()
    cases = [("en_encoder.down_trans.trans", 180, 1), ("en_decoder.up_trans.trans", 180, 3), ("en_decoder.up_trans.trans", 177, 8),
             ("en_decoder.local_trans", 60, 1), ("en_encoder.local_trans", 60, 5), ("en_decoder.local_trans", 45, 2),
             ("en_decoder.up_trans.trans", 100, 1), ("en_decoder.up_trans.trans", 128, 4), ("en_decoder.local_trans", 1, 2),
             ("en_decoder.up_trans.trans", 192, 2), ("en_decoder.local_trans", 17, 7), ("en_decoder.up_trans.trans", 180, 32),
             ("en_decoder.local_trans", 60, 21)]
    for block, t, bsz in cases:
        x = _rand((bsz, t, 128), 900 + t + bsz)
        coop = [G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (bsz, t, 128)).cpu() for _ in range(3)]
        ctx.set_option("trans_coop", 0)
        try:
            single = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (bsz, t, 128)).cpu()
        finally:
            ctx.set_option("trans_coop", 1)
        assert torch.isfinite(single).all()
        for i, c in enumerate(coop):
            assert torch.equal(c, single), f"{block} T={t} B={bsz}: cooperative launch {i} differs from the one-workgroup form"
    # a clip alone (cooperative) == the same clip inside a batch the cooperative form does not take (one workgroup per clip)
    for block, t in (("en_decoder.up_trans.trans", 180), ("en_decoder.local_trans", 60)):
        x = _rand((40, t, 128), 950 + t)
        big = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (40, t, 128)).cpu()
        for i in (0, 17, 39):
            alone = G.op_block(ctx, "l3ac_op_local_trans", block, x[i:i + 1].cuda(), (1, t, 128)).cpu()
            assert torch.equal(alone[0], big[i])
    # many launches back to back, other work in between, then under uneven load: a long kernel on another stream while the
    # cooperative workgroups exchange their partials
    block, t = "en_decoder.up_trans.trans", 180
    x = _rand((2, t, 128), 999)
    want = G.op_block(ctx, "l3ac_op_local_trans", block, x.cuda(), (2, t, 128)).cpu()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(64 << 20, device="cuda")
    xg = x.cuda()
    for i in range(300):  # every word of every launch is compared: a stale partial shows as a different bit pattern
        with torch.cuda.stream(side):
            if i % 3 == 0:
                (a @ a).sum()        # matrix-core load on the other CUs
            elif i % 3 == 1:
                big.mul_(1.0001)     # a 512 MB memory stream through every L2
        got = G.op_block(ctx, "l3ac_op_local_trans", block, xg, (2, t, 128))
        assert torch.equal(got.cpu(), want), f"launch {i}"
    torch.cuda.synchronize()
    assert ctx.coop_timeout_count() == 0  # no arrival poll of any launch above ran into its time limit


def _stack_launch_names(fn):
    """Names of the trans_stack launches `fn` makes (the library's own launch profile): '<coop>' marks the cooperative form."""
    with _capi.profile() as prof:
        fn()
    return [e["name"] for e in prof.entries if e["name"].startswith("trans_stack_kernel")]


def test_cooperative_stack_reports_a_lost_arrival():
    """A cooperative launch whose six workgroups per clip are not all there (here: a test hook makes workgroup 3 of every clip
    withhold its first arrival, and the time limit is 5 ms) must not return wrong tokens silently: the kernel counts the expired
    polls in host-visible memory, every workgroup stops waiting (the launch ends at once), and the host hears it — through
    encode_audio(validate=True), through l3ac_coop_timeout_count / _pending, and (without either) as L3AC_ECOOP from a LATER call (the
    next one entered after the failing launch has run: here the stream is drained in between).  The context
    then runs the one-workgroup form and the repeated call returns the right tokens."""
    ref_codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    ref_codec.network.to(device="cuda").eval()
    ref_codec.network.context().set_option("trans_coop", 0)
    audio = ((torch.rand(2, 16000, generator=torch.Generator().manual_seed(4)) * 2 - 1) * 0.5).cuda()
    want = ref_codec.encode_audio(audio)[1]["indices"]
    want_wave = ref_codec.decode_audio(indices=want)

    def fresh():
        c = l3ac_amd.get_model("1kbps", synthetic_seed=0)
        c.network.to(device="cuda").eval()
        k = c.network.context()
        k.set_option("coop_timeout_ms", 5)
        k.set_option("coop_test_fault", 3 + 1)
        return c, k

    # (1) validate=True: the call itself raises, the message says what happened, the retry is right
    codec, ctx = fresh()
    assert any("<coop>" in n for n in _stack_launch_names(lambda: codec.encode_audio(audio[:1])))  # (the hook only bites cooperative launches)
    assert ctx.coop_timeout_count(reset=True) > 0
    ctx.set_option("trans_coop", 1)  # (acting on the count switched the form off: back on for the checks below)
    t0 = time.perf_counter()
    with pytest.raises(_capi.L3acError, match="(?s)cooperative.*time limit.*invalid"):
        codec.encode_audio(audio, validate=True)
    assert time.perf_counter() - t0 < 5.0  # every workgroup gives up after ONE expired poll: milliseconds, not one limit per phase
    got = codec.encode_audio(audio, validate=True)[1]["indices"]  # one-workgroup form now
    assert torch.equal(got, want)
    assert not any("<coop>" in n for n in _stack_launch_names(lambda: codec.encode_audio(audio)))
    # (2) no validate: the NEXT call returns L3AC_ECOOP once, then the context works
    codec, ctx = fresh()
    bad = codec.encode_audio(audio)[1]["indices"]
    torch.cuda.synchronize()
    with pytest.raises(_capi.L3acError, match="(?s)error -5.*earlier call.*INVALID"):
        codec.decode_audio(indices=bad)
    got = codec.encode_audio(audio)[1]["indices"]
    assert torch.equal(got, want)
    assert torch.equal(codec.decode_audio(indices=got), want_wave)
    assert ctx.coop_timeout_count() > 0   # cumulative since the last reset; asking again does not raise
    assert ctx.coop_timeout_count(reset=True) > 0 and ctx.coop_timeout_count() == 0
    # (3) decode_audio(validate=True) the same way
    codec, ctx = fresh()
    with pytest.raises(_capi.L3acError, match="decode_audio.*cooperative"):
        codec.decode_audio(indices=want, validate=True)
    assert torch.equal(codec.decode_audio(indices=want, validate=True), want_wave)
    # (4) ADVICE r5: an UNVALIDATED failing call followed by a validated one.  The validated call must not take the earlier failure
    # into its own baseline (round 5 did: count-before acknowledged it, count-after minus count-before was 0 and the invalid tokens
    # were decoded without a word): it raises before running anything, names the earlier call, and the context is good afterwards
    codec, ctx = fresh()
    bad = codec.encode_audio(audio)[1]["indices"]            # loses its cooperative launch; nobody asked
    assert ctx.coop_timeout_pending() > 0                    # (synchronises; reading it changes nothing:)
    assert ctx.coop_timeout_pending() > 0
    with pytest.raises(_capi.L3acError, match="(?s)EARLIER call.*invalid.*Nothing was run"):
        codec.decode_audio(indices=bad, validate=True)
    assert ctx.coop_timeout_pending() == 0                   # delivered by that exception: no second report
    got = codec.encode_audio(audio, validate=True)[1]["indices"]
    assert torch.equal(got, want)
    assert torch.equal(codec.decode_audio(indices=got, validate=True), want_wave)
    # the unsynchronised entry check bounds its report: "one of the last N call(s)"
    codec, ctx = fresh()
    codec.encode_audio(audio)
    torch.cuda.synchronize()
    with pytest.raises(_capi.L3acError, match=r"(?s)one of the last 1 call\(s\)"):
        codec.encode_audio(audio)


def test_cooperative_stacks_of_two_contexts_cannot_starve_each_other():
    """Two contexts x 32 clips on two streams: 2 x 192 cooperative workgroups do not fit 256 CUs, and workgroups that wait for
    partners which cannot be scheduled would wait for ever.  Contexts claim their CUs in a per-device registry
    (l3ac_coop_claimed_cus): the context whose claim no longer fits runs the one-workgroup form (same bits).  Small batches of both
    contexts fit together and both stay cooperative.  (Contexts of other tests that are still alive hold claims too: the batch is
    sized from what is free.)"""
    import gc
    gc.collect()  # contexts of earlier tests return their claims when they are destroyed
    lib = _capi.load_library()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    base = lib.l3ac_coop_claimed_cus(0)
    n = min(32, (cus - 6 - base) // 6)  # clips per batch: context a's claim fits, context b's then does not
    if n < 2 or 12 * n <= cus - base:
        pytest.skip(f"{base} of {cus} CUs claimed by other live contexts: no batch size separates the two cases")
    mk = lambda: l3ac_amd.get_model("1kbps", synthetic_seed=0)
    a, b = mk(), mk()
    for c in (a, b):
        c.network.to(device="cuda").eval()
    ca, cb = a.network.context(), b.network.context()
    g = torch.Generator().manual_seed(8)
    audio = ((torch.rand(n, 16000, generator=g) * 2 - 1) * 0.5).cuda()
    # one clip each: both cooperative
    assert any("<coop>" in x for x in _stack_launch_names(lambda: a.encode_audio(audio[:1])))
    assert any("<coop>" in x for x in _stack_launch_names(lambda: b.encode_audio(audio[:1])))
    assert lib.l3ac_coop_claimed_cus(0) == base + 12
    # n clips each: the first to ask gets its 6 n CUs, the second does not
    names_a = _stack_launch_names(lambda: a.encode_audio(audio))
    names_b = _stack_launch_names(lambda: b.encode_audio(audio))
    assert all("<coop>" in x for x in names_a) and not any("<coop>" in x for x in names_b), (names_a, names_b)
    assert lib.l3ac_coop_claimed_cus(0) == base + 6 * n + 6
    ref = mk()
    ref.network.to(device="cuda").eval()
    ref.network.context().set_option("trans_coop", 0)
    want = ref.encode_audio(audio)[1]["indices"]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(6):  # side by side, repeatedly
        with torch.cuda.stream(sa):
            ia = a.encode_audio(audio)[1]["indices"]
        with torch.cuda.stream(sb):
            ib = b.encode_audio(audio)[1]["indices"]
        outs.append((ia, ib))
    torch.cuda.synchronize()
    for ia, ib in outs:
        assert torch.equal(ia, want) and torch.equal(ib, want)
    assert ca.coop_timeout_count() == 0 and cb.coop_timeout_count() == 0
    # a claim only grows — unless the context gives it back (option coop_release_claim): the registry shrinks by exactly that claim
    held = lib.l3ac_coop_claimed_cus(0)
    ca.set_option("coop_release_claim", 1)
    assert lib.l3ac_coop_claimed_cus(0) < held
    # the claim goes back with the context
    del a, ca, names_a, outs, ia
    gc.collect()
    assert lib.l3ac_coop_claimed_cus(0) == base + 6
    assert all("<coop>" in x for x in _stack_launch_names(lambda: b.encode_audio(audio)))  # ... and b's launches fit now


# ---------------------------------------------------------------------------------------------------
def test_fsq_known_answers_and_exactness():
    kat = np.load(GOLDEN / "fsq_kat.npz")
    for tag, feat in (("l7", 128), ("l9977", 128), ("even", 16), ("tiny", 16)):
        levels = kat[f"{tag}_levels"].tolist()
        d = len(levels)
        g = torch.Generator().manual_seed(5)
        w_out = (torch.rand(feat, d, generator=g) - 0.5).cuda()
        b_out = (torch.rand(feat, generator=g) - 0.5).cuda()
        z = torch.from_numpy(kat[f"{tag}_z"])
        q, idx, li, _ = G.fsq_forward(None, levels, None, None, w_out, b_out, latents_in=z.cuda())
        n_bad, ok = index_mismatch_report(idx.cpu().numpy(), kat[f"{tag}_indices"], z.numpy(), levels, tau=1e-6)
        print(f"[fsq {tag}] {n_bad} of {z.shape[0]} indices differ from the reference (1-ulp tanh boundary cases)")
        assert ok and n_bad == 0  # the reference's own known-answer vectors: bit for bit (observed: 0 on every set; round 5 allowed 1)
        np.testing.assert_array_equal(li.cpu().numpy(), kat[f"{tag}_level_indices"])
        # decode path == reference indices_to_codes followed by project_out
        sel = torch.from_numpy(kat[f"{tag}_dec_idx"])
        codes = torch.from_numpy(kat[f"{tag}_dec_codes"])
        ref = codes @ w_out.cpu().T + b_out.cpu()
        got = G.fsq_decode(sel.cuda(), levels, w_out, b_out)
        _close(f"fsq_decode {tag}", got.cpu(), ref, atol=1e-6, rtol=1e-6)
    # rounding boundaries themselves, with no transcendental in between (tests/golden/fsq_boundary_kat.npz, reference answers from
    # SuperFSQ.quantize_act_value): products exactly k + 0.5 and their +-1 / +-2 ulp neighbours must agree BIT FOR BIT
    bk = np.load(GOLDEN / "fsq_boundary_kat.npz")
    for tag, feat in (("l7", 128), ("l9977", 128), ("even", 16), ("tiny", 16)):
        levels = bk[f"{tag}_levels"].tolist()
        d = len(levels)
        g = torch.Generator().manual_seed(6)
        w_out = (torch.rand(feat, d, generator=g) - 0.5).cuda()
        b_out = (torch.rand(feat, generator=g) - 0.5).cuda()
        act = torch.from_numpy(bk[f"{tag}_act"])
        q, idx, li = G.fsq_quantize_act(act.cuda(), levels, w_out, b_out)
        np.testing.assert_array_equal(li.cpu().numpy(), bk[f"{tag}_level_indices"])
        np.testing.assert_array_equal(idx.cpu().numpy(), bk[f"{tag}_indices"])
        ref_q = torch.from_numpy(bk[f"{tag}_q"]) @ w_out.cpu().T + b_out.cpu()
        _close(f"fsq boundary q_feature {tag}", q.cpu(), ref_q, atol=1e-6, rtol=1e-6)
    # exact ties: tanh(0) = 0 -> 0.5, 1.5, 2.5, 3.5 round half-to-even to 0, 2, 2, 4 (torch.round semantics)
    w_out = torch.zeros(8, 4).cuda()
    b_out = torch.zeros(8).cuda()
    _, idx, li, _ = G.fsq_forward(None, [2, 4, 6, 8], None, None, w_out, b_out, latents_in=torch.zeros(3, 4).cuda())
    assert li.cpu().tolist() == [[0.0, 2.0, 2.0, 4.0]] * 3
    assert idx.cpu().tolist() == [0 + 2 * 2 + 2 * 8 + 4 * 48] * 3


def test_fsq_fused_forward_and_roundtrip(full):
    codec, mc, w = full
    for tag in ("1kbps", "3kbps"):
        mc2, w2, _, _ = load_case(tag)
        x = _rand((4096, 128), 90, 2.0)
        q_ref, ind_ref, lat_ref = O.quantizer(w2, mc2, x)
        dev = {k: w2[f"quantizer.{k}"].cuda() for k in ("project_in.weight", "project_in.bias", "project_out.weight", "project_out.bias")}
        q, idx, li, lat = G.fsq_forward(x.cuda(), list(mc2.levels), dev["project_in.weight"], dev["project_in.bias"],
                                        dev["project_out.weight"], dev["project_out.bias"], want_latents=True)
        _close(f"latents {tag}", lat.cpu(), lat_ref, atol=2e-6, rtol=2e-6)
        n_bad, ok = index_mismatch_report(idx.cpu().numpy(), ind_ref["indices"].numpy(), lat_ref.numpy(), mc2.levels, tau=1e-4)
        print(f"[fsq fused {tag}] {n_bad}/4096 tokens differ (project_in summation order)")
        assert ok and n_bad <= 1  # observed on the MI355X: 0 (gate = observed + 1)
        # the hot form of the forward pass (fsq_forward128_kernel: no latents tap, one latent per lane end to end, q_d broadcast instead of
        # six divisions per lane) returns the generic kernel's bits — also on a token count that is not a multiple of its 32-token groups
        # and with saturating / tiny inputs
        for xs in (x, torch.cat([x[:1001] * 40.0, x[:37] * 1e-6])):
            qh, ih, lh, _ = G.fsq_forward(xs.cuda(), list(mc2.levels), dev["project_in.weight"], dev["project_in.bias"],
                                          dev["project_out.weight"], dev["project_out.bias"])
            qg, ig, lg_, _ = G.fsq_forward(xs.cuda(), list(mc2.levels), dev["project_in.weight"], dev["project_in.bias"],
                                           dev["project_out.weight"], dev["project_out.bias"], want_latents=True)
            assert torch.equal(ih, ig) and torch.equal(lh, lg_) and torch.equal(qh, qg), f"{tag}: hot and generic quantiser kernels differ"
        # from the SAME latents the indices are bit-exact
        q2, idx2, li2, _ = G.fsq_forward(None, list(mc2.levels), None, None, dev["project_out.weight"], dev["project_out.bias"],
                                         latents_in=lat_ref.cuda())
        n_bad2, ok2 = index_mismatch_report(idx2.cpu().numpy(), ind_ref["indices"].numpy(), lat_ref.numpy(), mc2.levels, tau=1e-6)
        assert ok2 and n_bad2 <= 1
        # to_features(indices) == q_feature, bit for bit (reference property)
        back = G.fsq_decode(idx, list(mc2.levels), dev["project_out.weight"], dev["project_out.bias"])
        assert torch.equal(back, q)
        # level indices consistent with indices
        lv = torch.tensor(mc2.levels)
        basis = torch.cumprod(torch.tensor([1] + list(mc2.levels)[:-1]), 0)
        assert torch.equal((li.cpu().long() * basis).sum(-1).int(), idx.cpu())
        assert (li.cpu() >= 0).all() and (li.cpu() <= (lv - 1)).all()


def test_vq_argmin_matches_closed_form():
    """The explicit-codebook search reproduces the closed-form FSQ indices, near-ties excluded."""
    for levels in ([7] * 6, [9, 9, 9, 7, 7, 7], [5, 3, 4]):
        g = torch.Generator().manual_seed(11)
        z = torch.randn(3000, len(levels), generator=g) * 1.2
        _, idx_ref, _ = O.fsq_quantize(z, levels)
        cb = O.codebook(levels)
        got = G.vq_argmin(torch.tanh(z).cuda(), cb.cuda()).cpu()
        lv = torch.tensor(levels, dtype=torch.float32)
        scaled = (torch.tanh(z) + 1) / 2 * (lv - 1)
        margin = ((scaled - scaled.floor()) - 0.5).abs().min(dim=1).values
        clear = margin > 1e-4
        assert clear.float().mean() > 0.99
        assert torch.equal(got[clear], idx_ref[clear]), f"levels={levels}"
        # brute force agrees with a CPU brute force everywhere except exact fp32 distance ties
        d = torch.cdist(torch.tanh(z).double(), cb.double())
        best = d.argmin(1)
        gap = d.gather(1, got.long().unsqueeze(1)).squeeze(1) - d.gather(1, best.unsqueeze(1)).squeeze(1)
        assert (gap.abs() < 1e-6).all()


def test_vq_argmin_scan_form_and_ties():
    """The two many-query forms (N >= 5120) against the wavefront form on the same queries, on a codebook with exact duplicates:
    the direct-form scan (per-block minima, winner recovered afterwards) and the screened form (matrix-core scores choose a block of
    16 candidates or send the query to the full direct-form search) must both return the LOWEST index among bit-identical
    distances, across block, tile, slice and ragged-tail boundaries, for every dimension's template."""
    g = torch.Generator().manual_seed(5)
    for k, d in ((4999, 6), (1030, 3), (777, 1), (2050, 8), (640, 5)):
        base = torch.randn(k, d, generator=g)
        cb = base.clone()
        dup = torch.randperm(k, generator=g)[: k // 3]
        src = torch.randint(0, k, (dup.numel(),), generator=g)
        cb[dup] = cb[src]                                  # exact duplicates at arbitrary positions
        cb[-1] = cb[0]                                     # ... including the very last code of the ragged tail
        q = torch.randn(20000, d, generator=g)
        q[:2000] = cb[torch.randint(0, k, (2000,), generator=g)]  # queries ON codes: zero distance, many ties
        q[2000:2100] = 0.5 * (cb[:100] + cb[100:200])      # queries midway between two codes: near-ties of the scores
        info = {}
        screen = G.vq_argmin(q.cuda(), cb.cuda(), info=info).cpu()
        scan = G.vq_argmin(q.cuda(), cb.cuda(), form=1).cpu()
        wave = torch.cat([G.vq_argmin(q[i:i + 5000].cuda(), cb.cuda()).cpu() for i in range(0, 20000, 5000)])
        assert torch.equal(scan, wave)
        assert torch.equal(screen, wave), f"k={k} d={d}: {(screen != wave).sum().item()} differ"
        assert 0 < info["listed"] < 20000                  # duplicates went to the full search, clear winners did not
        # lowest index among identical rows: the winner is the first occurrence of its row
        rows = {}
        for i, r in enumerate(cb.numpy().tobytes()[j * 4 * d:(j + 1) * 4 * d] for j in range(k)):
            rows.setdefault(r, i)
        first = torch.tensor([rows[cb[i].numpy().tobytes()] for i in scan.tolist()])
        assert torch.equal(first, scan.long())
        dist = torch.cdist(q.double(), cb.double())
        gap = dist.gather(1, scan.long().unsqueeze(1)).squeeze(1) - dist.min(1).values
        assert (gap.abs() < 1e-5).all()


def test_vq_argmin_screened_form_fsq_grids():
    """Screened form == direct-form scan, every query, on the two FSQ codebooks at the headline batch's query counts — including
    queries placed exactly on and one ulp around the half-way planes between grid points, where the matrix-core scores of two
    codes agree to the last bits — and the share of queries that needed the full direct-form search stays small."""
    for levels, n in (([7] * 6, 15360), ([9, 9, 9, 7, 7, 7], 42752)):
        g = torch.Generator().manual_seed(23)
        cb = O.codebook(levels)
        q = torch.tanh(torch.randn(n, len(levels), generator=g) * 1.2)
        lv = torch.tensor(levels, dtype=torch.float32)
        half = (torch.randint(0, 6, (3000, len(levels)), generator=g).float() + 0.5) / (lv - 1) * 2 - 1  # half-way planes
        pick = torch.rand(3000, len(levels), generator=g) < 0.3
        q[:3000] = torch.where(pick, half, q[:3000])
        q[3000:4000] = torch.nextafter(q[:1000], torch.ones(()))
        q[4000:5000] = torch.nextafter(q[:1000], -torch.ones(()))
        info = {}
        screen = G.vq_argmin(q.cuda(), cb.cuda(), info=info).cpu()
        scan = G.vq_argmin(q.cuda(), cb.cuda(), form=1).cpu()
        assert torch.equal(screen, scan), f"levels={levels}: {(screen != scan).sum().item()} differ"
        assert info["listed"] < 5000 + 0.03 * n, info
    # non-finite and huge queries take the full search and return what the scan returns
    q = torch.randn(6000, 6)
    q[5] = float("nan"); q[6, 2] = float("inf"); q[7] = 1e30; q[8] = -3e38
    cb = O.codebook([7] * 6)
    assert torch.equal(G.vq_argmin(q.cuda(), cb.cuda()).cpu(), G.vq_argmin(q.cuda(), cb.cuda(), form=1).cpu())


def test_vq_argmin_graph_capture():
    """include/l3ac_hip.h: l3ac_vq_argmin allocates nothing and can be captured into a hipGraph — for both forms (the screened
    form's listed-query pass sizes itself from a device counter): replays on new queries in the same buffers equal eager calls."""
    from l3ac_amd import _capi
    lib = _capi.load_library()
    cb = O.codebook([7] * 6).cuda()
    k = cb.shape[0]
    for n in (60, 6000):
        g = torch.Generator().manual_seed(n)
        sets = [torch.tanh(torch.randn(n, 6, generator=g) * 1.2).cuda() for _ in range(3)]
        sets[1][: n // 3] = cb[torch.randint(0, k, (n // 3,), generator=g).cuda()]  # queries on codes: many go to the full search
        eager = [G.vq_argmin(q, cb) for q in sets]
        q_static = sets[0].clone()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        nbytes = lib.l3ac_vq_argmin_scratch_bytes(n, k, 0)
        scratch = torch.zeros(max(nbytes, 4), dtype=torch.uint8, device="cuda")
        call = lambda: _capi.check(lib.l3ac_vq_argmin(q_static.data_ptr(), n, cb.data_ptr(), k, 6, out.data_ptr(), scratch.data_ptr(),
                                                      nbytes, 0, torch.cuda.current_stream().cuda_stream))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            call()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            call()
        for i in (1, 2, 0):
            q_static.copy_(sets[i])
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager[i]), f"n={n} set {i}"


def test_context_options_by_name(full):
    """l3ac_ctx_set_option: the route switches by name belong to ONE context, unknown names are refused (L3AC_EINVAL with a
    message naming the options) — among them the options retired in round 5 — and out-of-range values of `narrow_ring` are clamped."""
    codec, mc, w = full
    ctx = codec.network.context()
    for name in ("no_such_option", "wide_narrow", "ring_geometry"):
        with pytest.raises(RuntimeError, match="unknown option"):
            ctx.set_option(name, 1)
    other = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    other.network.cuda().eval()
    octx = other.network.context()
    try:
        ctx.set_option("gemm_split", 0)
        assert ctx.get_gemm_split() is False and octx.get_gemm_split() is True  # another context is untouched
    finally:
        ctx.set_option("gemm_split", 1)
    block, c, b, t = "decoder.blocks.10.0.module", 48, 3, 333
    x = _rand((b, c, t), 77)
    ctx.set_option("narrow_ring", 2)
    try:
        ref = G.from_frames(G.op_block(ctx, "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, c)))
        ctx.set_option("narrow_ring", 99)     # clamped to 2: still the ring kernel at this width
        got = G.from_frames(G.op_block(ctx, "l3ac_op_conv_unit", block, G.to_frames(x), (b, t, c)))
    finally:
        ctx.set_option("narrow_ring", 1)
    assert torch.equal(got, ref)
