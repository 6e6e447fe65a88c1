"""CPU-side tests: config / weight plumbing, the drop-in surface's error behaviour, and that the C-ABI library
loads and exports every symbol include/l3ac_hip.h declares (no compute without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

import l3ac_amd
from l3ac_amd import _capi, weights as W
from l3ac_amd.config import L3ACConfig, ModelConfig, resolve_config_file
from tests.helpers import GOLDEN

REPO = Path(__file__).resolve().parent.parent


def test_list_models_and_geometry():
    assert l3ac_amd.list_models() == ["0k75bps", "1k5bps", "1kbps", "3kbps"]
    want = {"0k75bps": (360, 117649, 748.6), "1kbps": (270, 117649, 998.2), "1k5bps": (180, 117649, 1497.3),
            "3kbps": (96, 250047, 2988.6)}  # reference README.md:73-76
    for name, (hop, k, bps) in want.items():
        cfg = L3ACConfig(config_file=resolve_config_file(name))
        mc = cfg.network_config
        assert cfg.sample_rate == 16000 and cfg.model_tag == f"{name}.v1"
        assert mc.hop_length == hop and mc.codebook_size == k
        info = l3ac_amd.get_model_info(l3ac_amd.L3AC(cfg))
        assert abs(info["bps"] - bps) < 0.06
        assert cfg.weight_url.startswith("https://huggingface.co/zhai-lw/L3AC/resolve/main/weights/" + name)
        assert cfg.model_path == Path.home() / ".cache" / "l3ac" / f"{name}.v1"


def test_config_validation():
    with pytest.raises(Exception):
        ModelConfig(encoder_dims=(1, 2), encoder_depths=(1,), compress_rates=(2,))
    with pytest.raises(Exception):
        ModelConfig(bogus_key=1)
    with pytest.raises(NotImplementedError):
        ModelConfig(decoder_last_layer="dilation", en_coder_dynamic_pos=True).check_supported()
    with pytest.raises(FileNotFoundError):
        resolve_config_file("no_such_model")


def test_parameter_counts_match_survey():
    mc = L3ACConfig(config_file=resolve_config_file("1kbps")).network_config
    count = lambda m: sum(int(torch.Size(s).numel()) for _, s in W.raw_keys(mc, m))
    assert count("encoder") == 869316 and count("quantizer") == 1670 and count("decoder") == 8494082
    assert len(W.raw_keys(mc, "encoder")) == 109 and len(W.raw_keys(mc, "decoder")) == 241


def test_synthetic_weights_are_deterministic_and_fold():
    mc = L3ACConfig(config_file=GOLDEN / "tiny.toml").network_config
    a = W.synthetic_state_dicts(mc, seed=3)
    b = W.synthetic_state_dicts(mc, seed=3)
    c = W.synthetic_state_dicts(mc, seed=4)
    for m in W.MODULE_NAMES:
        for k in a[m]:
            assert torch.equal(a[m][k], b[m][k])
    assert not torch.equal(a["encoder"]["blocks.0.conv_1.bias"], c["encoder"]["blocks.0.conv_1.bias"])
    W.check_state_dicts(a, mc)
    folded = W.folded_weights(a)
    g = a["encoder"]["blocks.0.conv_1.parametrizations.weight.original0"]
    v = a["encoder"]["blocks.0.conv_1.parametrizations.weight.original1"]
    w = folded["encoder.blocks.0.conv_1.weight"]
    assert torch.allclose(w, g * v / v.reshape(80, -1).norm(dim=1).reshape(80, 1, 1), atol=1e-7)
    assert torch.allclose(w.reshape(80, -1).norm(dim=1), g.reshape(-1), rtol=1e-5)
    assert not any("parametrizations" in k for k in folded)
    bad = {m: dict(sd) for m, sd in a.items()}
    del bad["decoder"]["blocks.0.bias"]
    with pytest.raises(KeyError):
        W.check_state_dicts(bad, mc)


def test_weight_files_roundtrip(tmp_path):
    cfg = L3ACConfig(config_file=GOLDEN / "tiny.toml", model_dir=tmp_path)
    sds = W.synthetic_state_dicts(cfg.network_config, seed=1)
    W.save_state_dicts(sds, cfg.model_path)
    assert sorted(p.name for p in cfg.model_path.iterdir()) == sorted(f"{m}.pt" for m in W.MODULE_NAMES)
    codec = l3ac_amd.get_model(GOLDEN / "tiny.toml", model_dir=tmp_path)
    assert torch.equal(codec.network.state_dicts()["decoder"]["blocks.0.bias"], sds["decoder"]["blocks.0.bias"])
    (cfg.model_path / "decoder.pt").unlink()
    with pytest.raises(FileNotFoundError):
        l3ac_amd.get_model(GOLDEN / "tiny.toml", model_dir=tmp_path)
    with pytest.raises(FileNotFoundError):
        l3ac_amd.get_model("1kbps", model_dir=tmp_path / "nowhere")


def test_no_cpu_fallback():
    codec = l3ac_amd.get_model(GOLDEN / "tiny.toml", synthetic_seed=3)
    with pytest.raises(RuntimeError, match="eval"):
        codec.encode_audio(torch.zeros(1, 100))
    codec.network.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        codec.encode_audio(torch.zeros(1, 100))
    with pytest.raises(RuntimeError, match="no CPU path"):
        codec.decode_audio(indices=torch.zeros(1, 4, dtype=torch.int32))
    with pytest.raises(NotImplementedError):
        codec.network.train()
    x, n = codec.network.preprocess(torch.zeros(2, 25))
    assert x.shape == (2, 36) and n == 25


def test_library_exports_every_declared_symbol():
    header = (REPO / "include" / "l3ac_hip.h").read_text()
    declared = set(re.findall(r"\b(l3ac_[a-z0-9_]+)\s*\(", header))
    declared -= {"l3ac_ctx"}
    assert len(declared) >= 25
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load_library()  # raises if the .so is missing or a symbol is absent
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.l3ac_abi_version() == _capi.ABI_VERSION
    # struct layout agrees with the header (9 scalars + 6 stage arrays + the level array)
    assert ctypes.sizeof(_capi.Config) == 4 * (9 + 6 * _capi.MAX_STAGES + _capi.MAX_LEVELS)
    # argument validation happens before any device work
    assert lib.l3ac_reserve(None, 1, 1) != 0 and b"null context" in lib.l3ac_last_error()
    assert lib.l3ac_hop_length(None) == 0


def test_bf16x3_split_is_exact():
    """The operand split of the bf16x3 kernels (include/l3ac_hip.h, kernels/split_bf16.hpp), evaluated by the library's
    host routine that also builds the weight images — no GPU involved: three bf16 planes add back to the fp32 value
    EXACTLY, each plane is the round-to-nearest-even bf16 of the residual before it, and the planes shrink by 2^-8 each
    (which is what bounds the three dropped cross products by 2^-26 |a.w|)."""
    import numpy as np

    lib = _capi.load_library()
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.standard_normal(200_000).astype(np.float32),
        (rng.standard_normal(100_000) * np.exp(rng.uniform(-30, 30, 100_000))).astype(np.float32),  # wide dynamic range
        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 1e-30, 255.99998], np.float32),
    ])
    n = x.size
    planes = np.zeros((3, n), np.uint16)
    lib.l3ac_split3_host(x.ctypes.data_as(ctypes.c_void_p), n, planes.ctypes.data_as(ctypes.c_void_p))
    p = (planes.astype(np.uint32) << 16).view(np.float32)  # bf16 bit patterns -> fp32 values

    # exact reconstruction, summed smallest first in fp32 (each partial sum is itself exact)
    back = (p[2] + p[1]) + p[0]
    assert np.array_equal(back.view(np.uint32), x.view(np.uint32)) or np.array_equal(back, x)
    assert np.array_equal(back, x)

    def bf16_rne(v):  # numpy restatement of round-to-nearest-even to bf16
        u = v.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

    r1 = x - p[0]
    r2 = r1 - p[1]
    assert np.array_equal(p[0], bf16_rne(x)) and np.array_equal(p[1], bf16_rne(r1)) and np.array_equal(p[2], bf16_rne(r2))
    ax = np.abs(x).astype(np.float64)
    assert np.all(np.abs(p[1]) <= ax * 2.0 ** -8) and np.all(np.abs(p[2]) <= ax * 2.0 ** -16)


def test_macs_match_the_survey_and_model_info_reports_them():
    """SURVEY.md Appendix A (probed on the reference): 3 831.3 MMAC conv/linear per 1 s clip at 1kbps, 3 365.8 at 3kbps;
    get_model_info (reference l3ac/__init__.py:28-51) carries the count for its 10 s default."""
    import l3ac_amd
    from l3ac_amd.config import L3ACConfig, resolve_config_file
    from l3ac_amd.macs import path_macs
    for tag, mmac in (("1kbps", 3831.3), ("3kbps", 3365.8)):
        mc = L3ACConfig(config_file=resolve_config_file(tag)).network_config
        m = path_macs(mc, 16000)
        assert abs(m["conv_linear"] / 1e6 - mmac) < 0.06, (tag, m)
        assert m["total"] == m["conv_linear"] + m["transformer_linear"] + m["attention"]
    codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    info = l3ac_amd.get_model_info(codec.network)
    assert set(info) >= {"macs", "params", "codebook_size", "frame_rate", "bps", "receptive_field"}
    assert info["macs"] == path_macs(codec.network.mc, 160000)["total"]
    assert info["codebook_size"] == 117649 and abs(info["bps"] - 998.2) < 0.1 and abs(info["frame_rate"] - 59.26) < 0.01
    assert info["macs"] == l3ac_amd.get_model_info(codec)["macs"]  # the codec itself is accepted too


def test_traffic_summary_is_tied_to_the_kernel_sources(tmp_path):
    """bench.py attaches profiles/*/traffic.json only when it was collected on this build's kernel sources."""
    import json

    import bench
    roof = {"kernel": "gemm_split_kernel"}
    t = {"source_sha256": bench.source_fingerprint(), "workload": "1kbps b256 s16000 split",
         "kernels": {"gemm_split_kernel<false>": {"launches": 2, "hbm_bytes_per_launch_corrected": 100.0},
                     "gemm_split_kernel<true>": {"launches": 2, "hbm_bytes_per_launch_corrected": 300.0}}}
    (tmp_path / "traffic.json").write_text(json.dumps(t))
    bench.attach_traffic(roof, tmp_path, "1kbps b256 s16000 split")
    assert roof["traffic"] == 200.0 and roof["traffic_stale"] is False
    t["source_sha256"] = "0" * 16
    (tmp_path / "traffic.json").write_text(json.dumps(t))
    roof = {"kernel": "gemm_split_kernel"}
    bench.attach_traffic(roof, tmp_path, "1kbps b256 s16000 split")
    assert roof["traffic"] is None and roof["traffic_stale"] is True and roof["traffic_of_stale_profile"] == 200.0
    roof = {"kernel": "gemm_split_kernel"}
    bench.attach_traffic(roof, tmp_path, "3kbps b256 s16000 split")  # another workload: not attached at all
    assert roof["traffic"] is None and "traffic_stale" not in roof
    # the figure belongs to ONE kernel: an instantiation is matched by its full name, never averaged with its siblings (round 3
    # printed the mean of conv_unit_wide_kernel<256> and <192> beside the algorithmic bytes of <256> alone)
    t = {"source_sha256": bench.source_fingerprint(), "workload": "1kbps b256 s16000 split",
         "kernels": {"conv_unit_wide_kernel<256>": {"launches": 15, "hbm_bytes_per_launch_corrected": 1600.0},
                     "conv_unit_wide_kernel<192>": {"launches": 10, "hbm_bytes_per_launch_corrected": 150.0},
                     "conv_unit_ring_kernel<RGeo<96, 8, 2, 6, 3, false, 1, false>, 96>": {"launches": 15, "hbm_bytes_per_launch_corrected": 566.0},
                     "conv_unit_ring_kernel<RGeo<48, 16, 1, 0, 0, true, 1, false>, 48>": {"launches": 10, "hbm_bytes_per_launch_corrected": 623.0},
                     "row_kernel<2, 2, unsigned int, 12, 1>": {"launches": 5, "hbm_bytes_per_launch_corrected": 1.0},
                     "row_kernel<2, 2, unsigned int, 6, 1>": {"launches": 5, "hbm_bytes_per_launch_corrected": 2.0}}}
    (tmp_path / "traffic.json").write_text(json.dumps(t))
    for kernel, want in (("conv_unit_wide_kernel<256>", 1600.0), ("conv_unit_wide_kernel<192>", 150.0),
                         ("conv_unit_ring_kernel<96>", 566.0), ("conv_unit_ring_kernel<48>", 623.0),
                         ("row_kernel<LERP,CN>", None),       # abbreviated arguments that name no single instantiation: not attached
                         ("conv_unit_wide_kernel<128>", None)):
        roof = {"kernel": kernel, "algorithmic_bytes_per_launch": 800.0}
        bench.attach_traffic(roof, tmp_path, "1kbps b256 s16000 split")
        assert roof["traffic"] == want, (kernel, roof)
        if want is not None:
            assert roof["traffic_over_algorithmic"] == want / 800.0 and len(roof["traffic_kernels"]) == 1
    # the fingerprint covers SOURCES only: objects of a tagged diagnostic build next to them (csrc/build_<tag>/*.hip.o travel to the
    # GPU box with the tree) must not change it — they once did, and a profile collected beside such a directory went stale when it was removed
    csrc = bench.REPO / "l3ac_amd" / "csrc"
    before = bench.source_fingerprint()
    stray = csrc / "build_fingerprint_test"
    stray.mkdir(exist_ok=True)
    try:
        (stray / "kernels_x.hip.o").write_bytes(b"not a source")
        (stray / "y.hpp").write_text("// inside a build directory")
        assert bench.source_fingerprint() == before
    finally:
        for f in stray.iterdir():
            f.unlink()
        stray.rmdir()


def test_chunk_bookkeeping_matches_the_reference_restatement():
    """l3ac_amd.chunking.ChunkData (product, any dim) against oracle/chunk_oracle.ChunkData (reference l3ac/codec.py:159-188,
    dim 0): same chunks when cutting, same stream when merging, for ragged lengths, one-hop and multi-token overlaps."""
    import torch

    from l3ac_amd.chunking import ChunkData, plan
    from oracle.chunk_oracle import ChunkData as RefChunkData
    g = torch.Generator().manual_seed(0)
    for n, chunk_len, prefix in ((23, 8, 3), (80000, 16200, 270), (81001, 16000 // 270 * 270, 270), (7, 8, 1), (8, 8, 7), (9, 8, 7), (1000, 10, 9)):
        data = torch.randn(n, generator=g)
        a = ChunkData(chunk_len, prefix, original_data=data).chunk_data
        b = RefChunkData(chunk_len, prefix, original_data=data).chunk_data
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
        assert torch.equal(ChunkData(chunk_len, prefix, chunk_data=a).data, RefChunkData(chunk_len, prefix, chunk_data=b).data)
        assert torch.equal(ChunkData(chunk_len, prefix, chunk_data=a).data, data)  # cut then merge is the identity
    # token features (T, C) are cut / merged along dim 0 as well; the product also accepts another dim
    feat = torch.randn(37, 5, generator=g)
    a = ChunkData(10, 4, original_data=feat).chunk_data
    assert torch.equal(ChunkData(10, 4, chunk_data=a).data, feat)
    ft = feat.T.contiguous()
    at = ChunkData(10, 4, original_data=ft, dim=1).chunk_data
    assert all(torch.equal(x.T, y) for x, y in zip(at, a)) and torch.equal(ChunkData(10, 4, chunk_data=at, dim=1).data, ft)
    assert plan(270, 5 * 16000, 1) == (79920, 270)  # reference geometry at 1kbps: window rounded to hops, one hop of overlap
    with pytest.raises(ValueError):
        plan(270, 1000, 4)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` (no WORLD_SIZE) starts torch.distributed.run itself, one rank per GPU, rendezvous on
    127.0.0.1: the command it would run, without touching a GPU (BASELINE.json config 4)."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--print-launch"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = plan["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(str(REPO / "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]  # the ranks get the same flags, minus --print-launch
    assert plan["env"]["MASTER_ADDR"] == "127.0.0.1" and plan["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a rank started by the launcher (WORLD_SIZE set) with a mismatching --gpus still refuses
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env={**env, "WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_integration_snippet_binding_matches_the_library():
    """INTEGRATION.md §2 is executable: the block is extracted and executed (tests/helpers.py::integration_snippet), its `_Cfg` has
    the size and field offsets of the struct this package passes to `l3ac_create`, it asks the LIBRARY for the ABI version (round 3's
    text passed a literal 2 to an ABI-3 library), and the tensor names + values its `folded_tensors()` produced on the reference's own
    `EnCodec` (tests/golden/binding_names.npz, written by make_golden.py in the build container) are exactly the names + values
    `l3ac_amd.weights.folded_weights` hands to `l3ac_create` for the same weights."""
    import numpy as np

    from tests.helpers import integration_snippet
    ns = integration_snippet()
    cfg_a, cfg_b = ns["_Cfg"], _capi.Config
    assert ctypes.sizeof(cfg_a) == ctypes.sizeof(cfg_b)
    assert [(n, getattr(cfg_a, n).offset, getattr(cfg_a, n).size) for n, _ in cfg_a._fields_] == \
           [(n, getattr(cfg_b, n).offset, getattr(cfg_b, n).size) for n, _ in cfg_b._fields_]
    ta, tb = ns["_Tensor"], _capi.Tensor
    assert ctypes.sizeof(ta) == ctypes.sizeof(tb) and [n for n, _ in ta._fields_] == [n for n, _ in tb._fields_]
    assert ns["_lib"].l3ac_abi_version() == _capi.ABI_VERSION
    text = (REPO / "INTEGRATION.md").read_text()
    assert "abi_version=_lib.l3ac_abi_version()" in text and not re.search(r"abi_version=\d", text)
    assert isinstance(ns["_lib"].l3ac_last_error(), (bytes, type(None)))  # restype set: a 64-bit char* is not truncated to int
    fx = np.load(GOLDEN / "binding_names.npz")
    for tag in ("tiny", "1kbps", "3kbps"):
        cfg_file = GOLDEN / "tiny.toml" if tag == "tiny" else resolve_config_file(tag)
        mc = L3ACConfig(config_file=cfg_file).network_config
        folded = W.folded_weights(W.synthetic_state_dicts(mc, seed=int(fx[f"{tag}_seed"])))
        names = sorted(folded)
        assert names == [str(n) for n in fx[f"{tag}_names"]], set(names) ^ set(map(str, fx[f"{tag}_names"]))
        assert [folded[k].numel() for k in names] == fx[f"{tag}_numel"].tolist()
        s = np.array([folded[k].double().sum().item() for k in names])
        a = np.array([folded[k].double().abs().sum().item() for k in names])
        assert (np.abs(s - fx[f"{tag}_sum"]) <= 1e-6 * np.maximum(1.0, fx[f"{tag}_abssum"])).all()
        np.testing.assert_allclose(a, fx[f"{tag}_abssum"], rtol=1e-6)
        # the attributes HipPath.__init__ reads from the REFERENCE's ModelConfig have the values this package's config gives
        for attr in ("feature_dim", "encoder_dims", "encoder_depths", "compress_rates", "decoder_dims", "decoder_depths", "decode_rates",
                     "en_coder_depth", "en_coder_window_size", "en_coder_compress_rate", "hop_length"):
            assert np.array_equal(np.array(getattr(mc, attr), dtype=np.int64), fx[f"{tag}_mc_{attr}"]), attr
        assert list(mc.levels) == fx[f"{tag}_mc_levels"].tolist() == list(mc.vq_config["levels"])


def test_bench_rank_path_runs_at_world_2_on_cpu():
    """bench.py's N > 1 code — self-launch through torch.distributed.run, rendezvous on 127.0.0.1, the ranks' warm-up, barriers,
    async output gathers retired one step late, per-rank times all-gathered, max over ranks, rank 0's JSON relayed as the LAST line of
    stdout, launcher's exit code — executed with two real processes (`--dry-run-cpu`: gloo ranks on CPU, the oracle stand-in codec
    of tests/standin_codec.py).  The first run of this code with more than one rank must not be the driver's 8-GPU run."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu",
                        "--codec-factory", "tests.standin_codec:make_codec", "--batch", "4", "--seconds", "0.05"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert line["dry_run_cpu"] is True and line["metric"].startswith("DRY RUN") and line["collective_backend"] == "gloo"
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert len(line["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-6       # the slowest rank's time is the step time
    assert abs(line["value"] - 2 * 4 * 800 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]  # whole-job samples / max time
    assert line["gathered_shapes"] == [[8, 67], [8, 804]]                              # both ranks' clips: indices and waveforms
    assert line["roofline"] is None and "cpu_baseline" not in line
    # without the factory, or with the factory outside a dry run, it refuses: measurements always run the HIP library
    for extra in (["--dry-run-cpu"], ["--codec-factory", "tests.standin_codec:make_codec"]):
        r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2"] + extra, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0


def test_bench_rank_path_runs_at_world_8_at_config_4_shape():
    """BASELINE config 4 (2048 clips sharded over 8 ranks, outputs gathered) through bench.py's N-rank path with EIGHT real processes on
    CPU (gloo), 256 clips x 1 s per rank at the 1kbps geometry (tests/standin_codec.py::make_shape_codec: the real shapes, trivial
    arithmetic): the launcher's rendezvous, 8 barriers, async gathers of [2048, 60] tokens and [2048, 16200] samples retired one step
    late, per-rank times, max over ranks, the relayed JSON line.  The first real 8-GPU run can then only fail on RCCL, not on plumbing."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run-cpu",
                        "--codec-factory", "tests.standin_codec:make_shape_codec"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run_cpu"] is True and line["metric"].startswith("DRY RUN") and line["collective_backend"] == "gloo"
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["scaling"] == "weak"   # (rccl_ranks: an all-reduce of ones over the group — gloo here)
    assert line["config"]["batch_per_gpu"] == 256 and line["config"]["samples_per_clip"] == 16000
    assert line["gathered_shapes"] == [[2048, 60], [2048, 16200]]
    assert len(line["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-6
    assert abs(line["value"] - 8 * 256 * 16000 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    assert "summary" in line and list(line)[-1] == "summary"   # the compact summary is the LAST key of the line


def test_bench_summary_is_compact_and_kernel_names_do_not_alias():
    """the line's trailing `summary` object must survive a record that keeps a 2 000-character tail (< 400 characters
    with every field filled); and the traffic summary of `gemm_split_kernel` must not be attached to `gemm_split_kernel_w256` or the
    other way round (round 6 added a kernel whose name extends an existing one)."""
    import json

    import bench
    out = {"ms_per_step": 13.3186, "fsq_kernel": {"frac": 0.6101, "frac_of_copy_ceiling": 0.9263, "copy_ceiling_best_residency": {"frac_of_peak": 0.6601}},
           "configs": {"stream_1s_graph": {"ms_per_chunk": 0.8440, "pipelined_encode_decode": {"ms_per_chunk": 0.6223}}, "3kbps_b256": {"ms_per_step": 12.0881}},
           "exact_f32_mfma_route": {"ms_per_step": 23.3478}, "cpu_baseline": {"index_agreement": {"split": {"mismatches": 0}, "exact": {"mismatches": 0}}},
           "roofline": {"kernel": "conv_unit_wide_kernel<256>", "frac": 0.5860}, "kernels": [{"name": "gemm_split_kernel", "ms": 1.561}, {"name": "gemm_split_kernel_w256", "ms": 1.308},
                                                                                              {"name": "gemm_split_conv_kernel", "ms": 0.109}]}
    sm = bench.summary_of(out)
    assert len(json.dumps(sm)) < 400
    assert sm["chunk_ms"] == 0.844 and sm["fsq_frac"] == 0.61 and sm["index_mismatches"] == {"split": 0, "exact": 0} and sm["gemm_split_ms"] == 2.869
    assert bench.summary_of({})["ms_per_step"] is None  # a dry run / a partial line still yields the object
    profiled = ["gemm_split_kernel<false>", "gemm_split_kernel<true>", "gemm_split_kernel_w256<3>", "gemm_split_conv_kernel", "conv_unit_wide_kernel<256, 2>",
                "conv_unit_wide_kernel<192, 2>"]
    assert bench.match_traffic_kernels("gemm_split_kernel", profiled) == ["gemm_split_kernel<false>", "gemm_split_kernel<true>"]
    assert bench.match_traffic_kernels("gemm_split_kernel_w256", profiled) == ["gemm_split_kernel_w256<3>"]
    assert bench.match_traffic_kernels("conv_unit_wide_kernel<256>", profiled) == ["conv_unit_wide_kernel<256, 2>"]
