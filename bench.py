#!/usr/bin/env python3
"""Headline benchmark: audio samples/s, encode_audio + decode_audio, 1kbps @ 16 kHz, batch 256 x 1 s per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic clips already resident in HBM.  With N > 1 every
rank (one process per GPU) runs its own 256 clips (weak scaling, clips are independent) and the quantiser indices
and waveforms are all-gathered over RCCL at the end of the step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import torch

PEAK_F32_TFLOPS = 157.3  # MI355X fp32 MFMA = fp32 vector peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (same guide: ~2.5 PFLOP/s)
# gemm_split_kernel evaluates every fp32 MAC as 6 bf16 plane products (kernels/gemm_split.hip): its roofline in
# fp32-equivalent FLOPs is the bf16 peak / 6
PEAK_SPLIT_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0    # HBM3E spec


def algorithmic_gflop_per_clip_second(mc):
    """MACs of every conv / linear / attention product of the path for a 1 s clip, as GFLOP."""
    from l3ac_amd.weights import HEADS, en_decoder_layout, en_encoder_layout, trans_geometry
    hop, sr = mc.hop_length, 16000
    t0 = -(-sr // hop) * hop
    macs = 0.0
    unit = lambda c, t: t * (7 * c + 8 * c * c)
    t = t0
    macs += t * (140 + 1600 + 81 * mc.encoder_dims[0])
    for i, c in enumerate(mc.encoder_dims):
        macs += mc.encoder_depths[i] * unit(c, t)
        if i + 1 < len(mc.encoder_dims):
            s = mc.compress_rates[i]
            t //= s
            macs += t * s * c * mc.encoder_dims[i + 1]
    macs += t * 3 * mc.encoder_dims[-1] * mc.feature_dim
    dim = mc.feature_dim
    dh, inner, ffi = trans_geometry(dim)
    layer = lambda n: n * (3 * inner * dim + inner * dim + 2 * ffi * dim + ffi * dim) + HEADS * dh * n * (n + 1)
    frames = t
    n = frames
    for prefix, _, depth in en_encoder_layout(mc):
        macs += depth * layer(n)
        if prefix == "down_trans.trans":
            n //= mc.en_coder_compress_rate
            macs += n * mc.en_coder_compress_rate * dim * dim
    macs += n * 2 * len(mc.levels) * dim
    for prefix, _, depth in en_decoder_layout(mc):
        if prefix == "up_trans.trans":
            n *= mc.en_coder_compress_rate
        macs += depth * layer(n)
    t = frames
    macs += t * 3 * dim * mc.decoder_dims[0]
    for i, s in enumerate(mc.decode_rates):
        c = mc.decoder_dims[i]
        macs += mc.decoder_depths[i] * unit(c, t) + t * c * mc.decoder_dims[i + 1] + t * (28 + 4 * c)
        t *= s
    c = mc.decoder_dims[-1]
    macs += t * (3 * (7 * c * c + c * c) + 7 * c)
    return 2.0 * macs / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="1kbps")
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of outputs (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 measured fastest on the 256-thread GPU box; "
                         "torch's default of 128 is 3x slower: tools/experiments/cpu_threads.py)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--pipeline-only", action="store_true",
                    help="skip the stand-alone FSQ launch, the exact-route reference steps and the CPU baseline (profiling runs: "
                         "keeps the kernel trace to the timed workload)")
    ap.add_argument("--gemm", choices=["split", "exact"], default="split",
                    help="split: large fp32 contractions as exact bf16x3 operand splits on the bf16 matrix cores (default); "
                         "exact: every product on the fp32 MFMA instruction")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    import l3ac_amd
    from l3ac_amd import _capi

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    l3ac_amd.set_gemm_split(args.gemm == "split")
    codec = l3ac_amd.get_model(args.config, synthetic_seed=0)  # identical weights on every rank
    codec.network.to(device=dev).eval()
    mc = codec.network.mc
    samples = int(round(args.seconds * codec.config.sample_rate))
    b = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    audio = ((torch.rand(b, samples, generator=g) * 2 - 1) * 0.5).to(dev)
    codec.network.context().reserve(b, samples)
    n_tok = -(-samples // mc.hop_length)
    force_dist = world == 1 and os.environ.get("L3AC_BENCH_FORCE_DIST") == "1"  # test hook: 1-rank RCCL collectives
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    gather = (world > 1 or force_dist) and not args.no_gather
    from l3ac_amd.dist import gather_batch_async

    pending = []  # all-gathers of the previous step, still in flight on RCCL's stream

    def step():
        q, ind = codec.encode_audio(audio)
        wave = codec.decode_audio(q)
        if gather:
            # the only exchange step of the path: outputs to every rank over xGMI.  The collectives are queued behind this
            # step's kernels and overlap the NEXT step's encode; each step retires the previous step's pair.
            now = (gather_batch_async(ind["indices"], world * b, force=force_dist),
                   gather_batch_async(wave, world * b, force=force_dist))
            while pending:
                for h in pending.pop():
                    h.wait()
            pending.append(now)
        return ind, wave

    def drain():
        while pending:
            for h in pending.pop():
                h.wait()

    run = step
    if args.graph:
        assert not gather, "--graph is a single-GPU mode"
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = step()
        run = lambda: (graph.replay(), captured)[1]

    # the quantiser's stand-alone HBM roofline (its own large-N launch) is taken first, on an idle chip: after the MFMA-heavy
    # pipeline the same launch measures ~10 % lower while the clocks recover
    fsq_line = fsq_microbench(codec, dev) if rank == 0 and world == 1 and not args.pipeline_only else None
    for _ in range(args.warmup):
        run()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ind, wave = run()
    drain()  # the last step's gathers are part of the timed work
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * b * samples * args.steps / elapsed
        gflop_clip = algorithmic_gflop_per_clip_second(mc) * args.seconds
        # ---- per-kernel roofline: one extra, untimed step with HIP events around every launch ----------
        with _capi.profile() as prof:
            codec.decode_audio(codec.encode_audio(audio)[0])
        # profile records are tagged "kernel<instantiation> shape": aggregate per kernel (= what rocprofv3's
        # kernel_stats reports), keep the per-shape view of the GEMM for the report
        by_kernel = {}
        for e in prof.entries:
            k = by_kernel.setdefault(e["name"].split(" ")[0], dict(name=e["name"].split(" ")[0], launches=0, ms_total=0.0, flops=0.0, bytes=0.0))
            for f in ("launches", "ms_total", "flops", "bytes"):
                k[f] += e[f]
        shapes = sorted((e for e in prof.entries if " " in e["name"]), key=lambda e: -e["ms_total"])
        kernels = sorted(by_kernel.values(), key=lambda e: -e["ms_total"])
        total_ms = sum(e["ms_total"] for e in kernels)
        dom = kernels[0]
        dom_ms = dom["ms_total"] / dom["launches"]
        ai = dom["flops"] / max(dom["bytes"], 1.0)
        if dom["name"].startswith("gemm_split_kernel"):
            ach = dom["flops"] / dom["ms_total"] / 1e9
            roof = dict(bound="mfma", achieved=ach, peak=PEAK_SPLIT_TFLOPS, unit="TFLOP/s",
                        peak_note="fp32-equivalent FLOPs (2*m*n*k); each fp32 MAC = 6 bf16 MFMA plane products, so peak = dense bf16 "
                                  f"MFMA peak {PEAK_BF16_TFLOPS:g} / 6; hardware rate = 6 x achieved",
                        hw_bf16_tflops=6.0 * ach, vs_exact_f32_mfma_peak=ach / PEAK_F32_TFLOPS)
        elif ai > PEAK_F32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
            roof = dict(bound="mfma", achieved=dom["flops"] / dom["ms_total"] / 1e9, peak=PEAK_F32_TFLOPS, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=dom["bytes"] / dom["ms_total"] / 1e6, peak=PEAK_HBM_GBS, unit="GB/s")
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), which cannot run
        # inside this process: the committed summary of tools/collect_profiles.sh for this same workload is attached
        traffic, traffic_src = None, None
        tfile = REPO / "profiles" / "r01" / "traffic.json"
        if tfile.exists() and args.config == "1kbps" and b == 256 and samples == 16000:
            # rocprofv3 names carry the template arguments (gemm_split_kernel<false>): match on the bare kernel name and
            # average over its instantiations, weighted by launches
            hits = [v for k, v in json.load(open(tfile))["kernels"].items() if k.split("<")[0].strip() == dom["name"].split("<")[0]]
            if hits:
                n = sum(h["launches"] for h in hits)
                traffic = sum(h["hbm_bytes_per_launch_corrected"] * h["launches"] for h in hits) / n
                traffic_src = "profiles/r01/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        roof.update(frac=roof["achieved"] / roof["peak"], traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=dom["bytes"] / dom["launches"], kernel=dom["name"],
                    launches_per_step=dom["launches"], avg_launch_ms=dom_ms, share_of_step=dom["ms_total"] / total_ms)
        out = {
            "metric": "audio samples/sec encode+decode, 1kbps@16kHz, batch 256; indices bit-exact",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ("fp32 storage and fp32 accumulation everywhere; the large channel contractions split each fp32 operand "
                           "EXACTLY into 3 bf16 planes and sum the 6 plane products of order <= 2 on the bf16 matrix cores "
                           "(error vs fp64 <= the fp32 fmaf chain's, tests/test_gpu_blocks.py::test_gemm_split_accuracy); "
                           "all other products use v_mfma_f32_32x32x2_f32") if l3ac_amd.get_gemm_split() else
                          "fp32 everywhere: every product on v_mfma_f32_32x32x2_f32 (--gemm exact)",
            "config": {"workload": f"{args.config} config, {b} x {args.seconds:g} s 16 kHz clips per GPU, "
                                   "encode_audio + decode_audio(q_feature)" + (", RCCL all-gather of indices+waveforms (overlapping the next step)" if gather else ""),
                       "batch_per_gpu": b, "samples_per_clip": samples, "weights": "seeded synthetic (seed 0)",
                       "hipgraph": bool(args.graph)},
            "roofline": roof,
            "e2e": {"algorithmic_gflop_per_step": gflop_clip * b,
                    "achieved_tflops": gflop_clip * b * world / (ms_per_step * 1e-3) / 1e3,
                    "frac_of_f32_mfma_peak": gflop_clip * b / (ms_per_step * 1e-3) / 1e3 / PEAK_F32_TFLOPS,
                    "kernel_ms_sum_profiled": total_ms},
            "kernels": [{"name": e["name"], "launches": e["launches"], "ms": round(e["ms_total"], 4),
                         "tflops": round(e["flops"] / e["ms_total"] / 1e9, 2), "gbs": round(e["bytes"] / e["ms_total"] / 1e6, 1)}
                        for e in kernels],
            "gemm_shapes": [{"name": e["name"], "launches": e["launches"], "ms": round(e["ms_total"], 4),
                             "tflops": round(e["flops"] / e["ms_total"] / 1e9, 2)} for e in shapes[:16]],
        }
        if args.gemm == "split" and world == 1 and not args.graph and not args.pipeline_only:
            # the same step with every product on the exact fp32 MFMA instruction, for reference (5 steps, untimed above)
            l3ac_amd.set_gemm_split(False)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                ind_x, _ = step()
            torch.cuda.synchronize()
            ex = (time.perf_counter() - t1) / 5
            l3ac_amd.set_gemm_split(True)
            out["exact_f32_mfma_route"] = {"ms_per_step": ex * 1e3, "value": b * samples / ex, "unit": "samples/s",
                                           "token_differences_vs_split_route": int((ind_x["indices"] != ind["indices"]).sum())}
        out["fsq_kernel"] = fsq_line
        if not args.no_cpu_baseline and not args.pipeline_only and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(codec, audio, args.cpu_batch, ind, args.cpu_threads)
        try:  # RCCL writes a version banner through C stdio: flush it first so that the JSON line is the last line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if world > 1 or force_dist:
        dist.destroy_process_group()


def fsq_microbench(codec, dev, n_tokens=1 << 22):
    """HBM roofline of the closed-form FSQ kernel on its own: at batch 256 it moves 16 MB per launch (below launch
    latency), so its fraction of the HBM peak is measured at 2^22 tokens (1 052 algorithmic bytes per token)."""
    import ctypes as C

    from l3ac_amd import _capi, weights as W
    mc = codec.network.mc
    d, feat = len(mc.levels), mc.feature_dim
    w = W.folded_weights(codec.network.state_dicts())
    wt = {k: w[f"quantizer.{k}"].to(dev) for k in ("project_in.weight", "project_in.bias", "project_out.weight", "project_out.bias")}
    x = torch.randn(n_tokens, feat, device=dev)
    q = torch.empty_like(x)
    idx = torch.empty(n_tokens, dtype=torch.int32, device=dev)
    li = torch.empty(n_tokens, d, device=dev)
    lib = _capi.load_library()
    lv = (C.c_int32 * d)(*mc.levels)
    s = torch.cuda.current_stream(dev).cuda_stream
    call = lambda: _capi.check(lib.l3ac_fsq_forward(x.data_ptr(), n_tokens, feat, lv, d, wt["project_in.weight"].data_ptr(),
                                                    wt["project_in.bias"].data_ptr(), wt["project_out.weight"].data_ptr(),
                                                    wt["project_out.bias"].data_ptr(), q.data_ptr(), idx.data_ptr(),
                                                    li.data_ptr(), None, s))
    for _ in range(50):  # ~50 ms: lets the memory / fabric clocks ramp from idle before the timed launches
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bytes_per_token = 4 * feat * 2 + 4 + 4 * d
    gbs = n_tokens * bytes_per_token / ms / 1e6
    return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "tokens": n_tokens, "bytes_per_token": bytes_per_token, "ms": ms}


def cpu_baseline(codec, audio, cpu_batch, gpu_ind, threads):
    """The oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores, rank 0 only,
    on a bounded sample of the same workload; also reports index agreement of the GPU run on those clips."""
    import numpy as np

    from l3ac_amd import weights as W
    from oracle import l3ac_oracle as O
    from tests.helpers import index_mismatch_report

    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    x = audio[:cpu_batch].cpu()
    taps = {}
    O.decode_audio(w, mc, O.encode_audio(w, mc, x, taps=taps)[0])  # warm-up
    iters, t0 = 0, time.perf_counter()
    while True:
        q, ind = O.encode_audio(w, mc, x)
        O.decode_audio(w, mc, q)
        iters += 1
        if time.perf_counter() - t0 > 12.0 or iters >= 12:
            break
    dt = (time.perf_counter() - t0) / iters
    n_bad, ok = index_mismatch_report(gpu_ind["indices"][:cpu_batch].cpu().numpy(), ind["indices"].numpy(),
                                      taps["latents"].numpy(), mc.levels, tau=2e-3)
    return {"value": cpu_batch * x.shape[1] / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cpu_batch} of the batch's clips x {iters} iterations of encode_audio+decode_audio "
                      f"({dt:.2f} s each), torch {torch.__version__} CPU, nproc={os.cpu_count()}",
            "gpu_index_mismatches_on_sample": int(n_bad), "sample_tokens": int(np.prod(ind["indices"].shape)),
            "mismatches_are_boundary_flips": bool(ok)}


if __name__ == "__main__":
    main()
