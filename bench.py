#!/usr/bin/env python3
"""Headline benchmark: audio samples/s, encode_audio + decode_audio, 1kbps @ 16 kHz, batch 256 x 1 s per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic clips already resident in HBM.  With N > 1 every
rank (one process per GPU) runs its own 256 clips (weak scaling, clips are independent) and the quantiser indices
and waveforms are all-gathered over RCCL at the end of the step.  Rank 0 prints ONE JSON line.

Besides the headline fields the line carries (N = 1, default flags):
  roofline       the dominant kernel of the step against its own roofline (HIP-event durations measured in this run)
  cpu_baseline   the oracle timed on a bounded sample of the same workload on this box's host cores, plus
                 index_agreement: EVERY token of the headline batch compared with the oracle, on both GEMM routes
  configs        BASELINE.json configs 3 and 5 and the explicit-codebook argmin kernel, each with its own steps x ms
"""
import argparse
import hashlib
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
# bf16x3 kernels evaluate every fp32 MAC as 6 bf16 plane products (kernels/split_bf16.hpp): their roofline in
# fp32-equivalent FLOPs is the bf16 peak / 6
PEAK_SPLIT_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0    # HBM3E spec
SPLIT_KERNELS = ("gemm_split_kernel", "gemm_split_kernel_w256", "gemm_split_conv_kernel", "conv_unit_wide_kernel", "conv_unit_split_kernel", "legacy_unit_split_kernel")


def algorithmic_gflop_per_clip_second(mc):
    """MACs of every conv / linear / attention product of the path for a 1 s clip, as GFLOP."""
    from l3ac_amd.macs import path_macs
    return 2.0 * path_macs(mc, 16000)["total"] / 1e9


def source_fingerprint():
    """sha256 over the kernel sources and headers: what a profiles/*/traffic.json must have been collected on to still
    describe this build (the .so itself is not hashed: it is rebuilt on other boxes)."""
    h = hashlib.sha256()
    csrc = REPO / "l3ac_amd" / "csrc"
    # sources only (*.hip, *.hpp, *.h), never a build directory of any tag (build/, build_<tag>/: their *.hip.o would match "*.h*")
    files = sorted(f for f in csrc.rglob("*") if f.suffix in (".hip", ".hpp", ".h") and f.is_file()
                   and not any(part.startswith("build") for part in f.relative_to(csrc).parts[:-1]))
    files += sorted((REPO / "include").glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def aggregate(entries):
    """Profile records are tagged "kernel<instantiation> shape": aggregate per kernel (= what rocprofv3's kernel_stats
    reports) and keep the per-shape view."""
    by_kernel = {}
    for e in entries:
        name = e["name"].split(" ")[0]
        k = by_kernel.setdefault(name, dict(name=name, launches=0, ms_total=0.0, flops=0.0, bytes=0.0))
        for f in ("launches", "ms_total", "flops", "bytes"):
            k[f] += e[f]
    shapes = sorted((e for e in entries if " " in e["name"]), key=lambda e: -e["ms_total"])
    return sorted(by_kernel.values(), key=lambda e: -e["ms_total"]), shapes


def roofline_of(dom, total_ms):
    """Roofline entry for one aggregated kernel: achieved = algorithmic FLOPs (or bytes) of its launches / their summed
    device time (HIP events on the launch stream)."""
    dom_ms = dom["ms_total"] / dom["launches"]
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    if dom["name"].split("<")[0] in SPLIT_KERNELS:
        ach = dom["flops"] / dom["ms_total"] / 1e9
        roof = dict(bound="mfma", achieved=ach, peak=PEAK_SPLIT_TFLOPS, unit="TFLOP/s",
                    peak_note="fp32-equivalent FLOPs (2 per MAC); each fp32 MAC = 6 bf16 MFMA plane products, so peak = dense bf16 "
                              f"MFMA peak {PEAK_BF16_TFLOPS:g} / 6; hardware rate = 6 x achieved",
                    hw_bf16_tflops=6.0 * ach, vs_exact_f32_mfma_peak=ach / PEAK_F32_TFLOPS)
    elif ai > PEAK_F32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
        roof = dict(bound="mfma", achieved=dom["flops"] / dom["ms_total"] / 1e9, peak=PEAK_F32_TFLOPS, unit="TFLOP/s")
    else:
        roof = dict(bound="hbm", achieved=dom["bytes"] / dom["ms_total"] / 1e6, peak=PEAK_HBM_GBS, unit="GB/s")
    roof.update(frac=roof["achieved"] / roof["peak"], traffic=None,
                algorithmic_bytes_per_launch=dom["bytes"] / dom["launches"], kernel=dom["name"],
                launches_per_step=dom["launches"], avg_launch_ms=dom_ms, share_of_step=dom["ms_total"] / total_ms)
    return roof


def _template_args(name):
    """('base', [top-level template arguments]) of a kernel name, whitespace removed: 'k<RGeo<96, 8>, 96>' -> ('k', ['RGeo<96,8>', '96'])."""
    name = name.replace(" ", "")
    if "<" not in name:
        return name, None
    base, rest = name.split("<", 1)
    rest = rest[:rest.rindex(">")]
    args, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
            continue
        depth += ch == "<"
        depth -= ch == ">"
        cur += ch
    args.append(cur)
    return base, args


def match_traffic_kernels(kernel, profiled):
    """Entries of a traffic.json (`profiled`: rocprofv3's full template names) that are the SAME kernel as `kernel` (the library's
    profile name, which abbreviates template arguments).  Rules, in order: (1) the full name, whitespace aside; (2) a name without
    template arguments stands for every instantiation of that base name (the library aggregates them the same way);
    (3) a name WITH arguments matches the instantiations whose top-level arguments contain all of its own — and is attached only if
    exactly one instantiation does (round 3 averaged conv_unit_wide_kernel<256> with <192> by matching on the base name alone)."""
    base, args = _template_args(kernel)
    exact = [k for k in profiled if k.replace(" ", "") == kernel.replace(" ", "")]
    if exact:
        return exact
    same_base = [k for k in profiled if _template_args(k)[0] == base]
    if args is None:
        return same_base
    hits = [k for k in same_base if set(args) <= set(_template_args(k)[1] or [])]
    return hits if len(hits) == 1 else []


def attach_traffic(roof, profiles_dir, workload_key, prefix=""):
    """HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), which cannot run inside
    this process: the summary tools/collect_profiles.sh + tools/summarize_profiles.py wrote for this same workload is
    attached — only if it was collected on THIS build (kernel-source fingerprint), otherwise it is reported as stale."""
    tfile = Path(profiles_dir) / f"{prefix}traffic.json"
    if not tfile.exists():
        roof.update(traffic=None, traffic_source=None)
        return
    t = json.load(open(tfile))
    fp = source_fingerprint()
    names = match_traffic_kernels(roof["kernel"], list(t["kernels"])) if t.get("workload") == workload_key else []
    hits = [t["kernels"][k] for k in names]
    stale = t.get("source_sha256") != fp
    if hits:
        n = sum(h["launches"] for h in hits)
        val = sum(h["hbm_bytes_per_launch_corrected"] * h["launches"] for h in hits) / n
        roof.update(traffic=None if stale else val, traffic_stale=stale, traffic_kernels=names,
                    traffic_source=f"{tfile.relative_to(REPO) if tfile.is_relative_to(REPO) else tfile} (rocprofv3 --pmc FETCH_SIZE / "
                                   f"WRITE_SIZE; collected on sources {t.get('source_sha256')}, this build {fp})")
        if "algorithmic_bytes_per_launch" in roof and not stale:
            roof["traffic_over_algorithmic"] = val / roof["algorithmic_bytes_per_launch"]
        if stale:
            roof["traffic_of_stale_profile"] = val
    else:
        roof.update(traffic=None, traffic_source=None)


def latest_profiles_dir():
    """profiles/rNN of the highest round that holds a traffic.json (a stale one is reported as stale, never attached)."""
    rounds = sorted(d for d in (REPO / "profiles").glob("r[0-9][0-9]") if (d / "traffic.json").exists())
    return rounds[-1] if rounds else REPO / "profiles"


def launch_command(n_gpus, argv, port=None):
    """The command and environment `python bench.py --gpus N` starts for N > 1: torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = {"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
           "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", "8")}
    return cmd, env


def visible_gpu_count():
    """GPUs this node exposes, counted WITHOUT initialising HIP in this process (it is about to start the ranks as children and
    must stay GPU-untouched): KFD topology nodes with SIMDs, cut down by a *_VISIBLE_DEVICES list when one is set.  None if the
    topology is not readable (then the ranks themselves report a missing device)."""
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    try:
        n = 0
        for d in nodes.iterdir():
            props = dict(l.split(None, 1) for l in (d / "properties").read_text().splitlines() if " " in l)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """Run the N-rank bench as child processes of this (GPU-untouched) one; rank 0's JSON line is relayed as the LAST
    line of stdout, everything else the children print goes to stderr; returns the launcher's exit code."""
    import subprocess
    cmd, env = launch_command(args.gpus, [a for a in sys.argv[1:] if a != "--print-launch"])
    if args.print_launch:
        print(json.dumps({"cmd": cmd, "env": env}))
        return 0
    n_dev = None if args.dry_run_cpu else visible_gpu_count()
    if n_dev is not None and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} GPU(s)", file=sys.stderr)
        return 2
    proc = subprocess.run(cmd, env={**os.environ, **env}, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    json_line = next((l for l in reversed(lines) if l.lstrip().startswith('{"metric"')), None)
    for l in lines:
        if l is not json_line:
            print(l, file=sys.stderr)
    if json_line is not None:
        print(json_line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited 0 but printed no JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def time_steps(run, steps, warmup, drain=lambda: None):
    for _ in range(warmup):
        run()
    drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = run()
    drain()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


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
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (3kbps batch, streaming graph, argmin kernel)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 measured fastest on the 256-thread GPU box; "
                         "torch's default of 128 is 3x slower: tools/cpu_threads.py)")
    ap.add_argument("--agreement-clips", type=int, default=-1,
                    help="clips of the batch whose tokens are compared with the oracle (default: all of them)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--pipeline-only", action="store_true",
                    help="skip the stand-alone FSQ launch, the exact-route reference steps, the configs object and the CPU baseline "
                         "(profiling runs: keeps the kernel trace to the timed workload)")
    ap.add_argument("--gemm", choices=["split", "exact"], default="split",
                    help="split: large fp32 contractions as exact bf16x3 operand splits on the bf16 matrix cores (default); "
                         "exact: every product on the fp32 MFMA instruction")
    ap.add_argument("--profiles-dir", default=str(latest_profiles_dir()),
                    help="directory whose traffic.json (rocprofv3 PMC summary of this workload) is attached as roofline.traffic")
    ap.add_argument("--print-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the torch.distributed.run command this would start, and exit")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="plumbing check, NOT a measurement: the same launcher / rank / barrier / gather / timing / JSON code on CPU "
                         "ranks over gloo, with the codec taken from --codec-factory (tests/standin_codec.py); the line says dry_run_cpu")
    ap.add_argument("--codec-factory", default=None,
                    help="module:function returning an object with the codec's surface (--dry-run-cpu only)")
    args = ap.parse_args()
    dry = args.dry_run_cpu
    if dry and not args.codec_factory:
        raise SystemExit("--dry-run-cpu needs --codec-factory module:function (the product has no CPU path)")
    if args.codec_factory and not dry:
        raise SystemExit("--codec-factory is a --dry-run-cpu option: measurements always run the HIP library")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one fresh process per GPU BEFORE anything here touches the GPU (a process
        # that has initialised HIP must never be replaced by exec; children are started, their exit code is returned)
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    import l3ac_amd
    from l3ac_amd import _capi

    if dry:  # CPU ranks over gloo: everything below is the code the GPU ranks run, minus the device calls
        import importlib
        dev = torch.device("cpu")
        sync = lambda: None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
        mod, fn = args.codec_factory.split(":")
        codec = getattr(importlib.import_module(mod), fn)()
    else:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node exposes {torch.cuda.device_count()} GPU(s)")
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        sync = torch.cuda.synchronize
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
    if not dry:
        codec.network.context().reserve(b, samples)
    force_dist = world == 1 and not dry and os.environ.get("L3AC_BENCH_FORCE_DIST") == "1"  # test hook: 1-rank RCCL collectives
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    gather = (world > 1 or force_dist) and not args.no_gather
    from l3ac_amd.dist import PendingGathers, gather_batch_async

    # all-gathers of the previous step, still in flight on RCCL's stream (l3ac_amd/dist.py; gloo-tested on CPU)
    pending = PendingGathers()

    def step():
        q, ind = codec.encode_audio(audio)
        wave = codec.decode_audio(q)
        if gather:
            # the only exchange step of the path: outputs to every rank over xGMI.  The collectives are queued behind this
            # step's kernels and overlap the NEXT step's encode; each step retires the previous step's pair.
            pending.push(gather_batch_async(ind["indices"], world * b, force=force_dist),
                         gather_batch_async(wave, world * b, force=force_dist))
        return ind, wave

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
    extras = rank == 0 and world == 1 and not args.pipeline_only and not dry
    fsq_line = fsq_microbench(codec, dev) if extras else None
    rccl_ranks = None
    if world > 1 or force_dist:  # proof that RCCL sees every rank: an all-reduce of ones
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
    for _ in range(args.warmup):
        run()
    pending.drain()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ind, wave = run()
    pending.drain()  # the last step's gathers are part of the timed work
    sync()
    if world > 1:
        dist.barrier()
    elapsed_local = elapsed = time.perf_counter() - t0
    gathered_shapes = [tuple(r.shape) for r in pending.results] if gather else None
    per_rank_ms = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * b * samples * args.steps / elapsed
        gflop_clip = algorithmic_gflop_per_clip_second(mc) * args.seconds
        if dry:
            kernels, shapes, total_ms, roof, runner_up = [], [], 0.0, None, None
        else:
            # ---- per-kernel roofline: one extra, untimed step with HIP events around every launch ----------
            with _capi.profile() as prof:
                codec.decode_audio(codec.encode_audio(audio)[0])
            kernels, shapes = aggregate(prof.entries)
            total_ms = sum(e["ms_total"] for e in kernels)
            roof = roofline_of(kernels[0], total_ms)
            attach_traffic(roof, args.profiles_dir, f"{args.config} b{b} s{samples} {args.gemm}")
            # the two largest kernels are within 1 % of each other (gemm_split: 14 launches of 8 shapes; conv_unit_wide<256>: 3
            # launches): the runner-up's roofline is printed too, so that the line does not flip its story with the box
            runner_up = None
            if len(kernels) > 1:
                runner_up = roofline_of(kernels[1], total_ms)
                attach_traffic(runner_up, args.profiles_dir, f"{args.config} b{b} s{samples} {args.gemm}")
        out = {
            "metric": "audio samples/sec encode+decode, 1kbps@16kHz, batch 256; indices bit-exact" if not dry else
                      "DRY RUN ON CPU (gloo ranks, stand-in codec): plumbing check of the N-rank bench path, not a measurement",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ("fp32 storage and fp32 accumulation everywhere; the large channel contractions split each fp32 operand "
                           "EXACTLY into 3 bf16 planes and sum the 6 plane products of order <= 2 on the bf16 matrix cores "
                           "(error vs fp64 <= the fp32 fmaf chain's, tests/test_gpu_blocks.py::test_gemm_split_accuracy); "
                           "all other products use v_mfma_f32_32x32x2_f32") if l3ac_amd.get_gemm_split() else
                          "fp32 everywhere: every product on v_mfma_f32_32x32x2_f32 (--gemm exact)",
            "dry_run_cpu": dry,
            "parity_note": "'indices bit-exact' is the BASELINE metric's wording; what is measured: the quantiser kernel is bit-exact "
                           "for identical inputs (reference known-answer vectors incl. exact rounding boundaries), and end to end "
                           "every token of this batch is compared with the CPU oracle in cpu_baseline.index_agreement (mismatch "
                           "count and boundary margin reported; fp32 summation order differs between any two implementations)",
            "config": {"workload": f"{args.config} config, {b} x {args.seconds:g} s 16 kHz clips per GPU, "
                                   "encode_audio + decode_audio(q_feature)" + (", RCCL all-gather of indices+waveforms (overlapping the next step)" if gather else ""),
                       "batch_per_gpu": b, "samples_per_clip": samples, "weights": "seeded synthetic (seed 0)",
                       "hipgraph": bool(args.graph)},
            "roofline": roof,
            "roofline_runner_up": runner_up,
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
        if rccl_ranks is not None:
            out["rccl_ranks"] = rccl_ranks  # an all-reduce of ones over the process group (gloo_ranks in a dry run)
            out["per_rank_ms_per_step"] = per_rank_ms if per_rank_ms is not None else [elapsed_local / args.steps * 1e3]
            out["gathered_shapes"] = gathered_shapes  # [indices, waveforms] of the last retired step: world * batch rows each
        if dry:
            out["collective_backend"] = "gloo"
        ind_exact = None
        if args.gemm == "split" and world == 1 and not args.graph and not args.pipeline_only and not dry:
            # the same step with every product on the exact fp32 MFMA instruction, for reference (5 steps, untimed above)
            l3ac_amd.set_gemm_split(False)
            dt, (ind_x, _) = time_steps(step, 5, 2, pending.drain)
            l3ac_amd.set_gemm_split(True)
            ind_exact = ind_x["indices"]
            out["exact_f32_mfma_route"] = {"ms_per_step": dt / 5 * 1e3, "value": b * samples * 5 / dt, "unit": "samples/s", "steps": 5,
                                           "token_differences_vs_split_route": int((ind_x["indices"] != ind["indices"]).sum())}
        out["fsq_kernel"] = fsq_line
        if extras and not args.no_configs and not args.graph:
            out["configs"] = other_configs(dev, args)
        if not args.no_cpu_baseline and not args.pipeline_only and world == 1 and not dry:  # rank 0 at N = 1 only
            routes = {"split" if l3ac_amd.get_gemm_split() else "exact": ind["indices"]}
            if ind_exact is not None:
                routes["exact"] = ind_exact
            out["cpu_baseline"] = cpu_baseline(codec, audio, args.cpu_batch, routes, args.cpu_threads, args.agreement_clips)
        if roof is not None and fsq_line is not None:  # the north_star's one numeric target (>= 0.60 of the HBM peak on the quantiser kernel)
            roof["north_star_kernel"] = {"kernel": "fsq_forward128_kernel", "bound": "hbm", "frac": round(fsq_line["frac"], 4),
                                         "achieved": round(fsq_line["achieved"], 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                         "frac_of_copy_ceiling": round(fsq_line.get("frac_of_copy_ceiling", float("nan")), 4), "target": 0.60}
        out["summary"] = summary_of(out)  # LAST key, <= 400 characters: survives a record that keeps only the tail of the line
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


def summary_of(out):
    """The figures a round is about, compact and LAST in the line (the driver's record keeps a 2 000-character tail)."""
    r3 = lambda v: None if v is None else round(float(v), 3)
    fsq = out.get("fsq_kernel") or {}
    cfgs = out.get("configs") or {}
    stream = cfgs.get("stream_1s_graph") or {}
    agree = (out.get("cpu_baseline") or {}).get("index_agreement") or {}
    roof = out.get("roofline") or {}
    gemm = [k for k in out.get("kernels", []) if k["name"].startswith("gemm_split_kernel")]
    return {"ms_per_step": r3(out.get("ms_per_step")), "fsq_frac": r3(fsq.get("frac")), "fsq_frac_of_copy_ceiling": r3(fsq.get("frac_of_copy_ceiling")),
            "fsq_copy_ceiling_frac": r3((fsq.get("copy_ceiling_best_residency") or fsq.get("copy_ceiling") or {}).get("frac_of_peak")),
            "chunk_ms": r3(stream.get("ms_per_chunk")), "chunk_pipelined_ms": r3((stream.get("pipelined_encode_decode") or {}).get("ms_per_chunk")),
            "ms_3kbps": r3((cfgs.get("3kbps_b256") or {}).get("ms_per_step")), "exact_route_ms": r3((out.get("exact_f32_mfma_route") or {}).get("ms_per_step")),
            "index_mismatches": {k: v.get("mismatches") for k, v in agree.items()} or None,
            "roofline_kernel": roof.get("kernel"), "roofline_frac": r3(roof.get("frac")),
            "gemm_split_ms": r3(sum(k["ms"] for k in gemm)) if gemm else None}


def _sysfs_clocks():
    """Current shader / memory clock levels as sysfs prints them (the starred line of pp_dpm_sclk / pp_dpm_mclk of the first
    amdgpu card that shows them): plain file reads, no HIP call.  None where the files are not readable."""
    import glob
    out = {}
    for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
        out[name] = None
        for f in sorted(glob.glob(f"/sys/class/drm/card*/device/{name}")):
            try:
                lines = open(f).read().splitlines()
            except OSError:
                continue
            cur = [ln.strip() for ln in lines if ln.rstrip().endswith("*")]
            if cur:
                out[name] = cur[0]
                break
    return out


def fsq_microbench(codec, dev, n_tokens=1 << 22, window=20, rounds=5, min_warm_s=0.5, max_warm_s=4.0):
    """HBM roofline of the closed-form FSQ kernel on its own: at batch 256 it moves 16 MB per launch (below launch
    latency), so its fraction of the HBM peak is measured at 2^22 tokens (1 052 algorithmic bytes per token).

    Protocol (the same on a box that has just been leased and on one that has run for minutes): launch windows of `window`
    launches until two consecutive windows agree within 1 % and at least `min_warm_s` have passed (the clocks ramp from
    idle for a few hundred ms); then `rounds` interleaved rounds of (kernel window, copy-ceiling window).  Every round is
    printed; `achieved` is the MEDIAN round, `frac_of_copy_ceiling` the median of the per-round ratios."""
    import ctypes as C
    import statistics

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
    with_copy = feat == 128 and d == 6  # the same grid and access pattern with no arithmetic: what this box's HBM gives that pattern
    copy = lambda: _capi.check(lib.l3ac_fsq_copy_ceiling(x.data_ptr(), n_tokens, q.data_ptr(), idx.data_ptr(), li.data_ptr(), s))
    copy_at = lambda r: (lambda: _capi.check(lib.l3ac_fsq_copy_ceiling_at(x.data_ptr(), n_tokens, q.data_ptr(), idx.data_ptr(), li.data_ptr(), r, s)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, lead=3):  # one window: ms per launch (the first `lead` launches after a change of kernel are not timed)
        for _ in range(lead):
            fn()
        e0.record()
        for _ in range(window):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / window

    clocks_before = _sysfs_clocks()
    t0, prev, warm = time.perf_counter(), None, []
    while True:
        ms = timed(call)
        warm.append(round(ms, 4))
        waited = time.perf_counter() - t0
        if (prev is not None and abs(ms - prev) <= 0.01 * prev and waited >= min_warm_s) or waited >= max_warm_s:
            break
        prev = ms
    best_res, by_res = 0, {}
    if with_copy:
        timed(copy)
        # the copy kernel needs no LDS and few registers, so ITS best residency is not the quantiser's (2 workgroups per CU):
        # one window per residency, the fastest one is measured beside the kernel in every round below and is the ratio's denominator
        for r in (3, 4, 6, 8):
            by_res[r] = timed(copy_at(r))
        best_res = min(by_res, key=by_res.get)
    k_ms, c_ms, cb_ms = [], [], []
    for _ in range(rounds):
        k_ms.append(timed(call))
        if with_copy:
            c_ms.append(timed(copy))
            cb_ms.append(timed(copy_at(best_res)))
    bytes_per_token = 4 * feat * 2 + 4 + 4 * d
    to_gbs = lambda ms: n_tokens * bytes_per_token / ms / 1e6
    ms = statistics.median(k_ms)
    gbs = to_gbs(ms)
    out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
           "tokens": n_tokens, "bytes_per_token": bytes_per_token, "ms": ms,
           "protocol": f"windows of {window} launches until two agree within 1 % and >= {min_warm_s} s have passed, then {rounds} "
                       "interleaved rounds of (kernel, copy ceiling), each window behind 3 untimed launches of its own kernel; achieved = median round",
           "warmup_windows_ms": warm, "warmup_s": round(waited, 3),
           "rounds_gbs": [round(to_gbs(v), 1) for v in k_ms],
           "clocks_sysfs": {"before": clocks_before, "after": _sysfs_clocks()}}
    if with_copy:
        cms = statistics.median(c_ms)
        cgbs = to_gbs(cms)
        out["copy_ceiling"] = {"achieved": cgbs, "unit": "GB/s", "frac_of_peak": cgbs / PEAK_HBM_GBS, "ms": cms,
                               "rounds_gbs": [round(to_gbs(v), 1) for v in c_ms],
                               "what": "fsq_copy_ceiling_kernel: the grid and residency of the kernel it is measured beside (fsq_forward128_kernel), "
                                       "its token order and per-lane non-temporal loads / stores incl. the 4-B and 24-B side outputs, no arithmetic"}
        out["frac_of_copy_ceiling_matched_residency"] = statistics.median([c / k for k, c in zip(k_ms, c_ms)])
        best = [min(c, cb) for c, cb in zip(c_ms, cb_ms)]  # per round: the faster of the two residencies
        bgbs = to_gbs(statistics.median(best))
        out["copy_ceiling_best_residency"] = {"achieved": bgbs, "unit": "GB/s", "frac_of_peak": bgbs / PEAK_HBM_GBS, "workgroups_per_cu": best_res,
                                              "ms_by_workgroups_per_cu": {str(k): round(v, 4) for k, v in by_res.items()},
                                              "rounds_gbs": [round(to_gbs(v), 1) for v in cb_ms],
                                              "what": "the same copy kernel at ITS OWN best residency (no LDS, few registers: more workgroups per CU "
                                                      "than the quantiser can hold); frac_of_copy_ceiling divides by the faster of the two per round"}
        out["frac_of_copy_ceiling"] = statistics.median([c / k for k, c in zip(k_ms, best)])
        out["frac_of_copy_ceiling_rounds"] = [round(c / k, 4) for k, c in zip(k_ms, best)]
    return out


F2_AGREEMENT_CLIPS = 32
F2_TOKENS = {}  # config name -> (audio [32, T] on the host, GPU tokens of those clips), filled by other_configs


def other_configs(dev, args):
    """BASELINE.json configs 3 and 5 and the explicit-codebook L2-argmin kernel the north star names, each with its own
    steps x ms (same timing discipline as the headline: warm-up, synchronise, K steps, synchronise)."""
    import ctypes as C

    import l3ac_amd
    from l3ac_amd import _capi
    out = {}
    # ---- config 3: 3kbps, 250 047-entry implicit codebook, batch 256 x 1 s -------------------------------------------
    codec3 = l3ac_amd.get_model("3kbps", synthetic_seed=0)
    codec3.network.to(device=dev).eval()
    mc3 = codec3.network.mc
    g = torch.Generator(device="cpu").manual_seed(1234)
    audio3 = ((torch.rand(256, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    codec3.network.context().reserve(256, 16000)
    step3 = lambda: codec3.decode_audio(codec3.encode_audio(audio3)[0])
    steps = 10
    dt, _ = time_steps(step3, steps, 2)
    with _capi.profile() as prof:
        step3()
    kernels, _ = aggregate(prof.entries)
    total_ms = sum(e["ms_total"] for e in kernels)
    gflop3 = algorithmic_gflop_per_clip_second(mc3) * 256
    roof3 = roofline_of(kernels[0], total_ms)
    attach_traffic(roof3, args.profiles_dir, f"3kbps b256 s16000 {args.gemm}", prefix="3kbps_")
    out["3kbps_b256"] = {"workload": "3kbps config, 256 x 1 s clips, encode_audio + decode_audio(q_feature), one MI355X",
                         "steps": steps, "ms_per_step": dt / steps * 1e3, "value": 256 * 16000 * steps / dt, "unit": "samples/s",
                         "tokens_per_step": 256 * (-(-16000 // mc3.hop_length)), "algorithmic_gflop_per_step": gflop3,
                         "achieved_tflops": gflop3 * steps / dt / 1e3, "roofline": roof3,
                         "kernels": [{"name": e["name"], "launches": e["launches"], "ms": round(e["ms_total"], 4)} for e in kernels[:8]]}
    del audio3, codec3
    # ---- the other two shipped configs at the headline batch: 0k75bps (hop 360) and 1k5bps (hop 180) ------------
    for name in ("0k75bps", "1k5bps"):
        codec_f = l3ac_amd.get_model(name, synthetic_seed=0)
        codec_f.network.to(device=dev).eval()
        mcf = codec_f.network.mc
        g = torch.Generator(device="cpu").manual_seed(1234)
        audio_f = ((torch.rand(256, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
        codec_f.network.context().reserve(256, 16000)
        hold = {}

        def step_f():
            q, ind = codec_f.encode_audio(audio_f)
            hold["idx"] = ind["indices"]
            return codec_f.decode_audio(q)
        steps = 10
        dt, _ = time_steps(step_f, steps, 2)
        gflop = algorithmic_gflop_per_clip_second(mcf) * 256
        out[f"{name}_b256"] = {"workload": f"{name} config, 256 x 1 s clips, encode_audio + decode_audio(q_feature), one MI355X",
                               "steps": steps, "ms_per_step": dt / steps * 1e3, "value": 256 * 16000 * steps / dt, "unit": "samples/s",
                               "tokens_per_step": 256 * (-(-16000 // mcf.hop_length)), "algorithmic_gflop_per_step": gflop,
                               "achieved_tflops": gflop * steps / dt / 1e3}
        # tokens of the first clips, kept for the cpu_baseline leg (the only place of this file where the oracle may run)
        F2_TOKENS[name] = (audio_f[:F2_AGREEMENT_CLIPS].cpu(), hold["idx"][:F2_AGREEMENT_CLIPS].cpu())
        del audio_f, codec_f
    # ---- config 5: a 10-minute clip streamed as 600 x 1 s chunks through ONE captured hipGraph (encode + decode) -----
    codec1 = l3ac_amd.get_model("1kbps", synthetic_seed=0)
    codec1.network.to(device=dev).eval()
    codec1.network.context().reserve(1, 16000)
    g = torch.Generator(device="cpu").manual_seed(99)
    chunks = ((torch.rand(600, 16000, generator=g) * 2 - 1) * 0.5).to(dev)  # the 10-minute clip, resident in HBM
    static_in = torch.zeros(1, 16000, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        codec1.decode_audio(indices=codec1.encode_audio(static_in)[1]["indices"])
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q1, ind1 = codec1.encode_audio(static_in)
        wave1 = codec1.decode_audio(indices=ind1["indices"])
    tokens = torch.empty(600, ind1["indices"].shape[1], dtype=torch.int32, device=dev)
    waves = torch.empty(600, wave1.shape[1], device=dev)

    def replay_all():
        for i in range(600):
            static_in.copy_(chunks[i:i + 1])   # the chunk is already in HBM: a 64-KB device copy into the graph's input
            graph.replay()
            tokens[i].copy_(ind1["indices"][0])
            waves[i].copy_(wave1[0])
    replay_all()  # warm-up pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    replay_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eager_ind = codec1.encode_audio(chunks[599:600])[1]["indices"]
    out["stream_1s_graph"] = {"workload": "1kbps config, 10-minute 16 kHz clip streamed as 600 x 1 s chunks (B = 1), one captured hipGraph "
                                          "of encode_audio + decode_audio(indices) replayed per chunk; tokens and waveform kept per chunk",
                              "steps": 600, "ms_per_chunk": dt / 600 * 1e3, "x_real_time": 600.0 / dt, "value": 600 * 16000 / dt,
                              "unit": "samples/s", "last_chunk_tokens_equal_eager": bool(torch.equal(tokens[599:600], eager_ind))}
    # ---- the same stream with encoder and decoder PIPELINED (reported beside the figure above, never instead of it): one graph whose two
    # branches run encode_audio(chunk i) and decode_audio(tokens of chunk i - 1) side by side on two contexts (two workspaces) — what a
    # streaming deployment does, sender and receiver being different processes anyway.  The waveform of a chunk leaves one replay later.
    try:
        codec_d = l3ac_amd.get_model("1kbps", synthetic_seed=0)
        codec_d.network.to(device=dev).eval()
        codec_d.network.context().reserve(1, 16000)
        n_tok = ind1["indices"].shape[1]
        tok_prev = torch.zeros(1, n_tok, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm both contexts outside capture
            codec1.encode_audio(static_in)
            codec_d.decode_audio(indices=tok_prev)
        torch.cuda.current_stream().wait_stream(s)
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            cap = torch.cuda.current_stream()
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                wave_p = codec_d.decode_audio(indices=tok_prev)
            _, ind_p = codec1.encode_audio(static_in)
            cap.wait_stream(side)

        def replay_pipelined():
            for i in range(601):  # replay i encodes chunk i (the last one: a flush) and decodes chunk i - 1
                static_in.copy_(chunks[min(i, 599):min(i, 599) + 1])
                graph2.replay()
                if i > 0:
                    waves[i - 1].copy_(wave_p[0])
                if i < 600:
                    tokens[i].copy_(ind_p["indices"][0])
                tok_prev.copy_(ind_p["indices"])
        replay_pipelined()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        replay_pipelined()
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        eager_wave = codec1.decode_audio(indices=eager_ind)
        out["stream_1s_graph"]["pipelined_encode_decode"] = {
            "what": "one graph, two branches on two contexts: encode_audio(chunk i) beside decode_audio(tokens of chunk i - 1); 601 replays for "
                    "600 chunks; a chunk's waveform leaves one replay later.  Throughput of the stream, NOT the latency of a chunk "
                    "(ms_per_chunk above is the sequential encode + decode of one chunk)",
            "steps": 601, "ms_per_chunk": dtp / 600 * 1e3, "x_real_time": 600.0 / dtp,
            "last_chunk_tokens_equal_eager": bool(torch.equal(tokens[599:600], eager_ind)),
            "last_chunk_wave_equal_eager": bool(torch.equal(waves[599:600], eager_wave))}
        del codec_d
    except Exception as e:  # a diagnostic extra: never let it take the line down
        out["stream_1s_graph"]["pipelined_encode_decode"] = {"error": repr(e)}
    del chunks, waves
    # ---- explicit-codebook L2 nearest neighbour at config-3 size: K = 250 047 codes, N = 42 752 queries -------------
    from oracle import l3ac_oracle as O  # the checker: codebook table + closed-form answers for the parity count
    lib = _capi.load_library()
    levels = list(mc3.levels)
    k = 1
    for lv in levels:
        k *= lv
    n = 256 * 167
    g = torch.Generator(device="cpu").manual_seed(7)
    z = torch.randn(n, len(levels), generator=g) * 1.2
    queries = torch.tanh(z).to(dev).contiguous()
    codebook = O.codebook(levels).to(dev).contiguous()
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nbytes = lib.l3ac_vq_argmin_scratch_bytes(n, k, 0)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    call = lambda: _capi.check(lib.l3ac_vq_argmin(queries.data_ptr(), n, codebook.data_ptr(), k, len(levels), idx.data_ptr(),
                                                  scratch.data_ptr(), nbytes, 0, stream))
    steps = 5
    dt, _ = time_steps(call, steps, 2)
    listed = int(scratch[:4].view(torch.int32).item())  # screened form: queries that took the full direct-form search
    # the same call on the direct-form scan (the form the screened one must agree with on EVERY query, not only the clear ones)
    nb_scan = lib.l3ac_vq_argmin_scratch_bytes(n, k, 1)
    sc_scan = torch.empty(nb_scan, dtype=torch.uint8, device=dev)
    idx_scan = torch.empty_like(idx)
    call_scan = lambda: _capi.check(lib.l3ac_vq_argmin(queries.data_ptr(), n, codebook.data_ptr(), k, len(levels), idx_scan.data_ptr(),
                                                       sc_scan.data_ptr(), nb_scan, 1, stream))
    dt_scan, _ = time_steps(call_scan, 3, 1)
    _, idx_ref, _ = O.fsq_quantize(z, levels)
    lvt = torch.tensor(levels, dtype=torch.float64)
    scaled = (torch.tanh(z.double()) + 1) / 2 * (lvt - 1)
    clear = ((scaled - scaled.floor()) - 0.5).abs().min(dim=1).values > 1e-4
    ms = dt / steps * 1e3
    flop = 18.0 * n * k
    out["vq_argmin"] = {"workload": f"explicit-codebook L2 argmin, K = {k} codes x {len(levels)} dims, N = {n} queries (3kbps, 256 x 1 s)",
                        "steps": steps, "ms": ms, "algorithmic_gflop": flop / 1e9, "bound": "fp32 matrix pipe (v_mfma_f32_32x32x2_f32)",
                        # priced on what the kernel EXECUTES: 2 D N K FLOP of scores on the fp32 matrix pipe (the 3 D N K algorithmic
                        # FLOP of the direct form against the same-rate VALU peak read 1.5 x higher: algorithmic_frac_of_valu_peak)
                        "achieved": 12.0 * n * k / ms / 1e9, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": 12.0 * n * k / ms / 1e9 / PEAK_F32_TFLOPS,
                        "algorithmic_tflops": flop / ms / 1e9, "algorithmic_frac_of_valu_peak": flop / ms / 1e9 / PEAK_F32_TFLOPS,
                        "algorithmic_bytes": 24 * n + 24 * k + 4 * n, "gbs": (24 * n + 24 * k + 4 * n) / ms / 1e6,
                        "equal_to_closed_form_where_margin_gt_1e-4": bool(torch.equal(idx.cpu()[clear], idx_ref[clear])),
                        "queries_compared": int(clear.sum()),
                        "form": "screened: |c|^2 - 2 q.c per 32 x 32 tile on v_mfma_f32_32x32x2_f32 picks a block of 16 candidate codes per "
                                "query, the direct form (the definition) decides among them; queries whose two best blocks are within "
                                "128 u Qd take the full direct-form search",
                        "matrix_pipe_gflop": 12.0 * n * k / 1e9, "matrix_pipe_frac": 12.0 * n * k / ms / 1e9 / PEAK_F32_TFLOPS,
                        "queries_sent_to_full_search": listed,
                        "direct_form_scan": {"ms": dt_scan / 3 * 1e3, "frac": flop / (dt_scan / 3 * 1e3) / 1e9 / PEAK_F32_TFLOPS,
                                             "equal_on_every_query": bool(torch.equal(idx, idx_scan))}}
    return out


def cpu_baseline(codec, audio, cpu_batch, gpu_indices_by_route, threads, agreement_clips):
    """The oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores, rank 0 only, on a
    bounded sample of the same workload.  Then — untimed — the oracle encodes the WHOLE batch once, and every token the GPU
    produced (per GEMM route) is compared with it: `index_agreement`."""
    import numpy as np

    from l3ac_amd import weights as W
    from oracle import l3ac_oracle as O
    from tests.helpers import index_agreement

    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    mc = codec.network.mc
    w = W.folded_weights(codec.network.state_dicts())
    x = audio[:cpu_batch].cpu()
    O.decode_audio(w, mc, O.encode_audio(w, mc, x)[0])  # warm-up
    iters, t0 = 0, time.perf_counter()
    while True:
        q, _ = O.encode_audio(w, mc, x)
        O.decode_audio(w, mc, q)
        iters += 1
        if time.perf_counter() - t0 > 12.0 or iters >= 12:
            break
    dt = (time.perf_counter() - t0) / iters
    n_cmp = audio.shape[0] if agreement_clips < 0 else min(agreement_clips, audio.shape[0])
    t1 = time.perf_counter()
    idx_ref, lat_ref = [], []
    for b0 in range(0, n_cmp, 32):
        taps = {}
        _, ind = O.encode_audio(w, mc, audio[b0:min(b0 + 32, n_cmp)].cpu(), taps=taps)
        idx_ref.append(ind["indices"])
        lat_ref.append(taps["latents"])
    idx_ref, lat_ref = torch.cat(idx_ref).numpy(), torch.cat(lat_ref).numpy()
    agreement = {route: index_agreement(gi[:n_cmp].cpu().numpy(), idx_ref, lat_ref, mc.levels)
                 for route, gi in gpu_indices_by_route.items()}
    first = next(iter(agreement.values()))
    f2 = {}
    for name, (audio_f, gpu_idx) in F2_TOKENS.items():  # the other shipped configs: first 32 clips of their 256-clip batch
        import l3ac_amd
        from l3ac_amd.config import L3ACConfig, resolve_config_file
        mcf = L3ACConfig(config_file=resolve_config_file(name)).network_config
        wf = W.folded_weights(W.synthetic_state_dicts(mcf, seed=0))
        taps = {}
        _, ind = O.encode_audio(wf, mcf, audio_f, taps=taps)
        f2[name] = index_agreement(gpu_idx.numpy(), ind["indices"].numpy(), taps["latents"].numpy(), mcf.levels)
    return {"index_agreement_other_configs": f2, "value": cpu_batch * x.shape[1] / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cpu_batch} of the batch's clips x {iters} iterations of encode_audio+decode_audio "
                      f"({dt:.2f} s each), torch {torch.__version__} CPU, nproc={os.cpu_count()}",
            "sample_tokens": int(np.prod(idx_ref.shape)), "index_agreement": agreement,
            "gpu_index_mismatches_on_sample": first["mismatches"],
            "agreement_oracle_encode_s": round(time.perf_counter() - t1, 1)}


if __name__ == "__main__":
    main()
