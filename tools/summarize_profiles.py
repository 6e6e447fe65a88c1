#!/usr/bin/env python3
"""Condense a tools/collect_profiles.sh output directory into the files committed under profiles/:

    One set per workload directory tools/collect_profiles.sh wrote (main: no prefix; 3kbps_, b1_, vq_ otherwise):

    kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
    traffic.json       per kernel: launches, average FETCH_SIZE / WRITE_SIZE per launch and the corrected HBM bytes
                       (gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled, as
                       /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes; WRITE_SIZE is exact)
    mfma_util.json     per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), effective clock

traffic.json is stamped with the fingerprint of the kernel sources it was collected on (bench.py: source_fingerprint) and
the workload key; bench.py attaches it as roofline.traffic only while both still match, and says `traffic_stale` otherwise.

usage: tools/summarize_profiles.py gpurun_out/r03_final/raw profiles/r03
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0].strip()


def load(pattern):
    files = glob.glob(pattern)
    return list(csv.DictReader(open(files[0]))) if files else []


WORKLOADS = {"main": ("", "1kbps b256 s16000 split"), "3kbps": ("3kbps_", "3kbps b256 s16000 split"),
             "b1": ("b1_", "1kbps b1 s16000 split, eager"), "vq": ("vq_", "vq_argmin K250047 N42752 D6"),
             "fsq": ("fsq_", "fsq forward 2^22 tokens feat128 D6 (1052 B/token algorithmic = 4.412 GB per launch)")}


def main(src_root, dst):
    src_root = Path(src_root)
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from bench import source_fingerprint
    fp_file = src_root / "source_sha256.txt"  # written on the GPU box by collect_profiles.sh (same snapshot as the run)
    fingerprint = fp_file.read_text().strip() if fp_file.exists() else source_fingerprint()
    for wl, (prefix, key) in WORKLOADS.items():
        if (src_root / wl).is_dir():
            summarize(src_root / wl, Path(dst), prefix, key, fingerprint)


def summarize(src, dst, prefix, workload, fingerprint):
    dst.mkdir(parents=True, exist_ok=True)
    stats = glob.glob(str(src / "trace/*/*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], dst / f"{prefix}kernel_stats.csv")
    if (src / "trace_stdout.txt").exists():
        shutil.copy(src / "trace_stdout.txt", dst / f"{prefix}stdout_under_rocprof.txt")
    per = collections.defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0, wl=0))
    for r in load(str(src / "pmc_fetch/*/*_counter_collection.csv")):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = per[short(r["Kernel_Name"])]
            k["launches"] += 1
            k["fetch_kb"] += float(r["Counter_Value"])
    for r in load(str(src / "pmc_write/*/*_counter_collection.csv")):
        if r["Counter_Name"] == "WRITE_SIZE":
            k = per[short(r["Kernel_Name"])]
            k["wl"] += 1
            k["write_kb"] += float(r["Counter_Value"])
    traffic = {}
    for name, k in per.items():
        if not name or not k["launches"]:
            continue
        fetch = k["fetch_kb"] / k["launches"] * 1024
        write = k["write_kb"] / max(k["wl"], 1) * 1024
        traffic[name] = dict(launches=k["launches"], fetch_size_bytes_per_launch=fetch, write_size_bytes_per_launch=write,
                             hbm_bytes_per_launch_corrected=2 * fetch + write)
    if prefix == "fsq_":  # the north_star kernel: counter bytes against its algorithmic 1 052 B per token
        for name, t in traffic.items():
            t["algorithmic_bytes_per_launch"] = (1 << 22) * 1052
            t["traffic_over_algorithmic"] = t["hbm_bytes_per_launch_corrected"] / t["algorithmic_bytes_per_launch"]
    json.dump(dict(note="FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; averages per launch",
                   source_sha256=fingerprint, workload=workload, kernels=traffic), open(dst / f"{prefix}traffic.json", "w"), indent=1)
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in load(str(src / "pmc_sq/*/*_counter_collection.csv")):
        d = sq[short(r["Kernel_Name"])]
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            d["_n"] += 1
    util = {}
    for name, d in sq.items():
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8  # summed over the 8 XCDs
        if cyc <= 0 or not name:
            continue
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over SIMDs); SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES count quad-cycles per wave
        util[name] = dict(mfma_busy_fraction=d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc),
                          effective_clock_ghz=cyc / max(d["_ns"], 1.0),
                          wait_any_per_wave_cycle=d.get("SQ_WAIT_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                          valu_active_fraction=4.0 * d.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * cyc),
                          valu_insts=d.get("SQ_INSTS_VALU", 0.0), launches_counted=d.get("_n", 0.0))
    json.dump(util, open(dst / f"{prefix}mfma_util.json", "w"), indent=1)
    print(f"wrote {dst}/{prefix}kernel_stats.csv {prefix}traffic.json {prefix}mfma_util.json")
    for name in sorted(traffic, key=lambda n: -traffic[n]["hbm_bytes_per_launch_corrected"] * traffic[n]["launches"])[:8]:
        t = traffic[name]
        u = util.get(name, {})
        print(f"{name[:44]:44s} launches {t['launches']:4d} HBM/launch {t['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB "
              f"mfma busy {u.get('mfma_busy_fraction', float('nan')):.2f} clock {u.get('effective_clock_ghz', float('nan')):.2f} GHz")


if __name__ == "__main__":
    main(*sys.argv[1:3])
