#!/usr/bin/env python3
"""Per-kernel counter averages of one tools/pmc_ab.sh directory: every counter of every pass, averaged per launch, for the kernels
whose name contains one of the comma-separated patterns (default: all kernels with >= 1 % of the busy cycles)."""
import collections, csv, glob, sys


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0].strip()


d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else "")
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{d}/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        c = per[short(r["Kernel_Name"])][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "/sq/" in f:
            t = per[short(r["Kernel_Name"])]["_ns"]
            t[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            t[1] += 1
for name in sorted(per, key=lambda n: -per[n].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
    if pat and not any(p in name for p in pat.split(",")):
        continue
    c = {k: v[0] / max(v[1], 1) for k, v in per[name].items()}
    n = per[name].get("GRBM_GUI_ACTIVE", [0, 0])[1]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if cyc <= 0:
        continue
    line = [f"{name[:60]:60s} n={n:3d} us={c.get('_ns', 0) / 1e3:8.1f} clk={cyc / max(c.get('_ns', 1), 1):.2f}GHz",
            f"mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.3f}",
            f"valu_active={4 * c.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * cyc):.3f}",
            f"wait_any={c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}",
            f"wait_inst={c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}"]
    for k in sorted(c):
        if k.startswith(("SQ_INSTS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES", "SQ_LDS", "TCC", "TCP", "FETCH", "WRITE")):
            line.append(f"{k}={c[k]:.4g}")
    print(" ".join(line))
