"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_parse.py DIR [substr]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:70], r["Grid_Size"])
        if sub and sub not in key[0]:
            continue
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for key, v in agg.items():
        us = sum(dur[key]) / len(dur[key]) / 1e3
        out = {c: sum(x) / len(x) for c, x in v.items()}
        extra = ""
        if "GRBM_GUI_ACTIVE" in out and us > 0:
            extra = f" clock~{out['GRBM_GUI_ACTIVE'] / 8 / us / 1e3:.2f}GHz"
        print(key, f"{us:.1f}us", {c: f"{x:.4g}" for c, x in out.items()}, extra)
