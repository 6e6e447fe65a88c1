#!/usr/bin/env python3
"""bench.py's FSQ stand-alone measurement (fsq_microbench: warm-up to stability, interleaved kernel / copy-ceiling rounds) on its own:
   [L3AC_LIB_PATH=<tagged library>] python tools/fsq_protocol.py [tag]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench, l3ac_amd
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
r = bench.fsq_microbench(codec, torch.device("cuda:0"))
print(sys.argv[1] if len(sys.argv) > 1 else "-", json.dumps({k: r[k] for k in ("frac", "frac_of_copy_ceiling", "ms", "rounds_gbs", "frac_of_copy_ceiling_rounds")} | {"copy_frac": r["copy_ceiling"]["frac_of_peak"]}))
