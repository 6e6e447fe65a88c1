#!/usr/bin/env python3
"""Where a legacy_unit_split_kernel workgroup spends its cycles: s_memtime sums per phase of thread 0 of the first and of the last workgroup
(diagnostic build: git apply tools/patches/legacy_unit_stamps.patch, then L3AC_BUILD_TAG=lgstamps L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_LG_STAMPS
python -m l3ac_amd.build; L3AC_LIB_PATH=l3ac_amd/libl3ac_hip_lgstamps.so; the stamp hooks are kept out of the product sources).
usage: tools/lg_stamps.py [batch]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd
from l3ac_amd import _capi

PHASES = ["staging (snake + split of the S tile)", "barrier A", "prefetch issue", "first product", "activation + split", "second product",
          "residual + store", "barrier B + loop"]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.cuda().eval()
ctx = codec.network.context()
lib = ctx.lib
x = torch.randn(batch, 16200, 24, device="cuda")
y = torch.empty(batch, 16200, device="cuda")
call = lambda: _capi.check(lib.l3ac_op_last_block(ctx.handle, x.data_ptr(), batch, 16200, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
for _ in range(2):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    call()
e1.record()
torch.cuda.synchronize()
print(f"l3ac_op_last_block (three LegacyUnits + head), {batch} clips: {e0.elapsed_time(e1) / 5:.3f} ms per call")
if not hasattr(lib, "l3ac_debug_lg_stamps"):
    sys.exit(0)
buf = (C.c_longlong * 20)()
lib.l3ac_debug_lg_stamps(buf, 20, 1)
blk = (C.c_longlong * 2048)()
lib.l3ac_debug_lg_blocks(blk, 1)
reps = 4
for _ in range(reps):
    call()
torch.cuda.synchronize()
lib.l3ac_debug_lg_stamps(buf, 20, 1)
for who, off in (("first workgroup", 0), ("last workgroup", 8)):
    tot = sum(buf[off + i] for i in range(8))
    tiles = buf[16 + off // 8] / reps / 3
    print(f"{who}: {tot / reps / 3:.0f} cycles per LegacyUnit launch (thread 0), {tiles:.1f} tiles = {tot / reps / 3 / max(tiles, 1):.0f} cycles per tile")
    for i, name in enumerate(PHASES):
        print(f"  {name:<40} {buf[off + i] / reps / 3:10.0f} cycles  {100.0 * buf[off + i] / max(tot, 1):5.1f} %")

import numpy as np
lib.l3ac_debug_lg_blocks(blk, 1)
a = np.array(list(blk), dtype=np.float64).reshape(1024, 2)[:512] / reps / 3
life, tiles = a[:, 0], a[:, 1]
print("per workgroup (512): lifetime k cycles min/p10/p50/p90/max", np.percentile(life, [0, 10, 50, 90, 100]).round(-3) / 1e3, "| tiles min/p50/max", tiles.min(), np.median(tiles), tiles.max())
for lo, hi in ((0, 128), (128, 256), (256, 384), (384, 512)):
    print(f"  workgroups {lo:3d}..{hi - 1}: lifetime {life[lo:hi].mean() / 1e3:7.0f} k, tiles {tiles[lo:hi].mean():5.1f}, cycles per tile {life[lo:hi].sum() / tiles[lo:hi].sum():7.0f}")
