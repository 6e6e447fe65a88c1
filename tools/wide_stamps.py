#!/usr/bin/env python3
"""Where a pass of conv_unit_wide_kernel spends its cycles (diagnostic build only):

    L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_WIDE_STAMPS python -m l3ac_amd.build      # in the build container
    gpurun -- python tools/wide_stamps.py [C] [batch] [frames]

The kernel stamps s_memtime (shader cycles) at the phase boundaries of every pass of wave 0 of every workgroup into a
__device__ array that only this tool reads.  Phases: 0 pass start | 1 after the plane loads | 2 after the entry barrier |
3 after the first product of hidden tile 0 | 4 after the hidden-tile loop | 5 after the last tile | 6 after the residual store
(written by the NEXT stamp 0 / the final stamp).  (tools/experiments/conv_unit_wide_v1.hip (retired: git show 1a7dadb:tools/experiments/conv_unit_wide_v1.hip) / _v2.hip carry the same stamps.)
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

c = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 900
block = {256: "decoder.blocks.4.1.module", 192: "encoder.blocks.7.0.module", 96: "decoder.blocks.7.0.module"}[c]
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
import os
part = os.environ.get("STAMP_PART", "")  # experiment: run on the first / second half of an allocation twice the size
if part:
    xb = torch.randn(2 * batch, frames, c, device="cuda")
    yb = torch.empty_like(xb)
    x = xb[:batch] if part == "lo" else xb[batch:]
    y = yb[:batch] if part == "lo" else yb[batch:]
else:
    x = torch.randn(batch, frames, c, device="cuda")
    y = torch.empty_like(x)
lib = ctx.lib
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # back-to-back launches (no idle gap: clocks as inside a step); the stamps are the LAST one's
for _ in range(reps - 1):
    _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
e0.record()
_capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
e1.record()
torch.cuda.synchronize()
print(f"C={c} rows={batch * frames}: {e0.elapsed_time(e1):.3f} ms")
NB = 512
n = NB * 16 * 8
buf = np.zeros(n, dtype=np.uint64)
fn = lib.l3ac_debug_wide_stamps
fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p), n) == 0
st = buf.reshape(NB, 16, 8).astype(np.int64)
passes = min(15, int(np.ceil(np.ceil(batch * frames / 32) / 4 / (512 if c <= 96 else 256))))
names = ["plane loads", "entry barrier", "first product (tile 0)", "hidden-tile loop", "last tile", "residual + store"]
tot = []
for p in range(passes):
    blk = [b for b in range(NB) if st[b, p, 0] > 0 and st[b, p, 5] > 0]
    d = np.array([[st[b, p, i + 1] - st[b, p, i] for i in range(5)] for b in blk])
    # the pass ends where the next one starts (stamp 0), or — last pass of a block — at the exit stamp, which the kernel
    # stores under the pass counter's final value
    nxt = np.array([(st[b, p + 1, 0] if st[b, p + 1, 0] > 0 else st[b, p + 1, 6]) - st[b, p, 5] for b in blk])
    med = list(np.median(d, axis=0)) + [float(np.median(nxt))]
    tot.append(med)
    print(f"pass {p} ({len(blk)} blocks): " + " | ".join(f"{n_} {m / 1e3:.1f}k" for n_, m in zip(names, med)) + f" | total {sum(med) / 1e3:.1f}k cycles")
tot = np.array(tot)
print("share of a pass: " + ", ".join(f"{n_} {100 * v:.1f}%" for n_, v in zip(names, tot.sum(0) / tot.sum())))
span = np.array([st[b, :, 6].max() - st[b, 0, 0] for b in range(NB) if st[b, 0, 0] > 0])
print(f"kernel span per block: median {np.median(span) / 1e3:.0f}k cycles, max {span.max() / 1e3:.0f}k; "
      f"ideal MFMA-only: {passes * (4 * c // 32) * (c // 16 + c // 16) * 6 * 32 / 1e3:.0f}k")

if os.environ.get("STAMP_DETAIL"):
    t00 = min(st[b, 0, 0] for b in range(NB) if st[b, 0, 0] > 0)
    for p in range(passes):
        blk = [b for b in range(NB) if st[b, p, 0] > 0 and st[b, p, 5] > 0]
        loop = np.array([st[b, p, 4] - st[b, p, 3] for b in blk])
        start = np.array([st[b, p, 0] - t00 for b in blk])
        q = np.percentile(loop, [0, 10, 50, 90, 100]) / 1e3
        qs = np.percentile(start, [0, 10, 50, 90, 100]) / 1e3
        lo = np.median(loop[: len(blk) // 2]) / 1e3
        hi = np.median(loop[len(blk) // 2:]) / 1e3
        print(f"pass {p}: loop min/p10/p50/p90/max {q.round(1).tolist()}  blocks<half {lo:.1f}k >=half {hi:.1f}k | pass start (k ticks after the first) {qs.round(0).tolist()}")
