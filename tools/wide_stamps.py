#!/usr/bin/env python3
"""Where conv_unit_wide_kernel spends its cycles (diagnostic build only):

    L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_WIDE_STAMPS python -m l3ac_amd.build      # in the build container
    gpurun -- python tools/wide_stamps.py [C] [batch] [frames]

The kernel stamps s_memtime (shader cycles) for wave 0 (group A) and wave 4 (group B) of every workgroup at: 0 start of a
pass (= start of its memory phase) | 1 end of the memory phase | 2 end of the compute phase, into a __device__ array that
only this tool reads.
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
block = {256: "decoder.blocks.4.1.module", 192: "encoder.blocks.7.0.module"}[c]
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
x = torch.randn(batch, frames, c, device="cuda")
y = torch.empty_like(x)
lib = ctx.lib
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
rows = batch * frames
print(f"C={c} rows={rows}: {ms:.3f} ms = {rows * (16.0 * c * c) / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
n = 256 * 2 * 16 * 4
buf = np.zeros(n, dtype=np.uint64)
fn = lib.l3ac_debug_wide_stamps
fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p), n) == 0
st = buf.reshape(256, 2, 16, 4).astype(np.int64)
nb = 4 * c // 32
ideal_compute = nb * 2 * (c // 32) * 6 * 2 * 16  # MFMA cycles of one tile: blocks x (2 tiles x NK + CT) k steps x 6 x 16 cycles
for g in range(2):
    for p in range(16):
        ok = [b for b in range(256) if st[b, g, p, 0] > 0 and st[b, g, p, 2] > 0]
        if not ok:
            continue
        mem = np.median([st[b, g, p, 1] - st[b, g, p, 0] for b in ok])
        comp = np.median([st[b, g, p, 2] - st[b, g, p, 1] for b in ok])
        print(f"group {'AB'[g]} pass {p} ({len(ok)} blocks): memory phase {mem / 1e3:.1f}k | compute phase {comp / 1e3:.1f}k "
              f"(its own MFMA cycles: {ideal_compute / 1e3:.1f}k; the SIMD's pipe serves two such waves)")
first = st[:, :, 0, 0]
last = st[:, :, :, 1:3].max(axis=(2, 3))
span = (last.max(axis=1) - np.where(first > 0, first, np.iinfo(np.int64).max).min(axis=1))[first.max(axis=1) > 0]
passes = int(np.ceil(np.ceil(rows / 16) / 8 / 256))
print(f"kernel span per block: median {np.median(span) / 1e3:.0f}k cycles, max {span.max() / 1e3:.0f}k; MFMA-only lower bound "
      f"{passes * 2 * ideal_compute / 1e3:.0f}k (2 tiles per SIMD and pass, {passes} passes); clock ~{np.median(span) / ms / 1e6:.2f} GHz")
