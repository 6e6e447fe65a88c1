#!/usr/bin/env python3
"""Where the waves of a conv_unit_ring_kernel workgroup spend their cycles: s_memtime sums per phase of every wave of workgroup 0,
from a diagnostic build (compile kernels/conv_unit_ring.hip with -DL3AC_RING_STAMPS into a tagged library; run with
L3AC_LIB_PATH=<that library>).  The stamps fence the phases (every fragment read is waited for, every product group completed):
read the SHARES, not the lengths.   usage: tools/ring_stamps.py [channels 24|48|96] [batch]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd
from l3ac_amd import _capi

PHASES = ["tile front (dw-conv, LayerNorm, split)", "fragment read -> landed", "products (issue + completion)", "snake / GRN / split",
          "slot barrier", "LDS-DMA issue", "residual + store", "other"]
c = int(sys.argv[1]) if len(sys.argv) > 1 else 96
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
block, frames = {24: ("encoder.blocks.1.0.module", 16200), 48: ("decoder.blocks.10.0.module", 8100), 96: ("decoder.blocks.7.0.module", 2700)}[c]
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.cuda().eval()
ctx = codec.network.context()
ctx.set_option("narrow_ring", 2)
lib = ctx.lib
x = torch.randn(batch, frames, c, device="cuda")
y = torch.empty_like(x)
call = lambda: _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    call()
torch.cuda.synchronize()
n = 16 * 8
buf = (C.c_longlong * n)()
lib.l3ac_debug_ring_stamps(buf, n, 1)
reps = 5
for _ in range(reps):
    call()
torch.cuda.synchronize()
lib.l3ac_debug_ring_stamps(buf, n, 1)
print(f"{block} C={c} batch={batch} frames={frames}: cycles per launch, per wave of workgroup 0 (100 MHz s_memtime ticks x 1 = ticks)")
waves = [w for w in range(16) if sum(buf[8 * w + i] for i in range(8)) > 0]
tot = [sum(buf[8 * w + i] for i in range(8)) / reps for w in waves]
print("  wave:                                  " + " ".join(f"{w:>8d}" for w in waves))
print("  total ticks                            " + " ".join(f"{t:8.0f}" for t in tot))
for i, name in enumerate(PHASES):
    print(f"  {name:<38} " + " ".join(f"{100.0 * buf[8 * w + i] / reps / t:7.1f}%" for w, t in zip(waves, tot)))
