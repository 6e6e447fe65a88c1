#!/usr/bin/env python3
"""Stamps of gemm_split_kernel_w256d (diagnostic build, as tools/w256_stamps.py) through the C = 512 ConvUnit: start | entry of the SECOND
tile's k loop | its exit | end of the workgroup."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
lib = ctx.lib
s = torch.cuda.current_stream().cuda_stream
batch, frames = 136, 180
x = torch.randn(batch, frames, 512, device="cuda")
y = torch.empty_like(x)
ctx.reserve(256, 16000)
for _ in range(20):
    _capi.check(lib.l3ac_op_conv_unit(ctx.handle, b"decoder.blocks.1.2.module", x.data_ptr(), batch, frames, y.data_ptr(), s))
torch.cuda.synchronize()
NB = 4096
buf = np.zeros(NB * 4, dtype=np.uint64)
fn = lib.l3ac_debug_w256_stamps
fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p), NB * 4) == 0
st = buf.reshape(NB, 4).astype(np.int64)[:256]
st = st[st[:, 0] > 0]
d = np.diff(st, axis=1)
tiles = -(-(-(-batch * frames // 128)) // 8) * 8 * 8 / 256
print(f"{len(st)} workgroups, ~{tiles:.2f} tiles each: first tile + prologues {np.median(d[:, 0]) / 1e3:.1f}k | second tile's k loop {np.median(d[:, 1]) / 1e3:.1f}k = "
      f"{np.median(d[:, 1]) / 16:.0f} per k tile (MFMA floor 3072) | rest {np.median(d[:, 2]) / 1e3:.1f}k | whole workgroup {np.median(st[:, 3] - st[:, 0]) / 1e3:.1f}k cycles")
