#!/usr/bin/env python3
"""Time the wide ConvUnit (front end + main kernel) through l3ac_op_conv_unit, for A/B runs of two builds on ONE box:

    L3AC_LIB_PATH=.../libA.so python tools/wide_bench.py; L3AC_LIB_PATH=.../libB.so python tools/wide_bench.py
"""
import hashlib
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
lib = ctx.lib
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for c, block, batch, frames in ((256, "decoder.blocks.4.1.module", 256, 900), (192, "encoder.blocks.7.0.module", 256, 180)):
    x = torch.randn(batch, frames, c, device="cuda")
    y = torch.empty_like(x)
    f = lambda: _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ms = sorted(ts)[len(ts) // 2]
    digest = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]  # same box + same seed: equal digests = bit-identical results
    print(f"C={c} rows={batch * frames}: {ms:.4f} ms per unit (front end + main), {batch * frames * 16.0 * c * c / ms / 1e9:.1f} TFLOP/s fp32-equivalent, output sha256 {digest}")
