#!/usr/bin/env python3
"""The C = 512 ConvUnit (dw-conv + LayerNorm -> pw_conv1 + snake + GRN -> pw_conv2 + residual) through l3ac_op_conv_unit at the batch size
of the step's clip groups: time and a digest of the output, for A/B runs of L3AC_GEMM_W256 (0 / 1 / 5 ...) on one box."""
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
lib = ctx.lib
s = torch.cuda.current_stream().cuda_stream
block = "decoder.blocks.1.2.module"
for batch, frames in ((136, 180), (120, 180), (3, 180), (17, 97)):
    g = torch.Generator(device="cuda").manual_seed(batch)
    x = torch.randn(batch, frames, 512, device="cuda", generator=g)
    y = torch.empty_like(x)
    ctx.reserve(256, 16000)
    f = lambda: _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for _ in range(3):
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    print(f"C=512 unit, {batch} x {frames} frames: " + " ".join(f"{v:.4f}" for v in res) + f" ms  digest {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]}"
          f" finite {bool(torch.isfinite(y).all())}", flush=True)
