#!/usr/bin/env python3
"""Wide ConvUnit (l3ac_op_conv_unit) at small batches: the fused kernel (option wide_sliced = 0) against the sliced form (= 2), per
kernel (library profile, eager) and per unit (a captured graph of 20 units, replayed), with the output digests of both forms.

    python tools/wide_sliced_sweep.py [batches ...]        (default 1 2 3 4 6 8 12 16)
"""
import hashlib
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

batches = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 12, 16]
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
lib = ctx.lib
torch.manual_seed(0)
side = torch.cuda.Stream()
for c, block, frames in ((256, "decoder.blocks.4.1.module", 900), (192, "encoder.blocks.7.0.module", 178)):
    for batch in batches:
        x = torch.randn(batch, frames, c, device="cuda")
        y = torch.empty_like(x)
        line = [f"C={c} B={batch:<3} tiles16={-(-batch * frames // 16):<5}"]
        digests = []
        for mode in (0, 2):
            if mode == 2 and -(-batch * frames // 16) > 256:
                line.append("sliced: -")
                continue
            ctx.set_option("wide_sliced", mode)
            with torch.cuda.stream(side):
                s = side.cuda_stream
                f = lambda: _capi.check(lib.l3ac_op_conv_unit(ctx.handle, block.encode(), x.data_ptr(), batch, frames, y.data_ptr(), s))
                for _ in range(3):
                    f()
                side.synchronize()
                with _capi.profile() as prof:
                    f()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(20):
                        f()
                for _ in range(3):
                    g.replay()
                side.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    g.replay()
                    e1.record(side)
                    side.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20)
            digests.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12])
            parts = " + ".join(f"{e['ms_total'] * 1e3:.1f}" for e in prof.entries)
            line.append(f"{'fused ' if mode == 0 else 'sliced'}: {sorted(ts)[len(ts) // 2] * 1e3:6.1f} us/unit in a graph (eager kernels {parts})")
        ctx.set_option("wide_sliced", 1)
        line.append("same bits" if len(set(digests)) == 1 else "DIFFERENT BITS" if len(digests) == 2 else "")
        print(" | ".join(line), flush=True)
