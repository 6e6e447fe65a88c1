#!/usr/bin/env python3
"""bf16x3 GEMM at a single clip's row counts (the streaming chunk's shapes): microseconds per launch inside a replayed graph of
40 launches, for A/B runs of two builds on ONE box (L3AC_LIB_PATH), with an output digest per shape.

    python tools/gemm_small_m.py [rows]          (default 180)
"""
import hashlib
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from l3ac_amd import _capi

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 180
lib = _capi.load_library()
side = torch.cuda.Stream()
torch.manual_seed(0)
for n, k in ((512, 2048), (2048, 512), (256, 512), (192, 288), (512, 512)):
    a = torch.randn(rows, k, device="cuda")
    w = torch.randn(n, k, device="cuda") * 0.1
    bias = torch.randn(n, device="cuda")
    c = torch.empty(rows, n, device="cuda")
    img = torch.empty(lib.l3ac_gemm_split_image_bytes(n, k), dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(side):
        s = side.cuda_stream
        _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(), n, k, img.data_ptr(), s))
        f = lambda: _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(), k, img.data_ptr(), bias.data_ptr(), c.data_ptr(), n, rows, n, k, s))
        for _ in range(3):
            f()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(40):
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
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    digest = hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{rows} x {n} x {k}: {sorted(ts)[len(ts) // 2]:6.1f} us per launch (min {min(ts):.1f})  sha256 {digest}", flush=True)
