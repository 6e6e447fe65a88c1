#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM (l3ac_gemm_f32) on the shapes the 1kbps path launches (B=256)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from tests import gpu_ops as G

SHAPES = [  # (m, n, k, what)
    (4096, 4096, 4096, "square reference"),
    (46080, 2048, 512, "dec.1 pw_conv1"), (46080, 512, 2048, "dec.1 pw_conv2"),
    (230400, 1024, 256, "dec.4 pw_conv1"), (230400, 256, 1024, "dec.4 pw_conv2"),
    (691200, 384, 96, "dec.7 pw_conv1"), (691200, 96, 384, "dec.7 pw_conv2"),
    (2073600, 192, 48, "dec.10 pw_conv1"), (2073600, 48, 192, "dec.10 pw_conv2"),
    (4147200, 96, 24, "enc.1 pw_conv1"), (4147200, 24, 96, "enc.1 pw_conv2"),
    (46080, 768, 192, "enc.7 pw_conv1"), (46080, 192, 768, "enc.7 pw_conv2"),
    (46080, 576, 128, "to_qkv T=180"), (46080, 128, 344, "ff2 T=180"), (46080, 256, 512, "up0 1x1"),
]

if __name__ == "__main__":
    for m, n, k, what in SHAPES:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") * 0.05
        b = torch.randn(n, device="cuda")
        for _ in range(2):
            G.gemm(a, w, b)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        from l3ac_amd import _capi
        lib = _capi.load_library()
        c = torch.empty(m, n, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        ev0.record()
        for _ in range(reps):
            lib.l3ac_gemm_f32(a.data_ptr(), k, w.data_ptr(), b.data_ptr(), c.data_ptr(), n, m, n, k, s)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        print(f"{what:18s} m={m:8d} n={n:5d} k={k:5d}  {ms:8.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s  "
              f"{4.0 * (m * k + n * k + m * n) / ms / 1e6:8.1f} GB/s", flush=True)
        del a, w, b, c
