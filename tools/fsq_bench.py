#!/usr/bin/env python3
"""Micro-benchmark of the fused FSQ kernel (closed form) at large N, next to its own copy ceiling (same grid and per-lane
accesses, no arithmetic) and a plain device copy."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from l3ac_amd import _capi


def time_ms(fn, reps=20, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if __name__ == "__main__":
    lib = _capi.load_library()
    dev = torch.device("cuda")
    s = torch.cuda.current_stream().cuda_stream
    for levels in ([7] * 6, [9, 9, 9, 7, 7, 7]):
        d, feat = len(levels), 128
        lv = (C.c_int32 * d)(*levels)
        g = torch.Generator().manual_seed(0)
        w_in = (torch.rand(d, feat, generator=g) - 0.5).to(dev) * 0.2
        b_in = torch.zeros(d, device=dev)
        w_out = (torch.rand(feat, d, generator=g) - 0.5).to(dev)
        b_out = torch.zeros(feat, device=dev)
        for n in (15360, 1 << 20, 1 << 22):
            x = torch.randn(n, feat, device=dev)
            q = torch.empty_like(x)
            idx = torch.empty(n, dtype=torch.int32, device=dev)
            li = torch.empty(n, d, device=dev)
            bpt = 8 * feat + 4 + 4 * d
            fn = lambda: lib.l3ac_fsq_forward(x.data_ptr(), n, feat, lv, d, w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(),
                                              b_out.data_ptr(), q.data_ptr(), idx.data_ptr(), li.data_ptr(), None, s)
            cp = lambda: lib.l3ac_fsq_copy_ceiling(x.data_ptr(), n, q.data_ptr(), idx.data_ptr(), li.data_ptr(), s)
            for rnd in range(2):  # interleaved rounds in one process
                ms, cms = time_ms(fn), time_ms(cp)
                print(f"fsq levels={levels} n={n:8d} {ms * 1e3:9.1f} us {n * bpt / ms / 1e6:8.1f} GB/s ({n * bpt / ms / 1e6 / 8000:.1%} of 8 TB/s) | "
                      f"copy ceiling {cms * 1e3:9.1f} us {n * bpt / cms / 1e6:8.1f} GB/s ({n * bpt / cms / 1e6 / 8000:.1%}) | ratio {cms / ms:.3f}", flush=True)
    for mb in (512, 2048):  # calibration: a plain device-to-device copy of a similar volume (read + write counted)
        a = torch.empty(mb * (1 << 20) // 4, device=dev)
        b = torch.empty_like(a)
        ms = time_ms(lambda: b.copy_(a))
        print(f"hipMemcpy-style copy {mb} MiB: {2 * a.numel() * 4 / ms / 1e6:8.1f} GB/s ({2 * a.numel() * 4 / ms / 1e6 / 8000:.1%} of 8 TB/s)", flush=True)
