#!/usr/bin/env python3
"""Micro-benchmark of the fused FSQ kernel (closed form) and the explicit-codebook argmin at large N."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from l3ac_amd import _capi
from oracle import l3ac_oracle as O


def time_ms(fn, reps=10):
    for _ in range(3):
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
        w_in = (torch.rand(d, feat, generator=g) - 0.5).to(dev) * float(sys.argv[1] if len(sys.argv) > 1 else 0.2)
        b_in = torch.zeros(d, device=dev)
        w_out = (torch.rand(feat, d, generator=g) - 0.5).to(dev)
        b_out = torch.zeros(feat, device=dev)
        for n in (15360, 1 << 20, 1 << 22):
            x = torch.randn(n, feat, device=dev)
            q = torch.empty_like(x)
            idx = torch.empty(n, dtype=torch.int32, device=dev)
            li = torch.empty(n, d, device=dev)
            for with_li in (True, False):
                fn = lambda: lib.l3ac_fsq_forward(x.data_ptr(), n, feat, lv, d, w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(),
                                                  b_out.data_ptr(), q.data_ptr(), idx.data_ptr(), li.data_ptr() if with_li else None,
                                                  None, s)
                ms = time_ms(fn)
                bpt = 8 * feat + 4 + (4 * d if with_li else 0)
                print(f"fsq levels={levels} n={n:8d} level_indices={with_li!s:5s} {ms * 1e3:9.1f} us  {n * bpt / ms / 1e6:8.1f} GB/s "
                      f"({n * bpt / ms / 1e6 / 8000:.1%} of 8 TB/s)", flush=True)
        # explicit-codebook search (fp32-VALU bound: 18 N K FLOP)
        cb = O.codebook(levels).to(dev)
        n = 15360 if levels[0] == 7 else 42752
        qv = torch.tanh(torch.randn(n, d, device=dev))
        out = torch.empty(n, dtype=torch.int32, device=dev)
        fn = lambda: lib.l3ac_vq_argmin(qv.data_ptr(), n, cb.data_ptr(), cb.shape[0], d, out.data_ptr(), s)
        ms = time_ms(fn, reps=3)
        print(f"vq_argmin K={cb.shape[0]} N={n}: {ms:.3f} ms  {18.0 * n * cb.shape[0] / ms / 1e9:.1f} TFLOP/s "
              f"({18.0 * n * cb.shape[0] / ms / 1e9 / 157.3:.1%} of fp32 peak)", flush=True)
    # calibration: what a plain device-to-device copy of the same volume reaches (read + write counted)
    for mb in (512, 2048):
        a = torch.empty(mb * (1 << 20) // 4, device=dev)
        b = torch.empty_like(a)
        ms = time_ms(lambda: b.copy_(a))
        print(f"copy {mb} MiB: {2 * a.numel() * 4 / ms / 1e6:8.1f} GB/s ({2 * a.numel() * 4 / ms / 1e6 / 8000:.1%} of 8 TB/s)", flush=True)
