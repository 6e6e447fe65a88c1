#!/usr/bin/env python3
"""Explicit-codebook L2 argmin (l3ac_vq_argmin): time and VALU-roofline fraction at the sizes BASELINE.json names."""
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from l3ac_amd import _capi
from oracle import l3ac_oracle as O  # codebook table only (checker-side helper; nothing of the product imports it)

lib = _capi.load_library()
CASES = (([9, 9, 9, 7, 7, 7], 42752), ([7] * 6, 15360), ([7] * 6, 60), ([9, 9, 9, 7, 7, 7], 167))
QUICK = "--quick" in sys.argv  # first case, automatic form only (for counter runs)
for levels, n in (CASES[:1] if QUICK else CASES):
    k = 1
    for lv in levels:
        k *= lv
    q = torch.tanh(torch.randn(n, 6) * 1.2).cuda()
    cb = O.codebook(levels).cuda()
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for form in ((0,) if QUICK else (0, 1)):  # 0: automatic (screened form from 5 120 queries on), 1: the direct-form scan
        if form != 0 and n < 5120:
            continue
        nb = lib.l3ac_vq_argmin_scratch_bytes(n, k, form)
        sc = torch.zeros(max(nb, 4), dtype=torch.uint8, device="cuda")
        f = lambda: _capi.check(lib.l3ac_vq_argmin(q.data_ptr(), n, cb.data_ptr(), k, 6, idx.data_ptr(), sc.data_ptr(), nb, form, s))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        listed = int(sc[:4].view(torch.int32).item()) if (form != 1 and n >= 5120) else 0
        name = "wave" if n < 5120 else ("screened" if form == 0 else "scan")
        print(f"K={k} N={n} {name}: {ms:.3f} ms  {18.0 * n * k / ms / 1e9:.1f} TFLOP/s algorithmic ({18.0 * n * k / ms / 1e9 / 157.3:.3f} of the "
              f"fp32 VALU peak), {(28.0 * n + 24.0 * k) / ms / 1e6:.1f} GB/s algorithmic, listed for the full search: {listed}")
        with _capi.profile() as prof:
            f()
        print("    " + ", ".join(f"{e['name']} {e['ms_total'] * 1e3:.0f} us" for e in prof.entries))
        if form == 0:
            first = idx.clone()
        else:
            print(f"    forms agree on every query: {bool(torch.equal(first, idx))}")
