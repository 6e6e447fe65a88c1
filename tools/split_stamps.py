#!/usr/bin/env python3
"""Where a wave of gemm_split_kernel<.,16> spends its cycles (diagnostic build only):

    L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_SPLIT_STAMPS python -m l3ac_amd.build
    gpurun -- python tools/split_stamps.py [m n k]

Per k tile the kernel sums s_memtime differences of wave 0 over three phases: A wait + split (vector work), the 96 MFMAs
with their fragment reads, and the tail (wait for the next W tile, its LDS store, the block barrier).
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from l3ac_amd import _capi

lib = _capi.load_library()
shapes = [tuple(map(int, sys.argv[1:4]))] if len(sys.argv) > 3 else [(24480, 2048, 512), (24480, 512, 2048), (46080, 576, 128)]
s = torch.cuda.current_stream().cuda_stream
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    bias = torch.randn(n, device="cuda")
    c = torch.empty(m, n, device="cuda")
    img = torch.empty(lib.l3ac_gemm_split_image_bytes(n, k), dtype=torch.uint8, device="cuda")
    _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(), n, k, img.data_ptr(), s))
    f = lambda: _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(), k, img.data_ptr(), bias.data_ptr(), c.data_ptr(), n, m, n, k, s))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"m={m} n={n} k={k}: {ms:.4f} ms, {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
    fn = getattr(lib, "l3ac_debug_split_stamps", None)
    if fn is None:
        continue
    nblk = min(4096, -(-m // 128) * -(-n // 128))
    buf = np.zeros(8 * 4096, dtype=np.int64)
    fn.restype = C.c_int
    assert fn(buf.ctypes.data_as(C.c_void_p), 8 * 4096) == 0
    st = buf.reshape(4096, 8)[:nblk]
    med = np.median(st, axis=0)
    tiles = med[5]
    tot = med[3] + med[4]
    print(f"  per block (median of {nblk}): loop {med[3] / 1e3:.1f}k cycles over {int(tiles)} k tiles | epilogue {med[4] / 1e3:.1f}k")
    print(f"  per k tile: A wait + split {med[0] / tiles:.0f} | MFMA phase {med[1] / tiles:.0f} (96 x 16 = 1536 alone) | W store + barrier {med[2] / tiles:.0f}"
          f" | other {(med[3] - med[0] - med[1] - med[2]) / tiles:.0f}  -> shares {med[0] / tot:.2f} / {med[1] / tot:.2f} / {med[2] / tot:.2f}, epilogue {med[4] / tot:.2f}")
