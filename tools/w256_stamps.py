#!/usr/bin/env python3
"""Where a workgroup of gemm_split_kernel_w256 spends its cycles (diagnostic build only):

    L3AC_BUILD_TAG=stamps L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_W256_STAMPS python -m l3ac_amd.build
    gpurun -- 'L3AC_LIB_PATH=l3ac_amd/libl3ac_hip_stamps.so python tools/w256_stamps.py 24480x2048x512 24480x512x2048'

Stamps (s_memtime, wave 0 of every workgroup): start | loop entry | loop exit | end."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from l3ac_amd import _capi

lib = _capi.load_library()
s = torch.cuda.current_stream().cuda_stream
for arg in sys.argv[1:]:
    m, n, k = (int(v) for v in arg.split("x"))
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); bias = torch.randn(n, device="cuda"); c = torch.empty(m, n, device="cuda")
    img = torch.empty(lib.l3ac_gemm_split_image_bytes(n, k), dtype=torch.uint8, device="cuda")
    _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(), n, k, img.data_ptr(), s))
    f = lambda: _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(), k, img.data_ptr(), bias.data_ptr(), c.data_ptr(), n, m, n, k, s))
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    NB = 4096
    buf = np.zeros(NB * 4, dtype=np.uint64)
    fn = lib.l3ac_debug_w256_stamps
    fn.restype = C.c_int
    assert fn(buf.ctypes.data_as(C.c_void_p), NB * 4) == 0
    grid = -(-(-(-m // 192)) // 8) * 8 * (n // 256)
    st = buf.reshape(NB, 4).astype(np.int64)[:grid]
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    d = np.diff(st, axis=1)
    print(f"{m}x{n}x{k}: {len(st)} workgroups, k tiles {k // 32}; medians: prologue {np.median(d[:, 0]) / 1e3:.1f}k | loop {np.median(d[:, 1]) / 1e3:.1f}k "
          f"= {np.median(d[:, 1]) / (k // 32):.0f} per k tile (MFMA floor 4608) | epilogue {np.median(d[:, 2]) / 1e3:.1f}k | kernel span {(st[:, 3].max() - t0) / 1e3:.0f}k cycles; "
          f"start times (k cycles after the first): p25 {np.percentile(st[:, 0] - t0, 25) / 1e3:.0f} p50 {np.percentile(st[:, 0] - t0, 50) / 1e3:.0f} p75 {np.percentile(st[:, 0] - t0, 75) / 1e3:.0f} max {(st[:, 0] - t0).max() / 1e3:.0f}")
