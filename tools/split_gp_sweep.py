#!/usr/bin/env python3
"""gemm_split on the two C = 512 shapes (46080 x 2048 x 512, 46080 x 512 x 2048): time per launch; run once per L3AC_SPLIT_GP value
(the tile-order group size is read once per process)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from l3ac_amd import _capi
lib = _capi.load_library()
s = torch.cuda.current_stream().cuda_stream
for m, n, k in ((46080, 2048, 512), (46080, 512, 2048), (230400, 256, 512)):
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    nbytes = lib.l3ac_gemm_split_image_bytes(n, k)
    img = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); c = torch.empty(m, n, device="cuda")
    _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(), n, k, img.data_ptr(), s))
    call = lambda: _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(), k, img.data_ptr(), b.data_ptr(), c.data_ptr(), n, m, n, k, s))
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"GP={os.environ.get('L3AC_SPLIT_GP', 'default(8)')} m={m} n={n} k={k}: {ms*1e3:.1f} us {2.0*m*n*k/ms/1e9:.1f} TFLOP/s")
