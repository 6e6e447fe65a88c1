#!/usr/bin/env python3
"""Where a trans_stack_kernel workgroup spends its cycles: s_memtime sums per phase of wave 0 of workgroup 0, from a diagnostic
build (L3AC_BUILD_TAG=stamps L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_TS_STAMPS python -m l3ac_amd.build; run with
L3AC_LIB_PATH=l3ac_amd/libl3ac_hip_stamps.so).  usage: tools/ts_stamps.py [frames] [batch]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd
from l3ac_amd import _capi

PHASES = ["prologue", "LayerNorm 1", "q k v products", "attention", "out projection", "residual + LayerNorm 2", "FF-in products", "GEGLU",
          "FF-out products", "residual", "coop: partial stored", "coop: arrive + wait", "coop: partials read + added"]
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.cuda().eval()
ctx = codec.network.context()
lib = ctx.lib
block = b"en_decoder.local_trans" if frames <= 64 else b"en_decoder.up_trans.trans"
x = torch.randn(batch, frames, 128, device="cuda")
y = torch.empty_like(x)
call = lambda: _capi.check(lib.l3ac_op_local_trans(ctx.handle, block, x.data_ptr(), batch, frames, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    call()
torch.cuda.synchronize()
buf = (C.c_longlong * 16)()
lib.l3ac_debug_ts_stamps(buf, 16, 1)
reps = 10
for _ in range(reps):
    call()
torch.cuda.synchronize()
lib.l3ac_debug_ts_stamps(buf, 16, 1)
tot = sum(buf[i] for i in range(len(PHASES)))
print(f"{block.decode()} frames={frames} batch={batch}: {tot / reps:.0f} cycles per launch (wave 0 of workgroup 0)")
for i, name in enumerate(PHASES):
    print(f"  {name:<26} {buf[i] / reps:10.0f} cycles  {100.0 * buf[i] / tot:5.1f} %")
