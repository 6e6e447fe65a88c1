"""Helper of tests/test_gpu_blocks.py::test_gemm_split_w256_returns_the_same_bits (run in a subprocess per value of L3AC_GEMM_W256, which the
library reads once per process): digests of bf16x3 GEMM outputs on shapes that reach gemm_split_kernel_w256 — row counts that are not
multiples of its 192-row panels, both weight shapes of the C = 512 stage, a 256-column weight — and of the whole C = 512 ConvUnit (its first
product carries the snake + GRN epilogue, its second the residual), printed as one JSON line."""
import hashlib
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd
from l3ac_amd import _capi
from tests import gpu_ops as G


def sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]


out = {}
g = torch.Generator().manual_seed(7)
for m, n, k in ((24480 + 77, 512, 2048), (21600, 512, 1024), (46080 + 5, 256, 512), (33000, 2048, 512), (193 * 150, 768, 64)):
    a = torch.randn(m, k, generator=g) * torch.pow(10.0, torch.randint(-2, 2, (m, 1), generator=g).float())
    w = torch.randn(n, k, generator=g) * 0.1
    b = torch.randn(n, generator=g)
    out[f"gemm {m}x{n}x{k}"] = sha(G.gemm_split(a.cuda(), w.cuda(), b.cuda()))
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
ctx = codec.network.context()
ctx.reserve(256, 16000)
s = torch.cuda.current_stream().cuda_stream
for batch, frames in ((136, 180), (131, 97)):
    x = torch.randn(batch, frames, 512, generator=g).cuda()
    y = torch.empty_like(x)
    _capi.check(ctx.lib.l3ac_op_conv_unit(ctx.handle, b"decoder.blocks.1.2.module", x.data_ptr(), batch, frames, y.data_ptr(), s))
    torch.cuda.synchronize()
    out[f"unit512 {batch}x{frames}"] = sha(y)
print(json.dumps(out))
