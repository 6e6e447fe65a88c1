#!/usr/bin/env python3
"""The quantiser's stand-alone workload for tools/collect_profiles.sh (workload `fsq`): fsq_forward128_kernel and its copy ceiling at 2^22
tokens (1 052 algorithmic bytes per token), 1kbps levels — the launches bench.py's fsq_microbench times with HIP events, here under
rocprofv3 (kernel trace; FETCH_SIZE / WRITE_SIZE passes): counter bytes per launch against 4.41 GB algorithmic."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd
from l3ac_amd import _capi, weights as W

codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda").eval()
mc = codec.network.mc
d, feat, n = len(mc.levels), mc.feature_dim, 1 << 22
w = W.folded_weights(codec.network.state_dicts())
wt = {k: w[f"quantizer.{k}"].cuda() for k in ("project_in.weight", "project_in.bias", "project_out.weight", "project_out.bias")}
x = torch.randn(n, feat, device="cuda")
q = torch.empty_like(x)
idx = torch.empty(n, dtype=torch.int32, device="cuda")
li = torch.empty(n, d, device="cuda")
lib = _capi.load_library()
lv = (C.c_int32 * d)(*mc.levels)
s = torch.cuda.current_stream().cuda_stream
for _ in range(12):
    _capi.check(lib.l3ac_fsq_forward(x.data_ptr(), n, feat, lv, d, wt["project_in.weight"].data_ptr(), wt["project_in.bias"].data_ptr(),
                                     wt["project_out.weight"].data_ptr(), wt["project_out.bias"].data_ptr(), q.data_ptr(), idx.data_ptr(),
                                     li.data_ptr(), None, s))
    _capi.check(lib.l3ac_fsq_copy_ceiling(x.data_ptr(), n, q.data_ptr(), idx.data_ptr(), li.data_ptr(), s))
torch.cuda.synchronize()
print(f"fsq workload: {n} tokens x {8 * feat + 4 + 4 * d} B = {n * (8 * feat + 4 + 4 * d) / 1e9:.3f} GB algorithmic per launch")
