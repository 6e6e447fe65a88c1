#!/usr/bin/env python3
"""Per-kernel device time of ONE 1 s clip (B = 1, the streaming chunk) through encode_audio + decode_audio, eager."""
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import l3ac_amd
from l3ac_amd import _capi

name = sys.argv[1] if len(sys.argv) > 1 else "1kbps"
codec = l3ac_amd.get_model(name, synthetic_seed=0)
codec.network.cuda().eval()
g = torch.Generator().manual_seed(1234)
audio = ((torch.rand(1, 16000, generator=g) * 2 - 1) * 0.5).cuda()
for _ in range(3):
    q, ind = codec.encode_audio(audio)
    codec.decode_audio(indices=ind["indices"])
torch.cuda.synchronize()
with _capi.profile() as prof:
    q, ind = codec.encode_audio(audio)
    codec.decode_audio(indices=ind["indices"])
tot = sum(e["ms_total"] for e in prof.entries)
print(f"{name} B=1: {sum(e['launches'] for e in prof.entries)} launches, {tot * 1e3:.0f} us of kernel time")
for e in sorted(prof.entries, key=lambda e: -e["ms_total"]):
    print(f"  {e['name']:<48} x{e['launches']:<3} {e['ms_total'] * 1e3:8.1f} us  ({e['ms_total'] * 1e3 / e['launches']:.1f} each)")
