#!/usr/bin/env python3
"""The streaming chunk alone (bench.py's stream_1s_graph protocol: B = 1, one captured hipGraph of encode_audio + decode_audio(indices),
600 chunks of a resident 10-minute clip, tokens and waveform kept), three timed passes: ms per chunk.  L3AC_LIB_PATH selects a build.

    python tools/chunk_time.py [config]
"""
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import l3ac_amd

name = sys.argv[1] if len(sys.argv) > 1 else "1kbps"
dev = torch.device("cuda:0")
codec1 = l3ac_amd.get_model(name, synthetic_seed=0)
codec1.network.to(device=dev).eval()
codec1.network.context().reserve(1, 16000)
g = torch.Generator(device="cpu").manual_seed(99)
chunks = ((torch.rand(600, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
static_in = torch.zeros(1, 16000, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    codec1.decode_audio(indices=codec1.encode_audio(static_in)[1]["indices"])
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    q1, ind1 = codec1.encode_audio(static_in)
    wave1 = codec1.decode_audio(indices=ind1["indices"])
tokens = torch.empty(600, ind1["indices"].shape[1], dtype=torch.int32, device=dev)
waves = torch.empty(600, wave1.shape[1], device=dev)


def replay_all():
    for i in range(600):
        static_in.copy_(chunks[i:i + 1])
        graph.replay()
        tokens[i].copy_(ind1["indices"][0])
        waves[i].copy_(wave1[0])


replay_all()
torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter()
    replay_all()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 600 * 1e3)
eager = codec1.encode_audio(chunks[599:600])[1]["indices"]
import hashlib
print(f"{name} chunk: " + " ".join(f"{r:.4f}" for r in res) + f" ms; tokens == eager: {bool(torch.equal(tokens[599:600], eager))}; "
      f"tokens sha {hashlib.sha256(tokens.cpu().numpy().tobytes()).hexdigest()[:12]} waves sha {hashlib.sha256(waves.cpu().numpy().tobytes()).hexdigest()[:12]}")
