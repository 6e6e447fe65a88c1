"""Run in a FRESH process by tests/test_gpu_e2e.py::test_graph_capture_in_a_cold_process: the first use of every kernel of the
path happens under stream capture (include/l3ac_hip.h: after l3ac_reserve a call allocates nothing and can be captured — also when
nothing has run before, i.e. the per-device one-time kernel configuration must be legal under capture)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import l3ac_amd
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.cuda().eval()
audio = torch.randn(1, 16000, device="cuda") * 0.1
codec.network.context().reserve(1, 16000)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):   # FIRST use of every kernel happens under capture
        qf, ind = codec.encode_audio(audio)
        wav = codec.decode_audio(indices=ind["indices"])
g.replay()
torch.cuda.synchronize()
qf2, ind2 = codec.encode_audio(audio)
wav2 = codec.decode_audio(indices=ind2["indices"])
torch.cuda.synchronize()
print("captured without warm-up; tokens equal", bool((ind["indices"] == ind2["indices"]).all()), "wave equal", bool(torch.equal(wav, wav2)))
