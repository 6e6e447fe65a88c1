"""How the oracle (PyTorch-CPU restatement) scales with host threads on the GPU box: picks the cpu_baseline setting."""
import sys, time
sys.path.insert(0, '.')
import torch
from l3ac_amd import weights as W
from l3ac_amd.config import L3ACConfig, resolve_config_file
from oracle import l3ac_oracle as O
from tests.helpers import seeded_audio
mc = L3ACConfig(config_file=resolve_config_file("1kbps")).network_config
w = W.folded_weights(W.synthetic_state_dicts(mc, seed=0))
x = seeded_audio(8, 16000)
for th in (128, 64, 32, 16, 8):
    torch.set_num_threads(th)
    O.decode_audio(w, mc, O.encode_audio(w, mc, x)[0])
    t0 = time.perf_counter()
    O.decode_audio(w, mc, O.encode_audio(w, mc, x)[0])
    dt = time.perf_counter() - t0
    print(f"threads={th:4d}  B=8: {dt:.2f} s  {8 * 16000 / dt / 1e3:.1f} k samples/s", flush=True)
