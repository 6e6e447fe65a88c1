"""RCCL smoke test of the sharded path on ONE GPU: a single-rank `nccl` process group (backend "nccl" is RCCL on
ROCm) running l3ac_amd.dist.ShardedCodec with the real HIP codec.  Multi-rank logic is covered on CPU with gloo
(tests/test_dist_gloo.py); the 2/4/8-GPU scaling run is the driver's."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys
sys.path.insert(0, os.environ["L3AC_REPO"])
import torch, torch.distributed as dist
import l3ac_amd
from l3ac_amd.dist import PendingGathers, ShardedCodec, gather_batch, gather_batch_async
from tests.helpers import seeded_audio
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0)
codec.network.to(device="cuda:0").eval()
audio = seeded_audio(6, 8000).cuda()
idx, wave = ShardedCodec(codec).encode_decode(audio)
q, ind = codec.encode_audio(audio)
ref = codec.decode_audio(q)
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
out = torch.empty(6, 30, dtype=torch.int32, device="cuda")
dist.all_gather_into_tensor(out, ind["indices"])
ok = torch.equal(idx, ind["indices"]) and torch.equal(wave, ref) and torch.equal(out, ind["indices"]) and float(t.sum()) == 4.0
# the padded (ragged-batch) gather path and the bench's overlap loop, on RCCL
ok = ok and torch.equal(gather_batch(ref, 6, force_ragged=True), ref) and torch.equal(gather_batch(ind["indices"], 6, force_ragged=True), ind["indices"])
pend = PendingGathers()
for step in range(3):
    pend.push(gather_batch_async(ind["indices"] + step, 6, force=True), gather_batch_async(ref * (step + 1), 6, force=True))
pend.drain()
ok = ok and pend.retired == 3 and torch.equal(pend.results[0], ind["indices"] + 2) and torch.equal(pend.results[1], ref * 3)
dist.destroy_process_group()
print("RCCL_OK" if ok else "RCCL_MISMATCH")
"""


def test_single_rank_rccl_group():
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, L3AC_REPO=str(repo), MASTER_ADDR="127.0.0.1", MASTER_PORT="29531",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RCCL_OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
