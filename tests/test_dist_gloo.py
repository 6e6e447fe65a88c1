"""The N > 1 path (batch sharding + output all-gather) on CPU: world size 2, gloo backend, 127.0.0.1.
The per-rank compute is the oracle standing in for the HIP codec (tests may use it); what is under test is
l3ac_amd/dist.py: shard ranges, equal and ragged gathers, and that sharded == unsharded results."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from l3ac_amd.dist import PendingGathers, ShardedCodec, gather_batch, gather_batch_async, shard_range


def test_shard_ranges_partition_the_batch():
    for total in (1, 2, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


class _OracleCodec:
    """Stand-in with the product's method surface, computing with the oracle on CPU (test only)."""

    def __init__(self):
        from tests.helpers import load_case
        self.mc, self.w, _, _ = load_case("tiny")

    def encode_audio(self, audio):
        from oracle import l3ac_oracle as O
        return O.encode_audio(self.w, self.mc, audio)

    def decode_audio(self, q_feature=None, indices=None):
        from oracle import l3ac_oracle as O
        return O.decode_audio(self.w, self.mc, audio_feature=q_feature, indices=indices)


def _worker(rank, world, port, total, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.helpers import seeded_audio
        codec = _OracleCodec()
        audio = seeded_audio(total, 300)
        idx, wave = ShardedCodec(codec).encode_decode(audio)
        ref_q, ref_ind = codec.encode_audio(audio)
        ref_wave = codec.decode_audio(ref_q)
        ok = torch.equal(idx, ref_ind["indices"]) and torch.allclose(wave, ref_wave, atol=1e-6)
        # plain gather helper, int and float, equal and ragged
        start, stop = shard_range(total, rank, world)
        full = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3)
        ok = ok and torch.equal(gather_batch(full[start:stop], total), full)
        # non-blocking variant (what bench.py overlaps with the next step): two in flight, retired out of order
        h1 = gather_batch_async(full[start:stop], total)
        h2 = gather_batch_async((full * 2)[start:stop].to(torch.int64), total)
        ok = ok and torch.equal(h2.wait(), (full * 2).to(torch.int64)) and torch.equal(h1.wait(), full)
        ok = ok and torch.equal(h1.wait(), full)  # waiting twice is harmless
        # bench.py's overlap loop: every step pushes its two gathers and retires the previous step's pair; drain at the end
        if total % world == 0:
            pend = PendingGathers()
            for step in range(4):
                pend.push(gather_batch_async(full[start:stop] + step, total), gather_batch_async((full[start:stop] * step).to(torch.int32), total))
                ok = ok and len(pend) == 1 and pend.retired == step
                if step:
                    ok = ok and torch.equal(pend.results[0], full + (step - 1)) and torch.equal(pend.results[1], (full * (step - 1)).to(torch.int32))
            pend.drain()
            ok = ok and len(pend) == 0 and pend.retired == 4 and torch.equal(pend.results[0], full + 3)
        queue.put((rank, bool(ok), tuple(idx.shape), tuple(wave.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_sharded_equals_unsharded_world2(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, idx_shape, wave_shape in results:
        assert ok, f"rank {rank}: sharded result differs from the unsharded one"
        assert idx_shape[0] == total and wave_shape[0] == total
