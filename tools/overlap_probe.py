#!/usr/bin/env python3
"""do complementary kernels overlap?  Every batch kernel of the 256-clip step leaves about half of every resource idle
(matrix pipe 0.37-0.66 busy, vector issue 0.34-0.74, HBM at a quarter of its rate).  Measured here from Python, no new kernel:

  one        one context, 256 clips, one stream (the bench's step)
  halves     two contexts x 128 clips on two streams, started together (round 4's probe: + 0.5 %)
  staggered  the same halves, half B's encoder started when half A's encoder has finished (decoder of A beside encoder of B)
  pipelined  ONE batch of 256 per step, its encoder on stream 1 and the decoder of the PREVIOUS step's tokens on stream 2 (two contexts:
             two workspaces); steady-state time per step
  two_full   two contexts x 256 clips free-running on two streams; time per 256 clips (upper bound of what overlap can give)

Two interleaved rounds of every form; outputs of the split forms are compared with the one-call batch."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd


def make(cfg, dev, clips):
    c = l3ac_amd.get_model(cfg, synthetic_seed=0)
    c.network.to(device=dev).eval()
    c.network.context().reserve(clips, 16000)
    return c


def timed(fn, steps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "1kbps"
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(1234)
    audio = ((torch.rand(256, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    whole, whole2 = make(cfg, dev, 256), make(cfg, dev, 256)
    ha, hb = make(cfg, dev, 129), make(cfg, dev, 129)
    a_half, b_half = audio[:128].contiguous(), audio[128:].contiguous()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()

    def one():
        return whole.decode_audio(whole.encode_audio(audio)[0])

    def halves():
        s1.wait_stream(cur), s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            oa = ha.decode_audio(ha.encode_audio(a_half)[0])
        with torch.cuda.stream(s2):
            ob = hb.decode_audio(hb.encode_audio(b_half)[0])
        cur.wait_stream(s1), cur.wait_stream(s2)
        return oa, ob

    def staggered():
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            qa = ha.encode_audio(a_half)[0]
            enc_a_done = torch.cuda.Event()
            enc_a_done.record(s1)
            oa = ha.decode_audio(qa)
        s2.wait_stream(cur)
        s2.wait_event(enc_a_done)
        with torch.cuda.stream(s2):
            ob = hb.decode_audio(hb.encode_audio(b_half)[0])
        cur.wait_stream(s1), cur.wait_stream(s2)
        return oa, ob

    q_prev = [whole.encode_audio(audio)[0]]

    def pipelined():  # encoder of this step's batch beside the decoder of the previous step's tokens
        s1.wait_stream(cur), s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            out = whole2.decode_audio(q_prev[0])
        with torch.cuda.stream(s1):
            q_prev[0] = whole.encode_audio(audio)[0]
        cur.wait_stream(s1), cur.wait_stream(s2)
        return out

    def two_full():  # two whole steps side by side: the time of the pair, halved below
        s1.wait_stream(cur), s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            o1 = whole.decode_audio(whole.encode_audio(audio)[0])
        with torch.cuda.stream(s2):
            o2 = whole2.decode_audio(whole2.encode_audio(audio)[0])
        cur.wait_stream(s1), cur.wait_stream(s2)
        return o1, o2

    def enc_only():
        return whole.encode_audio(audio)[0]

    def dec_only():
        return whole2.decode_audio(q_prev[0])

    ref = one()
    torch.cuda.synchronize()
    oa, ob = halves()
    torch.cuda.synchronize()
    print(f"[{cfg}] halves == one call: {bool(torch.equal(torch.cat([oa, ob]), ref))}", flush=True)
    oa, ob = staggered()
    torch.cuda.synchronize()
    print(f"[{cfg}] staggered == one call: {bool(torch.equal(torch.cat([oa, ob]), ref))}", flush=True)
    pipelined()
    out = pipelined()
    torch.cuda.synchronize()
    print(f"[{cfg}] pipelined == one call: {bool(torch.equal(out, ref))}", flush=True)
    for rnd in range(2):
        t_one = timed(one)
        t_h = timed(halves)
        t_s = timed(staggered)
        t_p = timed(pipelined)
        t_2 = timed(two_full) / 2
        t_e, t_d = timed(enc_only), timed(dec_only)
        print(f"[{cfg}] round {rnd}: one {t_one:.3f} ms | halves {t_h:.3f} ({(t_one / t_h - 1) * 100:+.1f} %) | staggered {t_s:.3f} "
              f"({(t_one / t_s - 1) * 100:+.1f} %) | pipelined {t_p:.3f} ({(t_one / t_p - 1) * 100:+.1f} %) | two whole steps side by side "
              f"{t_2:.3f} per 256 clips ({(t_one / t_2 - 1) * 100:+.1f} %) | encoder alone {t_e:.3f}, decoder alone {t_d:.3f}", flush=True)
