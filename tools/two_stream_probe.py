#!/usr/bin/env python3
"""VERDICT r3 item 2, measured from Python before anything is built into the library: one call of 256 clips on one stream against
the same 256 clips as two half-batches on two contexts (two workspaces) and two streams (include/l3ac_hip.h: "for true overlap use
one context per stream").  Prints the step time of both forms, several orderings of the two halves."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import l3ac_amd


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
    whole = l3ac_amd.get_model(cfg, synthetic_seed=0)
    whole.network.to(device=dev).eval()
    whole.network.context().reserve(256, 16000)
    one = lambda: whole.decode_audio(whole.encode_audio(audio)[0])
    t_one = timed(one)
    print(f"[{cfg}] one context, 256 clips, one stream: {t_one:.3f} ms/step")
    for parts in (2, 3, 4):
        n = 256 // parts
        codecs, streams = [], []
        for i in range(parts):
            c = l3ac_amd.get_model(cfg, synthetic_seed=0)
            c.network.to(device=dev).eval()
            c.network.context().reserve(-(-256 // parts) + 1, 16000)
            codecs.append(c)
            streams.append(torch.cuda.Stream())
        bounds = [round(i * 256 / parts) for i in range(parts + 1)]
        halves = [audio[bounds[i]:bounds[i + 1]].contiguous() for i in range(parts)]
        cur = torch.cuda.current_stream()

        def split_whole_calls():
            for s in streams:
                s.wait_stream(cur)
            outs = []
            for c, s, a in zip(codecs, streams, halves):
                with torch.cuda.stream(s):
                    outs.append(c.decode_audio(c.encode_audio(a)[0]))
            for s in streams:
                cur.wait_stream(s)
            return outs

        def split_interleaved():  # encode of every part first, then the decodes: the launch order alternates between the streams
            for s in streams:
                s.wait_stream(cur)
            qs = []
            for c, s, a in zip(codecs, streams, halves):
                with torch.cuda.stream(s):
                    qs.append(c.encode_audio(a)[0])
            outs = []
            for c, s, q in zip(codecs, streams, qs):
                with torch.cuda.stream(s):
                    outs.append(c.decode_audio(q))
            for s in streams:
                cur.wait_stream(s)
            return outs

        def sequential_parts():  # the same part sizes on ONE stream: what the smaller launches cost without any overlap
            return [c.decode_audio(c.encode_audio(a)[0]) for c, a in zip(codecs, halves)]

        t_a, t_b, t_c = timed(split_whole_calls), timed(split_interleaved), timed(sequential_parts)
        ref = one()
        got = torch.cat(split_whole_calls())
        torch.cuda.synchronize()
        print(f"[{cfg}] {parts} contexts x ~{n} clips on {parts} streams: encode+decode per part {t_a:.3f} ms/step, encodes then decodes "
              f"{t_b:.3f}; the same parts on one stream {t_c:.3f}; vs one call {t_one:.3f} ({(t_one - min(t_a, t_b)) / t_one * 100:+.1f} %); "
              f"outputs equal to the one-call batch: {bool(torch.equal(got, ref))}")
        del codecs
    print(f"[{cfg}] one context again: {timed(one):.3f} ms/step")
