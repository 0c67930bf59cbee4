#!/usr/bin/env python3
"""Synchronous calls against two batches in flight (fz_query_topk_submit / _wait) on the C2 batch: wall time per step.
python tools/pipe_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import Corpus, synth  # noqa: E402
from tts_amd.preprocess import pack_prepared  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cps, offs = synth.make_corpus(1_000_000, max_chars=64)
q = pack_prepared(synth.make_queries(cps, offs, 1000, seed=synth.QUERY_SEED, max_chars=64))
with Corpus(cps, offs) as c:
    for _ in range(3):
        c.topk(q, 35, 10)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.topk(q, 35, 10)
    t_sync = (time.perf_counter() - t0) / steps * 1e3
    for depth1 in (True, False):
        c.topk_wait(c.topk_submit(q, 35, 10))
        t0 = time.perf_counter()
        prev = None
        ts = tw = 0.0
        for _ in range(steps):
            a = time.perf_counter()
            t = c.topk_submit(q, 35, 10)
            ts += time.perf_counter() - a
            if depth1:
                c.topk_wait(t)
                continue
            if prev is not None:
                a = time.perf_counter()
                c.topk_wait(prev)
                tw += time.perf_counter() - a
            prev = t
        print("  host time inside submit %.3f ms/step, inside wait %.3f ms/step" % (ts / steps * 1e3, tw / steps * 1e3))
        if prev is not None:
            c.topk_wait(prev)
        dt = (time.perf_counter() - t0) / steps * 1e3
        print("submit+wait %s: %.3f ms/step" % ("one at a time" if depth1 else "two in flight", dt))
    print("synchronous: %.3f ms/step" % t_sync)
