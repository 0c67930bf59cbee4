"""time of fz_distance_all kind 1 (Levenshtein) and kind 0 (InDel) on 10^5 candidates of <= 512 chars, 20 queries:
python tools/lev_ab.py   (FUZZYHIP_LIB selects the build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tts_amd import Corpus, synth
cps, offs = synth.make_corpus(100_000, max_chars=512, min_tokens=8, max_tokens=60, seed=5)
queries = synth.make_queries(cps, offs, 20, seed=6, max_chars=512)
with Corpus(cps, offs) as c:
    for kind in (1, 0):
        c.distance_all(queries[:2], kind)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); c.distance_all(queries, kind); ts.append((time.perf_counter() - t0) * 1e3)
        print("kind %d: %.2f ms wall, score %.2f ms (20 queries x 100000 candidates)" % (kind, np.median(ts), c.timing()["score_ms"]))
