"""small batches: wall time and event times of Corpus.topk by batch size (1e6 candidates)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tts_amd import Corpus, synth
n = int(os.environ.get("N", 1_000_000))
cps, offs = synth.make_corpus(n, max_chars=64)
qs = synth.make_queries(cps, offs, 512)
med = lambda a: float(np.median(a))
with Corpus(cps, offs) as c:
    for nq in (1, 2, 4, 8, 16, 32, 64):
        for r in range(5): c.topk(qs[:nq], 35, 10)
        wall, sc, se = [], [], []
        for r in range(30):
            b = qs[(r * nq) % 448:(r * nq) % 448 + nq]
            t0 = time.perf_counter(); c.topk(b, 35, 10); wall.append((time.perf_counter() - t0) * 1e3)
            t = c.timing(); sc.append(t["score_ms"]); se.append(t["select_ms"])
        print("nq %3d: p50 wall %.3f ms | score %.3f | select %.3f" % (nq, med(wall), med(sc), med(se)), flush=True)
