"""single-query latency breakdown: wall time of Corpus.topk vs the hipEvent times the library reports"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tts_amd import Corpus, synth
n = 1_000_000
cps, offs = synth.make_corpus(n, max_chars=64)
qs = synth.make_queries(cps, offs, 200)
with Corpus(cps, offs) as c:
    for q in qs[:20]: c.topk([q], 35, 10)
    wall, tot, sc, se, prep = [], [], [], [], []
    for q in qs[20:120]:
        t0 = time.perf_counter(); c.topk([q], 35, 10); wall.append((time.perf_counter() - t0) * 1e3)
        t = c.timing(); tot.append(t["total_ms"]); sc.append(t["score_ms"]); se.append(t["select_ms"]); prep.append(t["host_prep_ms"])
    med = lambda a: float(np.median(a))
    print("p50 wall %.3f ms | event total %.3f | score %.3f | select %.3f | host prep %.3f" % (med(wall), med(tot), med(sc), med(se), med(prep)))
