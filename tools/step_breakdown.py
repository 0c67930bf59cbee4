import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np
from tts_amd import Corpus, synth
from tts_amd.preprocess import pack_prepared
cps, offs = synth.make_corpus(1_000_000, max_chars=64)
queries = synth.make_queries(cps, offs, 1000)
qp = pack_prepared(queries)
with Corpus(cps, offs) as c:
    for _ in range(3): c.topk(qp, 35, 10)
    rows = []
    for _ in range(20):
        t0 = time.perf_counter(); c.topk(qp, 35, 10); w = (time.perf_counter() - t0) * 1e3
        t = c.timing(); rows.append((w, t["total_ms"], t["score_ms"], t["select_ms"], t["host_prep_ms"]))
    a = np.median(np.array(rows), axis=0)
    print("wall %.3f  gpu first->last event %.3f  score %.3f  select %.3f  host prep %.3f  => wall - gpu span %.3f, gpu span - kernels %.3f" % (a[0], a[1], a[2], a[3], a[4], a[0]-a[1], a[1]-a[2]-a[3]))
