"""Where does fz_select_topk spend its time?  select_ms for growing query counts and the slowest single rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tts_amd import Corpus, synth
from tts_amd.preprocess import pack_prepared
n = 1_000_000
cps, offs = synth.make_corpus(n, max_chars=64)
queries = synth.make_queries(cps, offs, 1000, max_chars=64)
c = Corpus(cps, offs)
for nq in (1, 10, 100, 250, 500, 1000):
    q = pack_prepared(queries[:nq])
    c.topk(q, 35, 10)
    c.topk(q, 35, 10)
    t = c.timing()
    print(nq, "score %.3f select %.3f" % (t["score_ms"], t["select_ms"]), flush=True)
per = []
for i in range(0, 1000):
    q = pack_prepared(queries[i:i + 1])
    c.topk(q, 35, 10)
    per.append(c.timing()["select_ms"])
per = np.array(per)
order = np.argsort(-per)[:10]
print("single-row select ms: median %.4f  p99 %.4f  max %.4f" % (np.median(per), np.percentile(per, 99), per.max()))
for i in order:
    print("%.4f  %r" % (per[i], queries[i]))
