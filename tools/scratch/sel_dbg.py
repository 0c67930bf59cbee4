import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import Corpus, synth
cps, offs = synth.make_corpus(1_000_000, max_chars=64)
qs = synth.make_queries(cps, offs, 512)
with Corpus(cps, offs) as c:
    for nq in (1, 4, 8):
        for r in range(3):
            c.topk(qs[r * nq:r * nq + nq], 35, 10)
            print(nq, c.timing()["select_ms"], flush=True)
