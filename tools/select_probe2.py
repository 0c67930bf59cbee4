import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import Corpus, synth
from tts_amd.preprocess import pack_prepared
cps, offs = synth.make_corpus(1_000_000, max_chars=64)
queries = synth.make_queries(cps, offs, 1000, max_chars=64)
c = Corpus(cps, offs)
for q in (["t6v6f0bf"], [queries[0]], queries[:100]):
    p = pack_prepared(q)
    c.topk(p, 35, 10); c.topk(p, 35, 10)
    print(len(q), c.timing()["select_ms"], flush=True)
