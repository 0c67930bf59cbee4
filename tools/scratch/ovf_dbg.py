import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tts_amd import Corpus, synth
from tts_amd.preprocess import pack_prepared
n = 12500032
per_block = n // 64
cps, offs = synth.make_corpus_blocks(0, 64, per_block, max_chars=64)
queries = synth.make_queries(cps, offs, 2000, seed=synth.QUERY_SEED, max_chars=64)
with Corpus(cps, offs) as c:
    def ovf(qs):
        c.topk(qs, 35, 10)
        t = c.timing()
        return t["fused"], t["fused_fallback_chunks"]
    print("all", ovf(queries), flush=True)
    lo, hi = 0, len(queries)
    while hi - lo > 8:
        mid = (lo + hi) // 2
        f, o = ovf(queries[lo:mid])
        if o: hi = mid
        else: lo = mid
        print(lo, hi, f, o, flush=True)
    for i in range(lo, hi):
        idx, sc, cnt = c.topk([queries[i]] * 8, 35, 10)
        t = c.timing()
        print(i, repr(queries[i]), t["fused"], t["fused_fallback_chunks"], sc[0].tolist(), idx[0].tolist(), flush=True)
