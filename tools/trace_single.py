"""kernel timeline of single-query calls (bench.py's p50 workload): python tools/trace_single.py [n_calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import Corpus, synth  # noqa: E402
from tts_amd.preprocess import pack_prepared  # noqa: E402

cps, offs = synth.make_corpus(1_000_000, max_chars=64)
queries = synth.make_queries(cps, offs, 16)
with Corpus(cps, offs) as c:
    c.topk(pack_prepared(queries), 35, 10)
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        c.topk(pack_prepared([queries[i % 16]]), 35, 10)
