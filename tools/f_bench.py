#!/usr/bin/env python3
"""p50 latency per query of the three drop-in classes over 10^5 candidates (SURVEY.md §8 rows a/f1/f2):
FuzzySearch.find_best_matches (src/fuzzy_search.py:16-38), WeightedFuzzySearch.find_best_matches
(app.py:981-1002) and AdvancedFuzzyMatcher.fuzzy_search (app.py:378-460).  Corpus resident, the first
call of every class (view build, index build) excluded.  Prints one JSON object.

    python tools/f_bench.py [--candidates 100000] [--queries 30] > profiles/r02_f1f2_latency.json"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tts_amd import FuzzySearch, synth  # noqa: E402
from tts_amd.advanced_matcher import AdvancedFuzzyMatcher  # noqa: E402
from tts_amd.weighted_search import WeightedFuzzySearch  # noqa: E402


def p50(fn, queries, reps):
    # what first calls build (device views per scorer, the AdvancedFuzzyMatcher index) is built OUTSIDE the timed calls:
    # one query of the batch, and one that matches nothing exactly (an exact hit returns before any index is made)
    fn(queries[0])
    fn("zzqq warm up 0")
    fn(queries[len(queries) // 2])
    per_query = []
    for q in queries:
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn(q)
            ts.append((time.perf_counter() - t0) * 1e3)
        per_query.append(statistics.median(ts))
    per_query.sort()
    return {"p50_ms": round(statistics.median(per_query), 4), "min_ms": round(per_query[0], 4),
            "max_ms": round(per_query[-1], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=30)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    cps, offs = synth.make_corpus(a.candidates, max_chars=64)
    names = synth.to_strings(cps, offs)
    # mixed case + a pack size so that the weighted and the advanced class do their real preprocessing
    products = [(n.title() + (" 330ml" if i % 7 == 0 else "")) for i, n in enumerate(names)]
    queries = [q.upper() if i % 3 == 0 else q for i, q in enumerate(synth.make_queries(cps, offs, a.queries))]
    out = {"candidates": a.candidates, "queries": a.queries, "reps_per_query": a.reps,
           "what": "median over queries of the median wall time of one call, corpus resident, Python call included"}

    fs = FuzzySearch(products, [], threshold=0.35)
    out["FuzzySearch.find_best_matches"] = p50(lambda q: fs.find_best_matches(q, "product"), queries, a.reps)
    fs.close()

    ws = WeightedFuzzySearch(products, [], threshold=0.70)
    out["WeightedFuzzySearch.find_best_matches"] = p50(lambda q: ws.find_best_matches(q, "product"), queries, a.reps)
    ws.close()

    out["AdvancedFuzzyMatcher.fuzzy_search"] = p50(lambda q: AdvancedFuzzyMatcher.fuzzy_search(q, products), queries, a.reps)
    AdvancedFuzzyMatcher.clear_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
