"""The oracle against the golden vectors, the hand-derived known answers and itself (CPU only).

PARITY UNPINNED (see oracle/fuzz_oracle.c): the reference holds no vectors and its scorer library is
not installed; kats.json carries the seven answers derivable by hand from the published formula."""
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SCORERS = {"token_set": 0, "ratio": 1, "token_sort": 2, "partial": 3}


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def test_known_answers(oracle):
    for k in _load("kats.json"):
        assert oracle.score(SCORERS[k["scorer"]], k["a"], k["b"]) == k["score"], k


def test_golden_pairs(oracle):
    for r in _load("pairs.json"):
        q, c = r["q"], r["c"]
        assert oracle.score(oracle.TOKEN_SET, q, c) == r["token_set"], r
        assert oracle.score(oracle.RATIO, q, c) == r["ratio"], r
        assert oracle.score(oracle.TOKEN_SORT, q, c) == r["token_sort"], r
        assert oracle.indel(q, c) == r["indel"], r
        assert oracle.levenshtein(q, c) == r["levenshtein"], r
        if r["partial"] is not None and len(q) * len(c) < 4000:
            assert oracle.score(oracle.PARTIAL, q, c) == r["partial"], r


def test_c_equals_pure_python_restatement(oracle):
    """two independent restatements of token_set_ratio agree (C DP vs pure Python)"""
    for r in _load("pairs.json")[:400]:
        if len(r["q"]) * len(r["c"]) > 6000:
            continue
        assert int(round(oracle.py_token_set_ratio(r["q"], r["c"]))) == r["token_set"], r


def test_bitparallel_lcs_equals_dp(oracle):
    rnd = random.Random(5)
    for _ in range(300):
        a = "".join(rnd.choice("abcd ") for _ in range(rnd.randint(0, 200)))
        b = "".join(rnd.choice("abcde") for _ in range(rnd.randint(0, 200)))
        assert oracle.lcs_bitpar(a, b) == oracle.lcs_dp(a, b)
    for n in (63, 64, 65, 127, 128, 129):
        a = "".join(rnd.choice("ab") for _ in range(n))
        b = "".join(rnd.choice("ab") for _ in range(n + 3))
        assert oracle.lcs_bitpar(a, b) == oracle.lcs_dp(a, b)
    oracle.lib().orc_set_bitpar(1)
    try:
        for r in _load("pairs.json")[:300]:
            assert oracle.score(oracle.TOKEN_SET, r["q"], r["c"]) == r["token_set"]
    finally:
        oracle.lib().orc_set_bitpar(0)


def test_distance_properties(oracle):
    rnd = random.Random(9)
    for _ in range(200):
        a = "".join(rnd.choice("abc") for _ in range(rnd.randint(0, 30)))
        b = "".join(rnd.choice("abc") for _ in range(rnd.randint(0, 30)))
        d, l = oracle.indel(a, b), oracle.levenshtein(a, b)
        assert d == oracle.indel(b, a) and l == oracle.levenshtein(b, a)
        assert l <= d <= 2 * l                       # substitution = 1 vs 2
        assert (d - len(a) - len(b)) % 2 == 0
        assert oracle.indel(a, a) == 0 and oracle.levenshtein(a, "") == len(a)


def test_rounding_ties_follow_python_round(oracle):
    L = oracle.lib()
    for lensum in range(1, 600):
        for dist in range(0, lensum + 1):
            x = 100.0 - 100.0 * dist / lensum
            assert L.orc_norm_distance_100(dist, lensum) == x
            assert L.orc_py_round(x) == int(round(x))
            y = (1.0 - dist / lensum) * 100
            assert L.orc_norm_sim_100(dist, lensum) == y
    assert int(round(87.5)) == 88 and int(round(62.5)) == 62 and int(round(98.5)) == 98


def test_topk_is_stable_descending(oracle):
    cand = ["a b", "a", "b", "a b", "a c", "c", "a", "zz"]
    buf, offs = oracle.pack(cand)
    got = oracle.topk(oracle.TOKEN_SET, "a", buf, offs, 35, 5)
    scores = [oracle.score(oracle.TOKEN_SET, "a", c) for c in cand]
    order = sorted(range(len(cand)), key=lambda i: -scores[i])      # Python's sort is stable
    exp = [(i, scores[i]) for i in order if scores[i] >= 35][:5]
    assert got == exp
    assert oracle.c_find_best_matches(cand, "A", 0.35, 5) == oracle.py_find_best_matches(cand, "A", 0.35, 5)


def test_min_score_for_threshold(oracle):
    for thr in (0.0, 0.35, 0.29, 0.57, 0.58, 0.7, 0.8, 0.99, 1.0, 1.01, -1.0, 0.07, 0.14, 0.28):
        ms = oracle.min_score_for_threshold(thr)
        for s in range(0, 101):
            assert (s / 100.0 >= thr) == (s >= ms)
