"""GPU parity: libfuzzyhip.so (through the C ABI) against the CPU oracle, bit-exact.

Every test here needs an MI355X; the oracle is the checker only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tts_amd import Corpus, FuzzySearch, _lib, synth  # noqa: E402
from tts_amd.preprocess import min_score_for_threshold, prepare  # noqa: E402


def _oracle_scores(oracle, queries, cand):
    buf, offs = oracle.pack(cand)
    return np.stack([oracle.score_all(oracle.TOKEN_SET, q, buf, offs) for q in queries]).astype(np.uint8)


def _check_topk(oracle, corpus, queries, cand, min_score, k):
    idx, score, cnt = corpus.topk(queries, min_score, k)
    buf, offs = oracle.pack(cand)
    for qi, q in enumerate(queries):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf, offs, min_score, k)
        got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
        assert got == exp, (qi, q, got[:5], exp[:5])


def test_score_matrix_synthetic_short(oracle):
    cps, offs = synth.make_corpus(3000, max_chars=64, seed=1)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 40, seed=2) + ["", "zz", cand[5], cand[7] + " extra tok"]
    with Corpus(cps, offs) as c:
        got = c.score_all(queries)
    exp = _oracle_scores(oracle, queries, cand)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]


def test_topk_order_under_ties(oracle):
    # tiny vocabulary => massive score ties; order must be (score desc, original index asc)
    cps, offs = synth.make_corpus(5000, max_chars=32, seed=3, vocab_size=12)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 12, seed=4, max_chars=32) + [cand[0].split()[0]]
    with Corpus(cps, offs) as c:
        for k, ms in ((10, 35), (1, 0), (64, 90), (5, 100), (1024, 50)):
            _check_topk(oracle, c, queries, cand, ms, k)


def test_edge_cases(oracle):
    cand = ["", "a", "a a a", "b a", "a b", "   ", "abc def", "def abc", "abc", "x" * 70, "ab " * 30,
            "coke 330ml 1x24 np sleek can", "coke sleek can", "fuzzy fuzzy was a bear"]
    queries = ["", "a", "a b", "abc", "coke sleek can", "fuzzy was a bear", "x" * 70, "q" * 5, "ab"]
    with Corpus.from_prepared(cand) as c:
        got = c.score_all(queries)
        exp = _oracle_scores(oracle, queries, cand)
        assert (got == exp).all(), (got.tolist(), exp.tolist())
        for ms in (0, 1, 50, 100, 101):
            _check_topk(oracle, c, queries, cand, ms, 5)
        idx, score, cnt = c.topk(queries, 0, 0)
        assert cnt.sum() == 0
    with Corpus.from_prepared([]) as c:
        idx, score, cnt = c.topk(["abc"], 0, 5)
        assert cnt.tolist() == [0]


def test_multiword_patterns(oracle):
    rng = np.random.default_rng(7)
    cps, offs = synth.make_corpus(700, max_chars=512, seed=5, min_tokens=8, max_tokens=60, vocab_size=300)
    cand = synth.to_strings(cps, offs)
    queries = [cand[int(i)] for i in rng.integers(0, len(cand), 6)]
    queries += synth.make_queries(cps, offs, 6, seed=6, max_chars=512)
    queries += [" ".join(cand[3].split()[:k]) for k in (5, 9, 17)]
    with Corpus(cps, offs) as c:
        got = c.score_all(queries)
        exp = _oracle_scores(oracle, queries, cand)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, [(queries[q][:40], cand[i][:40], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]
        _check_topk(oracle, c, queries[:4], cand, 35, 10)


def test_fuzzysearch_dropin(oracle):
    products = ["Coke Zero 330ml", "COKE 330ML 1x24 NP Sleek Can", "Sprite Zero", "Fanta Orange 1.5L",
                "Café Crème", "Coke-Zero (330ml)", "مياه معدنية 500ml", "coke_zero", "İstanbul Cola", ""]
    customers = ["FATHIMA GROCERY", "Al Madina Supermarket", "fathima grocery llc", "Carrefour"]
    fs = FuzzySearch(products, customers, threshold=0.35)
    for q in ("coke sleek can", "coke zero", "sprite", "مياه", "istanbul", "zzz", ""):
        got = fs.find_best_matches(q, "product")
        exp = oracle.c_find_best_matches(products, q, 0.35, 5)
        assert got == exp, (q, got, exp)
    got = fs.find_best_matches("fatima grocery", "anything-else-means-customers")
    assert got == oracle.c_find_best_matches(customers, "fatima grocery", 0.35, 5)
    many = fs.find_best_matches_many(["coke zero", "fanta"], "product")
    assert many[0] == fs.find_best_matches("coke zero", "product")
    fs.close()
