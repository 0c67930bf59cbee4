"""GPU parity of the scorer variants (BASELINE.json config 5) and of the raw distances:
ratio, token_sort_ratio, InDel distance, Myers Levenshtein -- bit-exact against the oracle."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tts_amd import Corpus, _lib, distance_batch, synth  # noqa: E402
from tts_amd.preprocess import prepare  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _unicode_corpus(n, seed):
    """mixed-case names with Latin-1 (dropped by full_process), Arabic tokens (kept), punctuation"""
    rnd = random.Random(seed)
    base = ["Coke", "Zero", "330ML", "Sleek", "Can", "Sprite", "Fanta", "Orange", "Light", "1.5L", "NP",
            "Café", "Crème", "Niño", "1×24", "مياه", "معدنية", "عصير", "برتقال", "٥٠٠", "Al-Madina", "(PET)",
            "Fathima", "Grocery", "&", "Sons", "L.L.C", "Super/Market", "x" * 9, "ab", "a"]
    out = []
    for _ in range(n):
        out.append(" ".join(rnd.choice(base) for _ in range(rnd.randint(1, 6))))
    return out


def test_ratio_and_token_sort_synthetic(oracle):
    cps, offs = synth.make_corpus(2500, max_chars=64, seed=31, vocab_size=200)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 24, seed=32) + ["", cand[3], " ".join(reversed(cand[4].split()))]
    buf, o = oracle.pack(cand)
    with Corpus(cps, offs) as c:
        for name, sc in (("ratio", _lib.RATIO), ("token_sort", _lib.TOKEN_SORT)):
            got = c.score_all(queries, sc)
            orc = oracle.RATIO if sc == _lib.RATIO else oracle.TOKEN_SORT
            exp = np.stack([oracle.score_all(orc, q, buf, o) for q in queries]).astype(np.uint8)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (name, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])
            idx, score, cnt = c.topk(queries[:6], 50, 7, sc)
            for qi in range(6):
                e = oracle.topk(orc, queries[qi], buf, o, 50, 7)
                assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == e


def test_config5_unicode_all_scorers(oracle):
    raw = _unicode_corpus(3000, 41) + ["", "   ", "İstanbul Cola", "coke_zero"]
    cand = [prepare(s) for s in raw]
    rawq = ["coke zero", "COKE-ZERO 330ml", "مياه معدنية", "Café Crème", "al madina super market", "fanta orange 1.5l",
            "fathima grocery & sons", "", "٥٠٠ عصير", "istanbul"]
    queries = [prepare(q) for q in rawq]
    buf, o = oracle.pack(cand)
    with Corpus.from_strings(raw) as c:
        for sc, orc in ((_lib.TOKEN_SET, oracle.TOKEN_SET), (_lib.TOKEN_SORT, oracle.TOKEN_SORT), (_lib.RATIO, oracle.RATIO)):
            got = c.score_all(queries, sc)
            exp = np.stack([oracle.score_all(orc, q, buf, o) for q in queries]).astype(np.uint8)
            if sc == _lib.RATIO:
                # the corpus was ingested through prepare_and_pack; ratio compares strings verbatim, and the
                # bulk path may leave different whitespace than prepare() -- compare on a prepared corpus
                continue
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (sc, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])
    with Corpus.from_prepared(cand) as c:
        for sc, orc in ((_lib.RATIO, oracle.RATIO), (_lib.PARTIAL, oracle.PARTIAL)):
            got = c.score_all(queries, sc)
            exp = np.stack([oracle.score_all(orc, q, buf, o) for q in queries]).astype(np.uint8)
            assert (got == exp).all(), sc


def test_distances_all_lengths(oracle):
    rnd = random.Random(51)
    alpha = "abcdefgh "
    cand = ["", "a", "ab", "ba"] + ["".join(rnd.choice(alpha) for _ in range(rnd.randint(0, 150))) for _ in range(600)]
    cand += ["".join(rnd.choice("ab") for _ in range(n)) for n in (31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 400)]
    queries = ["", "a", "abc"] + ["".join(rnd.choice(alpha) for _ in range(n)) for n in (5, 31, 32, 33, 63, 64, 65, 96, 97, 128, 129, 200, 257, 384, 500)]
    with Corpus.from_prepared(cand) as c:
        ind = c.distance_all(queries, 0)
        lev = c.distance_all(queries, 1)
    for qi, q in enumerate(queries):
        for i, s in enumerate(cand):
            assert int(ind[qi, i]) == oracle.indel(q, s), ("indel", q, s)
            assert int(lev[qi, i]) == oracle.levenshtein(q, s), ("lev", q, s)


def test_distance_batch_and_golden(oracle):
    with open(os.path.join(HERE, "golden", "pairs.json")) as f:
        recs = json.load(f)[:300]
    a = [r["q"] for r in recs]
    b = [r["c"] for r in recs]
    ind = distance_batch(a, b, 0)
    lev = distance_batch(a, b, 1)
    for i, r in enumerate(recs):
        assert int(ind[i]) == r["indel"] and int(lev[i]) == r["levenshtein"], r


def test_golden_ratio_token_sort():
    with open(os.path.join(HERE, "golden", "pairs.json")) as f:
        recs = json.load(f)
    cands = [r["c"] for r in recs]
    with Corpus.from_prepared(cands) as c:
        for lo in range(0, len(recs), 64):
            chunk = recs[lo:lo + 64]
            qs = [r["q"] for r in chunk]
            r1 = c.score_all(qs, _lib.RATIO)
            r2 = c.score_all(qs, _lib.TOKEN_SORT)
            r3 = c.score_all(qs, _lib.PARTIAL)
            for j, r in enumerate(chunk):
                assert int(r1[j, lo + j]) == r["ratio"], r
                assert int(r2[j, lo + j]) == r["token_sort"], r
                if r["partial"] is not None:
                    assert int(r3[j, lo + j]) == r["partial"], r


def test_partial_ratio(oracle):
    """fuzz.partial_ratio (reference app.py:1023): every window of the longer string, both directions
    at equal lengths; needles shorter and longer than the candidates, multi-word patterns, top-k."""
    rnd = random.Random(61)
    cps, offs = synth.make_corpus(1500, max_chars=48, seed=62, vocab_size=120)
    cand = synth.to_strings(cps, offs) + ["", "a", "ab", "x" * 70, "abc abc"]
    cand += ["".join(rnd.choice("abc ") for _ in range(rnd.randint(1, 14))) for _ in range(300)]
    queries = ["", "a", "abc", "b a c", cand[3], cand[3][:9], "abcabcabca", "x" * 33, "xy" * 40]
    queries += synth.make_queries(cps, offs, 10, seed=63, max_chars=48)
    buf, o = oracle.pack(cand)
    with Corpus.from_prepared(cand) as c:
        got = c.score_all(queries, _lib.PARTIAL)
        exp = np.stack([oracle.score_all(oracle.PARTIAL, q, buf, o) for q in queries]).astype(np.uint8)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]
        idx, score, cnt = c.topk(queries[:8], 60, 9, _lib.PARTIAL)
        for qi in range(8):
            e = oracle.topk(oracle.PARTIAL, queries[qi], buf, o, 60, 9)
            assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == e


def _wide_corpus(n, seed, max_tok=5):
    """names over > 254 distinct code points (CJK, Arabic, Cyrillic, ASCII): 16-bit symbols"""
    rnd = random.Random(seed)
    alpha = [chr(c) for c in range(0x4E00, 0x4E00 + 900)] + [chr(c) for c in range(0x0621, 0x064A)] + \
            [chr(c) for c in range(0x0430, 0x0450)] + list("abcdefghij0123456789")
    vocab = ["".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 6))) for _ in range(700)]
    return [" ".join(rnd.choice(vocab) for _ in range(rnd.randint(1, max_tok))) for _ in range(n)], vocab


def test_wide_alphabet_all_scorers(oracle):
    """A corpus with more than 254 distinct code points (16-bit symbols, indirect Peq rows)."""
    cand, vocab = _wide_corpus(2500, 71)
    cand += ["", vocab[0], "x" * 70, " ".join(vocab[:40])]
    assert len(set("".join(cand))) > 600
    queries = [cand[3], cand[10] + " " + vocab[5], vocab[7], "", "zz", "丁丂 abc", " ".join(vocab[10:32]), cand[77][1:]]
    buf, o = oracle.pack(cand)
    with Corpus.from_prepared(cand) as c:
        assert c.info()["alphabet_size"] > 256
        for sc, orc in ((_lib.TOKEN_SET, oracle.TOKEN_SET), (_lib.RATIO, oracle.RATIO),
                        (_lib.TOKEN_SORT, oracle.TOKEN_SORT), (_lib.PARTIAL, oracle.PARTIAL)):
            got = c.score_all(queries, sc)
            exp = np.stack([oracle.score_all(orc, q, buf, o) for q in queries]).astype(np.uint8)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (sc, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])
        idx, score, cnt = c.topk(queries, 35, 10)
        for qi, q in enumerate(queries):
            e = oracle.topk(oracle.TOKEN_SET, q, buf, o, 35, 10)
            assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == e
        lev = c.distance_all(queries[:3], 1)
        for qi in range(3):
            for i in range(0, len(cand), 37):
                assert int(lev[qi, i]) == oracle.levenshtein(queries[qi], cand[i])


def test_weighted_fuzzysearch_dropin(oracle):
    """The reference's weighted FuzzySearch (app.py:949-1032) end to end: dict de-duplication,
    exact-key and ratio>98 bypasses, 0.5*partial + 0.5*token_set capped at 0.99, top_n."""
    from tts_amd.weighted_search import WeightedFuzzySearch
    cps, offs = synth.make_corpus(3000, max_chars=40, seed=81, vocab_size=60)
    base = synth.to_strings(cps, offs)
    products = [s.upper() if i % 4 == 0 else s for i, s in enumerate(base)]
    products += ["Coke 1×24 NP Sleek Can", "coke 1x24 np sleek can", "Sprite 330ml 1x24", "Sprite Zero", "", "  ",
                 "Fanta Orange 1.5L (PET)", "fanta-orange 1.5l", "abcdefgh" * 13 + "x", "abcdefgh" * 13 + "y"]
    customers = ["FATHIMA GROCERY", "Al Madina Supermarket", "fathima grocery llc", "Carrefour", "fathima grocery"]
    fs = WeightedFuzzySearch(products, customers)
    assert fs.products_normalized == {oracle.weighted_normalize_string(p): p for p in products}
    queries = ["sprite 330ml", "coke 1×24 np sleek can", "COKE 1X24 NP SLEEK CAN", "fanta orange", "sprit zero", "",
               "abcdefgh" * 13 + "z", "zzzz", base[5], base[9][:6], base[11] + " xx"] + synth.make_queries(cps, offs, 6, seed=82, max_chars=40)
    for q in queries:
        got = fs.find_best_matches(q, "product")
        exp = oracle.weighted_find_best_matches(products, q)
        assert got == exp, (q, got[:3], exp[:3])
    for q in ("fatima grocery", "fathima grocery", "madina", "carrefour express"):
        assert fs.find_best_matches(q, "customer") == oracle.weighted_find_best_matches(customers, q), q
    fs.close()
    lo = WeightedFuzzySearch(products, customers, threshold=0.30, top_n=25)
    for q in queries[:6]:
        assert lo.find_best_matches(q, "product") == oracle.weighted_find_best_matches(products, q, 0.30, 0.98, 25), q
    lo.close()


def test_advanced_fuzzy_matcher_dropin(oracle):
    """AdvancedFuzzyMatcher.fuzzy_search (reference app.py:378-460): exact, unit conversion, prefix,
    substring, SequenceMatcher, bigram/trigram Jaccard and Levenshtein strategies; same dicts out."""
    from tts_amd.advanced_matcher import AdvancedFuzzyMatcher
    rnd = random.Random(101)
    brands = ["Coke", "Coke Zero", "Sprite", "Fanta Orange", "Mirinda", "Aquafina", "Mountain Dew", "7UP", "Pepsi Max"]
    sizes = ["330ml", "1.5L", "1500ml", "2 kg", "2000g", "500 ML", "1L", "1 ltr", "24 pcs", "24 pieces", "0.33l", ""]
    packs = ["1x24", "NP", "Sleek Can", "PET", "(Glass)", "Promo-Pack", ""]
    candidates = []
    for _ in range(2500):
        parts = [rnd.choice(brands), rnd.choice(sizes), rnd.choice(packs)]
        rnd.shuffle(parts)
        candidates.append(" ".join(p for p in parts if p))
    candidates += ["", "&&", "coke", "COKE", "Sprite Zero Sugar 330ml", "x" * 250 + " long tail name " + "y" * 30]
    queries = ["coke", "Coke 1.5L", "coke 1500ml", "sprit", "fanta orng 330", "mountan dew 2kg", "7up 24 pcs", "aquafina 0.5l",
               "pepsi", "zero", "Sprite Zero Sugar 330 ml", "mirnda pet", "1x24", "qqqq", "coke zero 330ml sleek can"]
    for q in queries:
        got = AdvancedFuzzyMatcher.fuzzy_search(q, candidates)
        exp = oracle.advanced_fuzzy_search(q, candidates)
        assert got == exp, (q, got[:3], exp[:3])
    for thr, top in ((0.3, 40), (0.9, 3), (0.55, 12)):
        for q in queries[:8]:
            got = AdvancedFuzzyMatcher.fuzzy_search(q, candidates, threshold=thr, top_n=top)
            assert got == oracle.advanced_fuzzy_search(q, candidates, thr, top), (thr, top, q)
    assert AdvancedFuzzyMatcher.fuzzy_search("", candidates) == [] and AdvancedFuzzyMatcher.fuzzy_search("coke", []) == []
    # SequenceMatcher's matches from the GPU for every survivor (fz_distance_all kind 4) and from difflib for every
    # survivor: the same dicts either way
    for force in (0, 10 ** 9):
        AdvancedFuzzyMatcher.SEQUENCE_ON_GPU_FROM = force
        for q in queries:
            assert AdvancedFuzzyMatcher.fuzzy_search(q, candidates, threshold=0.3, top_n=40) == oracle.advanced_fuzzy_search(q, candidates, 0.3, 40), (force, q)
    AdvancedFuzzyMatcher.SEQUENCE_ON_GPU_FROM = 48
    AdvancedFuzzyMatcher.clear_cache()


def test_sequence_matcher_matches_kind4():
    """fz_distance_all kind 4 against difflib.SequenceMatcher (stdlib) on the GPU: 3 000 candidates, small alphabets"""
    from difflib import SequenceMatcher
    rnd = random.Random(141)
    def word(alpha, lo, hi):
        return "".join(rnd.choice(alpha) for _ in range(rnd.randint(lo, hi)))
    cand = [word("ab", 0, 64) for _ in range(1000)] + [word("abcd", 0, 50) for _ in range(1000)] + [word("abcdefghijklmnop0123", 1, 64) for _ in range(1000)]
    cand += ["", "a" * 64, "x" * 65, "ab" * 50]
    queries = ["", "a", "ab" * 16, "abcabcabc", word("ab", 33, 64), word("abcd", 20, 32), word("abcdefghijklmnop0123", 40, 64), "a" * 64]
    with Corpus.from_prepared(cand) as c:
        got = c.distance_all(queries, 4)
        with pytest.raises(Exception):
            c.distance_all(["q" * 65], 4)
    for qi, q in enumerate(queries):
        for i, s in enumerate(cand):
            exp = 0xFFFF if len(s) > 64 else sum(b.size for b in SequenceMatcher(None, q, s, autojunk=False).get_matching_blocks())
            assert int(got[qi, i]) == exp, (q, s, int(got[qi, i]), exp)


def test_common_tokens_and_substring_kinds():
    """fz_distance_all kind 2 (|A & B| over distinct tokens) and kind 3 (substring search), definitional checks"""
    rnd = random.Random(111)
    toks = ["ab", "bc", "cd", "de", "ef", "zz", "q1", "w2", "abc", "bcd"] + ["t%d" % i for i in range(40)]
    cand = [" ".join(rnd.choice(toks) for _ in range(rnd.randint(0, 30))) for _ in range(1500)] + ["", "ab ab ab"]
    queries = ["ab bc cd", "", "zz", "q1 q1 w2 nope", " ".join(toks)]
    with Corpus.from_prepared(cand) as c:
        got = c.distance_all(queries, 2)
    for qi, q in enumerate(queries):
        for i, s in enumerate(cand):
            assert int(got[qi, i]) == len(set(q.split()) & set(s.split())), (q, s)
    text = ["".join(rnd.choice("abc") for _ in range(rnd.randint(0, 200))) for _ in range(1500)] + ["", "abc", "\x01abc"]
    pats = ["", "a", "abc", "cab", "abca" * 9, "\x01ab", "c" * 33, "ab" * 40, "x", "abcabc" * 30]
    with Corpus.from_prepared(text) as c:
        got = c.distance_all(pats, 3)
    for qi, q in enumerate(pats):
        for i, s in enumerate(text):
            assert int(got[qi, i]) == (1 if q in s else 0), (q, s)
