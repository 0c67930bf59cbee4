#!/usr/bin/env python3
"""Generates tests/golden/*.json with the repo's own oracle (oracle/fuzz_oracle.c, plain DP).

PARITY UNPINNED: thefuzz/rapidfuzz are not installed here and the reference holds no vectors, so these
fixtures pin the GPU path and later refactors of the oracle to the oracle as first written -- plus the
seven hand-derived known answers of SURVEY.md section 4, which are stored with their derivation.
Run from the repo root:  python tests/golden/make_golden.py"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fuzz_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def record(q, c, tag):
    x, det = O.token_set_detail(q, c)
    return {"tag": tag, "q": q, "c": c,
            "sect_len": det[0], "ab_len": det[1], "ba_len": det[2], "indel_diff": det[3], "lensum": det[4],
            "token_set": O.score(O.TOKEN_SET, q, c), "ratio": O.score(O.RATIO, q, c),
            "token_sort": O.score(O.TOKEN_SORT, q, c), "partial": O.score(O.PARTIAL, q, c) if len(q) * len(c) < 40000 else None,
            "indel": O.indel(q, c), "levenshtein": O.levenshtein(q, c)}


def main():
    rnd = random.Random(20251017)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    arabic = "ابتثجحخدذرزسشصضطظعغفقكلمنهوي٠١٢٣"
    words = ["coke", "zero", "330ml", "1x24", "np", "sleek", "can", "sprite", "fanta", "orange", "light",
             "grocery", "fathima", "al", "madina", "supermarket", "a", "b", "ab", "ba"] + \
            ["".join(rnd.choice(alpha) for _ in range(rnd.randint(2, 9))) for _ in range(60)] + \
            ["".join(rnd.choice(arabic) for _ in range(rnd.randint(2, 7))) for _ in range(15)]

    def name(lo=1, hi=5):
        return " ".join(rnd.choice(words) for _ in range(rnd.randint(lo, hi)))

    def mutate(s):
        s = list(s)
        for _ in range(rnd.randint(1, 3)):
            if not s:
                break
            i = rnd.randrange(len(s)); op = rnd.randrange(3)
            if op == 0: s[i] = rnd.choice(alpha)
            elif op == 1: s.insert(i, rnd.choice(alpha))
            else: del s[i]
        return " ".join("".join(s).split())

    recs = []
    # fixed edge cases
    for q, c, tag in [("", "", "both-empty"), ("", "abc", "q-empty"), ("abc", "", "c-empty"), ("abc", "abc", "identical"),
                      ("abc def", "def abc", "permuted"), ("abc", "xyz", "disjoint"), ("a", "a a a", "dup-tokens"),
                      ("coke sleek can", "coke 330ml 1x24 np sleek can", "subset-100"),
                      ("coke zero", "coke light", "one-common"), ("a b", "b c", "one-common-short"),
                      ("fuzzy was a bear", "fuzzy fuzzy was a bear", "kat"), ("aaaa", "aa", "single-token"),
                      ("ab", "ba", "single-token"), ("abab abab", "abab", "dup-subset")]:
        recs.append(record(q, c, tag))
    # rounding ties: exact x.5 scores for token_set r0 (no common token): lensum L, dist d with 200*d/L odd
    for L, d in [(16, 2), (16, 6), (16, 10), (48, 6), (80, 2), (80, 6), (80, 14), (400, 2), (400, 6), (400, 10)]:
        # q = 'a'*la, c = 'a'*k + 'b'*... build strings with given indel: q = x^la, c = x^lcs y^(lb-lcs)
        # (a tie needs 200*d/L odd; indel parity forces d == L mod 2, so L and d are both even)
        la = L // 2; lb = L - la; lcs = (L - d) // 2
        q = "x" * la; c = "x" * lcs + "y" * (lb - lcs)
        recs.append(record(q, c, f"tie-L{L}-d{d}"))
    # boundary lengths for the word-count classes
    for n in (31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 512):
        toks = []
        while len(" ".join(toks)) < n:
            toks.append("".join(rnd.choice(alpha) for _ in range(rnd.randint(2, 9))))
        s = " ".join(sorted(set(toks)))[:n].rstrip()
        recs.append(record(s, mutate(s), f"len{n}-mut"))
        recs.append(record(s, name(2, 6), f"len{n}-vs-short"))
        recs.append(record(name(1, 3), s, f"short-vs-len{n}"))
    # random name pairs, related and unrelated
    for i in range(900):
        a = name()
        b = mutate(a) if i % 3 == 0 else (" ".join(rnd.sample(a.split() + name().split(), k=min(4, len(a.split()) + 1))) if i % 3 == 1 else name())
        recs.append(record(a, b, "random"))
    with open(os.path.join(HERE, "pairs.json"), "w") as f:
        json.dump(recs, f, ensure_ascii=False, indent=0)

    # raw strings through the reference's lower() + thefuzz full_process (assumptions flagged)
    raw = ["Coke Zero 330ml", "COKE-ZERO (330ML)", "Café Crème", "1×24", "coke_zero", "  padded  ", "İstanbul",
           "مياه معدنية 500ml", "Straße 7", "ÀÉÎ", "tab\tsep\nnl", "a.b/c&d-e(f)", "ǅ mixed", "K kelvin", "ß", "٣ عبوات",
           "é combining", "Ÿ", ""]
    pre = [{"raw": r, "prepared": O.full_process(r.lower()),
            "assumption": "isalnum-class" if ("_" in r or any(ord(ch) > 255 for ch in r)) else "ascii/latin1"} for r in raw]
    with open(os.path.join(HERE, "preprocess.json"), "w") as f:
        json.dump(pre, f, ensure_ascii=False, indent=0)

    # hand-derived known answers (and the upstream README's examples)
    kats = [
        {"scorer": "ratio", "a": "this is a test", "b": "this is a test!", "score": 97,
         "why": "lens 14+15=29, LCS 14, indel 1, 100*(1-1/29)=96.55"},
        {"scorer": "partial", "a": "this is a test", "b": "this is a test!", "score": 100, "why": "window == needle"},
        {"scorer": "ratio", "a": "fuzzy wuzzy was a bear", "b": "wuzzy fuzzy was a bear", "score": 91,
         "why": "22+22=44, indel 4, 90.9"},
        {"scorer": "token_sort", "a": "fuzzy wuzzy was a bear", "b": "wuzzy fuzzy was a bear", "score": 100, "why": "same sorted tokens"},
        {"scorer": "token_sort", "a": "fuzzy was a bear", "b": "fuzzy fuzzy was a bear", "score": 84,
         "why": "'a bear fuzzy was'(16) vs 'a bear fuzzy fuzzy was'(22), indel 6, 100*32/38=84.2"},
        {"scorer": "token_set", "a": "fuzzy was a bear", "b": "fuzzy fuzzy was a bear", "score": 100, "why": "token sets equal"},
        {"scorer": "token_set", "a": "coke sleek can", "b": "coke 330ml 1x24 np sleek can", "score": 100,
         "why": "reference src/fuzzy_search.py:23-25: query tokens are a subset"},
    ]
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(kats, f, indent=0)
    print(len(recs), "pairs,", len(pre), "preprocess cases,", len(kats), "kats")


if __name__ == "__main__":
    main()
