"""CPU oracle for the candidate-scoring loop -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package (``tts_amd``) never does, and fails loudly without its HIP library.

PARITY UNPINNED.  The reference ships no golden vectors for this path (its ``tests/tests.py`` is
empty) and its scorer lives in third-party packages absent from ``/root/reference`` and from this
image: ``thefuzz 0.22.1`` (reference ``uv.lock:4634-4635``) and ``rapidfuzz 3.14.1``
(``uv.lock:3842-3843``).  What follows restates their published algorithm, anchored on the
reference's call sites (``src/fuzzy_search.py:16-38``; ``app.py:1010,1023,1026``).

Two layers:
  * ``fuzz_oracle.c`` (built to ``oracle/_build/liboracle.so``): distances and scorers on UTF-32.
  * this file: the Python-level steps that depend on Python's own Unicode tables
    (``str.lower``, ``str.isalnum``) and a pure-Python ``token_set_ratio`` used only on tiny cases
    to cross-check the C restatement.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

TOKEN_SET, RATIO, TOKEN_SORT, PARTIAL = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """gcc-compile fuzz_oracle.c (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "fuzz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        u32p, u64p, i32p, i64p = (ctypes.POINTER(t) for t in
                                  (ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int64))
        pair = [u32p, ctypes.c_int64, u32p, ctypes.c_int64]
        for name in ("orc_lcs_dp", "orc_lcs_bitpar", "orc_indel_distance", "orc_levenshtein"):
            getattr(L, name).argtypes = pair
            getattr(L, name).restype = ctypes.c_int64
        for name in ("orc_ratio", "orc_token_sort_ratio", "orc_token_set_ratio", "orc_partial_ratio"):
            getattr(L, name).argtypes = pair
            getattr(L, name).restype = ctypes.c_double
        L.orc_token_set_ratio_detail.argtypes = pair + [i64p]
        L.orc_token_set_ratio_detail.restype = ctypes.c_double
        L.orc_norm_distance_100.argtypes = [ctypes.c_int64, ctypes.c_int64]
        L.orc_norm_distance_100.restype = ctypes.c_double
        L.orc_norm_sim_100.argtypes = [ctypes.c_int64, ctypes.c_int64]
        L.orc_norm_sim_100.restype = ctypes.c_double
        L.orc_py_round.argtypes = [ctypes.c_double]
        L.orc_py_round.restype = ctypes.c_int
        L.orc_set_bitpar.argtypes = [ctypes.c_int]
        L.orc_canonical.argtypes = [u32p, ctypes.c_int64, ctypes.c_int, u32p]
        L.orc_canonical.restype = ctypes.c_int64
        L.orc_score_int.argtypes = [ctypes.c_int] + pair
        L.orc_score_int.restype = ctypes.c_int
        L.orc_score_all.argtypes = [ctypes.c_int, u32p, ctypes.c_int64, u32p, u64p, ctypes.c_int64, i32p]
        L.orc_topk.argtypes = [ctypes.c_int, u32p, ctypes.c_int64, u32p, u64p, ctypes.c_int64,
                               ctypes.c_int32, ctypes.c_int64, i64p, i32p]
        L.orc_topk.restype = ctypes.c_int64
        _lib = L
    return _lib


# ----------------------------------------------------------------------------------------------
# Python-level preprocessing
# ----------------------------------------------------------------------------------------------
_DROP_LATIN1 = {i: None for i in range(128, 256)}   # thefuzz utils: translation_table, force_ascii


def full_process(s: str) -> str:
    """thefuzz ``utils.full_process(s, force_ascii=True)``: drop U+0080..U+00FF, then rapidfuzz
    ``default_process``: every non-alphanumeric character -> ' ', lowercase, strip.

    ASSUMPTION (SURVEY.md 8c, low confidence, isolated here): the character class is
    ``str.isalnum()`` -- so ``'_'`` becomes a space -- which is how the compiled rapidfuzz wheel
    generates its tables; the pure-Python fallback of rapidfuzz uses the regex ``\\W`` and would keep
    ``'_'``."""
    s = s.translate(_DROP_LATIN1)
    out = []
    for ch in s:
        out.append(ch if ch.isalnum() else " ")
    return "".join(out).lower().strip()


def reference_inputs(query: str, item: str) -> Tuple[str, str]:
    """What reaches the scorer at reference src/fuzzy_search.py:19,26: both sides lowercased by the
    reference itself, then ``full_process``-ed by the thefuzz wrapper."""
    return full_process(query.lower()), full_process(item.lower())


def min_score_for_threshold(threshold: float) -> int:
    """Smallest integer score s in 0..100 with ``s / 100.0 >= threshold`` in Python floats
    (src/fuzzy_search.py:29-31); 101 when none qualifies."""
    for s in range(0, 101):
        if s / 100.0 >= threshold:
            return s
    return 101


# ----------------------------------------------------------------------------------------------
# packing helpers
# ----------------------------------------------------------------------------------------------
def cps(s: str) -> np.ndarray:
    return np.frombuffer(s.encode("utf-32-le"), dtype=np.uint32).copy()


def _p32(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def pack(strings: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """-> (code points uint32, offsets uint64[n+1])"""
    lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=len(strings))
    offs = np.zeros(len(strings) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    buf = np.frombuffer("".join(strings).encode("utf-32-le"), dtype=np.uint32).copy()
    if buf.size == 0:
        buf = np.zeros(1, dtype=np.uint32)
    return buf, offs


def _call2(fn, a: str, b: str):
    A, B = cps(a), cps(b)
    if A.size == 0:
        A = np.zeros(1, np.uint32)
    if B.size == 0:
        B = np.zeros(1, np.uint32)
    return fn(_p32(A), len(a), _p32(B), len(b))


def lcs_dp(a: str, b: str) -> int:
    return _call2(lib().orc_lcs_dp, a, b)


def lcs_bitpar(a: str, b: str) -> int:
    return _call2(lib().orc_lcs_bitpar, a, b)


def indel(a: str, b: str) -> int:
    return _call2(lib().orc_indel_distance, a, b)


def levenshtein(a: str, b: str) -> int:
    return _call2(lib().orc_levenshtein, a, b)


def score(scorer: int, a: str, b: str) -> int:
    """thefuzz-level integer score of already-processed strings: int(round(rapidfuzz scorer))."""
    A, B = cps(a), cps(b)
    if A.size == 0:
        A = np.zeros(1, np.uint32)
    if B.size == 0:
        B = np.zeros(1, np.uint32)
    return lib().orc_score_int(scorer, _p32(A), len(a), _p32(B), len(b))


def token_set_detail(a: str, b: str):
    A, B = cps(a), cps(b)
    if A.size == 0:
        A = np.zeros(1, np.uint32)
    if B.size == 0:
        B = np.zeros(1, np.uint32)
    det = (ctypes.c_int64 * 5)()
    x = lib().orc_token_set_ratio_detail(_p32(A), len(a), _p32(B), len(b), det)
    return x, list(det)


def score_all(scorer: int, query: str, buf: np.ndarray, offs: np.ndarray) -> np.ndarray:
    n = len(offs) - 1
    out = np.zeros(max(n, 1), dtype=np.int32)
    Q = cps(query)
    if Q.size == 0:
        Q = np.zeros(1, np.uint32)
    lib().orc_score_all(scorer, _p32(Q), len(query), _p32(buf),
                        offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n,
                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out[:n]


def topk(scorer: int, query: str, buf: np.ndarray, offs: np.ndarray, min_score: int, k: int):
    """-> list of (index, int score): the C restatement of the loop + stable sort + slice."""
    n = len(offs) - 1
    oi = np.zeros(max(k, 1), dtype=np.int64)
    os_ = np.zeros(max(k, 1), dtype=np.int32)
    Q = cps(query)
    if Q.size == 0:
        Q = np.zeros(1, np.uint32)
    c = lib().orc_topk(scorer, _p32(Q), len(query), _p32(buf),
                       offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n, min_score, k,
                       oi.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                       os_.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return [(int(oi[i]), int(os_[i])) for i in range(c)]


# ----------------------------------------------------------------------------------------------
# pure-Python restatement (tiny cases only) -- independent of the C file
# ----------------------------------------------------------------------------------------------
def _py_lcs(a: str, b: str) -> int:
    prev = [0] * (len(b) + 1)
    for ch in a:
        cur = [0]
        for j, dh in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if ch == dh else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[len(b)]


def _py_norm_distance(dist: int, lensum: int) -> float:
    return 100.0 - 100.0 * dist / lensum if lensum > 0 else 100.0


def py_token_set_ratio(s1: str, s2: str) -> float:
    """rapidfuzz fuzz_py.token_set_ratio, restated."""
    ta, tb = set(s1.split()), set(s2.split())
    if not ta or not tb:
        return 0
    inter, dab, dba = ta & tb, ta - tb, tb - ta
    if inter and (not dab or not dba):
        return 100
    ja, jb = " ".join(sorted(dab)), " ".join(sorted(dba))
    sect_len = len(" ".join(inter))
    e = 1 if sect_len else 0
    la, lb = sect_len + e + len(ja), sect_len + e + len(jb)
    dist = len(ja) + len(jb) - 2 * _py_lcs(ja, jb)
    result = _py_norm_distance(dist, la + lb)
    if not sect_len:
        return result
    r1 = _py_norm_distance(e + len(ja), sect_len + la)
    r2 = _py_norm_distance(e + len(jb), sect_len + lb)
    return max(result, r1, r2)


def py_thefuzz_token_set_ratio(s1: str, s2: str) -> int:
    """thefuzz.fuzz.token_set_ratio(s1, s2): full_process both, score, int(round())  (row a4)."""
    p1, p2 = full_process(s1), full_process(s2)
    return int(round(py_token_set_ratio(p1, p2)))


def py_find_best_matches(items: List[str], query: str, threshold: float, top_n: int):
    """The loop of reference src/fuzzy_search.py:16-38, statement for statement, over the
    pure-Python scorer above (baseline B3 of BASELINE.md; tiny inputs only)."""
    query_lower = query.lower()
    matches = []
    for item in items:
        similarity_score = py_thefuzz_token_set_ratio(query_lower, item.lower())
        similarity_float = similarity_score / 100.0
        if similarity_float >= threshold:
            matches.append((item, similarity_float))
    matches.sort(key=lambda x: x[1], reverse=True)
    return matches[:top_n]


def c_find_best_matches(items: List[str], query: str, threshold: float, top_n: int,
                        scorer: int = TOKEN_SET):
    """Same loop with the C scorer: preprocess in Python, score/sort/slice in C."""
    proc = [full_process(it.lower()) for it in items]
    buf, offs = pack(proc)
    q = full_process(query.lower())
    res = topk(scorer, q, buf, offs, min_score_for_threshold(threshold), top_n)
    return [(items[i], s / 100.0) for i, s in res]


# ----------------------------------------------------------------------------------------------
# the reference's WEIGHTED FuzzySearch (app.py:949-1032), statement for statement over the C scorers
# ----------------------------------------------------------------------------------------------
def weighted_normalize_string(s) -> str:
    """app.py:974-979"""
    if not isinstance(s, str):
        return ""
    return s.lower().replace('\u00d7', 'x').replace('\u00d7', 'x').strip()


def weighted_score(query_norm: str, item_norm: str) -> float:
    """app.py:1014-1032.  thefuzz: partial_ratio does not preprocess, token_set_ratio does."""
    if query_norm == item_norm:
        return 1.0
    partial_score = score(PARTIAL, query_norm, item_norm) / 100.0
    token_set_score = score(TOKEN_SET, full_process(query_norm), full_process(item_norm)) / 100.0
    sc = (partial_score * 0.5) + (token_set_score * 0.5)
    return min(sc, 0.99)


def weighted_find_best_matches(items: List[str], query: str, threshold: float = 0.70,
                               exact_match_threshold: float = 0.98, top_n: int = 10):
    """app.py:981-1012 for one entity list."""
    items_map = {weighted_normalize_string(p): p for p in items}
    query_normalized = weighted_normalize_string(query)
    exact_match = None
    if query_normalized in items_map:
        exact_match = items_map[query_normalized]
    else:
        for norm_item, original_item in items_map.items():
            if score(RATIO, query_normalized, norm_item) > (exact_match_threshold * 100):
                exact_match = original_item
                break
    if exact_match:
        return [(exact_match, 1.0)]
    matches = []
    for normalized_item, original_item in items_map.items():
        final_score = weighted_score(query_normalized, normalized_item)
        if final_score >= threshold:
            matches.append((original_item, final_score))
    matches.sort(key=lambda x: x[1], reverse=True)
    return matches[:top_n]


# ----------------------------------------------------------------------------------------------
# the reference's AdvancedFuzzyMatcher (app.py:130-460), per candidate, statement for statement.
# Pure Python + difflib (stdlib) + the C Levenshtein DP above; tiny inputs only.
# ----------------------------------------------------------------------------------------------
import re as _re
from difflib import SequenceMatcher as _SM

ADV_CONFIG = {'threshold': 0.7, 'top_n': 5, 'prefix_weight': 1.0, 'substring_weight': 0.9,
              'sequence_weight': 0.85, 'jaccard_weight': 0.8, 'levenshtein_weight': 0.8,
              'unit_conversion_weight': 0.95}                                   # app.py:253-262
ADV_UNITS = {                                                                   # app.py:116-128
    'ml': {'ml': 1, 'l': 0.001, 'lt': 0.001, 'ltr': 0.001},
    'l': {'ml': 1000, 'l': 1, 'lt': 1, 'ltr': 1},
    'lt': {'ml': 1000, 'l': 1, 'lt': 1, 'ltr': 1},
    'ltr': {'ml': 1000, 'l': 1, 'lt': 1, 'ltr': 1},
    'g': {'g': 1, 'kg': 0.001},
    'kg': {'g': 1000, 'kg': 1},
    'pcs': {'pcs': 1, 'pc': 1, 'piece': 1, 'pieces': 1},
    'pc': {'pcs': 1, 'pc': 1, 'piece': 1, 'pieces': 1},
}


def adv_normalize_text(text):                        # app.py:271-273
    return _re.sub(r'[^a-z0-9]', '', text.lower())


def _adv_extract(text):                              # app.py:137-151
    m = _re.search(r'(\d+\.?\d*)\s*([a-zA-Z]+)', text.lower())
    if m:
        try:
            return float(m.group(1)), m.group(2).lower()
        except (ValueError, AttributeError):
            return None
    return None


def _adv_norm_unit(unit):                            # app.py:154-167
    u = unit.lower()
    if u in ADV_UNITS:
        return u
    for std, conv in ADV_UNITS.items():
        if u in conv:
            return std
    return None


def _adv_equivalent(t1, t2, tol=0.01):               # app.py:170-220
    a, b = _adv_extract(t1), _adv_extract(t2)
    if not a or not b:
        return False
    (q1, u1), (q2, u2) = a, b
    n1, n2 = _adv_norm_unit(u1), _adv_norm_unit(u2)
    if not n1 or not n2:
        return False
    if n1 not in ADV_UNITS or n2 not in ADV_UNITS[n1]:
        return False
    converted = q1 * ADV_UNITS[n1][n2]
    return abs(converted - q2) <= (q2 * tol)


def _adv_unit_score(query, target):                  # app.py:348-376
    if not _adv_extract(query) or not _adv_extract(target):
        return 0.0
    if _adv_equivalent(query, target):
        a = _re.sub(r'\d+\.?\d*\s*[a-zA-Z]+', '', query).strip()
        b = _re.sub(r'\d+\.?\d*\s*[a-zA-Z]+', '', target).strip()
        if a.lower() == b.lower():
            return 1.0
        sim = _SM(None, adv_normalize_text(a), adv_normalize_text(b)).ratio()
        return sim if sim > 0.6 else 0.0
    return 0.0


def _adv_jaccard(s1, s2, n):                         # app.py:276-293
    a, b = adv_normalize_text(s1), adv_normalize_text(s2)
    g1 = set(a[i:i + n] for i in range(len(a) - n + 1))
    g2 = set(b[i:i + n] for i in range(len(b) - n + 1))
    if not g1 or not g2:
        return 0.0
    inter, union = len(g1 & g2), len(g1 | g2)
    return inter / union if union > 0 else 0.0


def _adv_prefix(query, target):                      # app.py:320-334
    qn, tn = adv_normalize_text(query), adv_normalize_text(target)
    if tn.startswith(qn):
        return 1.0
    for word in target.lower().split():
        if adv_normalize_text(word).startswith(qn):
            return 0.9
    return 0.0


def _adv_substring(query, target):                   # app.py:337-345
    qn, tn = adv_normalize_text(query), adv_normalize_text(target)
    if qn in tn:
        return len(qn) / len(tn)
    return 0.0


def advanced_fuzzy_search(query, candidates, threshold=None, top_n=None):
    """app.py:378-460"""
    if threshold is None:
        threshold = ADV_CONFIG['threshold']
    if top_n is None:
        top_n = ADV_CONFIG['top_n']
    if not query or not candidates:
        return []
    results = []
    for candidate in candidates:
        if query.lower() == candidate.lower():
            results.append({'value': candidate, 'score': 1.0, 'method': 'exact'})
            continue
        unit_score = _adv_unit_score(query, candidate)
        if unit_score > 0:
            results.append({'value': candidate, 'score': unit_score * ADV_CONFIG['unit_conversion_weight'],
                            'method': 'unit_conversion'})
            continue
        prefix_score = _adv_prefix(query, candidate)
        substring_score = _adv_substring(query, candidate)
        qn, cn = adv_normalize_text(query), adv_normalize_text(candidate)
        sequence_score = _SM(None, qn, cn).ratio()
        j2 = _adv_jaccard(query, candidate, 2)
        j3 = _adv_jaccard(query, candidate, 3)
        lev_distance = levenshtein(qn, cn)           # app.py:296-317: Wagner-Fischer on the normalised strings
        max_len = max(len(qn), len(cn))
        lev_score = 1 - (lev_distance / max_len) if max_len > 0 else 0
        final_score = max(prefix_score * ADV_CONFIG['prefix_weight'], substring_score * ADV_CONFIG['substring_weight'],
                          sequence_score * ADV_CONFIG['sequence_weight'], j2 * ADV_CONFIG['jaccard_weight'],
                          j3 * 0.75, lev_score * ADV_CONFIG['levenshtein_weight'])
        method_scores = {'prefix': prefix_score, 'substring': substring_score, 'sequence': sequence_score,
                         'jaccard': max(j2, j3), 'levenshtein': lev_score}
        best_method = max(method_scores, key=method_scores.get)
        if final_score >= threshold:
            results.append({'value': candidate, 'score': final_score, 'method': best_method})
    results.sort(key=lambda x: x['score'], reverse=True)
    return results[:top_n]
