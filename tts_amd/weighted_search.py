"""Drop-in for the reference's WEIGHTED ``FuzzySearch`` (reference ``app.py:949-1032``): dict
de-duplication by normalised key, exact-key bypass, ``fuzz.ratio > 98`` bypass, then
``0.5 * partial_ratio/100 + 0.5 * token_set_ratio/100`` capped at 0.99 -- with every scorer call of
the three loops running on the MI355X through ``libfuzzyhip.so``.

    reference (app.py)                                    here
    __init__ dict comprehensions          :969-970        same dicts (first-insertion order, last value)
    normalize_string                      :974-979        same statement
    _check_exact_match key lookup         :1006-1007      host dict lookup
    _check_exact_match ratio scan         :1009-1011      fz_query_topk(RATIO | ORDER_BY_INDEX, k=1)
    _calculate_weighted_score loop        :995-999        fz_query_topk_pair(PARTIAL, TOKEN_SET)
    fp64 weighting, cap, threshold        :1029-1032,:998 host evaluates the expression for all 101x101
                                                          integer score pairs; the device orders by rank
    stable sort desc + [:top_n]           :1001-1002      fz_select16_topk: (rank desc, dict order asc)

There is no CPU path: construction raises if the library or a GPU is missing."""
from __future__ import annotations

import ctypes
import functools
import logging
from typing import Dict, List, Literal, Tuple

import numpy as np

from . import _lib
from .corpus import Corpus, _u32, _u64
from .preprocess import full_process, pack_prepared

logger = logging.getLogger(__name__)


def weighted_score(partial_int: int, token_set_int: int) -> float:
    """reference app.py:1023-1032 on the two integer scorer results."""
    partial_score = partial_int / 100.0
    token_set_score = token_set_int / 100.0
    score = (partial_score * 0.5) + (token_set_score * 0.5)
    return min(score, 0.99)


@functools.lru_cache(maxsize=16)
def rank_table(threshold: float):
    """-> (rank_tab uint16 [101*101], values list (rank r has value values[r-1]), min_rank).
    About 30 000 Python float evaluations: cached per threshold, not redone per query."""
    vals = sorted({weighted_score(a, b) for a in range(101) for b in range(101)})
    rank_of = {v: i + 1 for i, v in enumerate(vals)}
    tab = np.zeros(101 * 101, dtype=np.uint16)
    for a in range(101):
        for b in range(101):
            tab[a * 101 + b] = rank_of[weighted_score(a, b)]
    tab.setflags(write=False)                          # shared by every caller of the cache
    passing = [rank_of[v] for v in vals if v >= threshold]
    return tab, vals, (min(passing) if passing else 0x10000)


class _Entity:
    """One entity list: the normalised-key dict of the reference plus its two device corpora."""

    def __init__(self, mapping: Dict[str, str], device: int):
        self.mapping = mapping
        self.keys = list(mapping.keys())                 # dict order = first-insertion order
        self.originals = list(mapping.values())
        self.raw = Corpus.from_prepared(self.keys, device=device)                      # ratio, partial_ratio
        self.processed = Corpus.from_prepared([full_process(k) for k in self.keys], device=device)  # token_set_ratio

    def close(self):
        self.raw.close()
        self.processed.close()


class WeightedFuzzySearch:
    """A robust system for fuzzy matching (reference app.py:949-956)."""

    def __init__(self, products: List[str], customers: List[str], threshold: float = 0.70,
                 exact_match_threshold: float = 0.98, top_n: int = 10, *, device: int = 0):
        self.products = products
        self.customers = customers
        self.threshold = threshold
        self.exact_match_threshold = exact_match_threshold
        self.top_n = top_n
        self.products_normalized = {self.normalize_string(p): p for p in self.products}
        self.customers_normalized = {self.normalize_string(c): c for c in self.customers}
        self._lib = _lib.load()
        self._entities = {"product": _Entity(self.products_normalized, device),
                          "customer": _Entity(self.customers_normalized, device)}
        logger.info(f"FuzzySearch initialized with {len(products)} products and {len(customers)} customers.")

    def normalize_string(self, s: str) -> str:
        """Normalizes a string by lowercasing and replacing look-alike characters."""
        if not isinstance(s, str):
            return ""
        return s.lower().replace('×', 'x').replace('×', 'x').strip()

    # -- app.py:1004-1012 ---------------------------------------------------------------------------
    def _check_exact_match(self, query_normalized: str, ent: _Entity):
        if query_normalized in ent.mapping:
            return ent.mapping[query_normalized]
        limit = self.exact_match_threshold * 100
        min_score = next((s for s in range(101) if s > limit), 101)      # fuzz.ratio(...) > limit
        if min_score > 100 or not ent.keys:
            return None
        idx, _, cnt = ent.raw.topk([query_normalized], min_score, 1, _lib.RATIO | _lib.ORDER_BY_INDEX)
        return ent.originals[int(idx[0, 0])] if int(cnt[0]) else None

    # -- app.py:981-1002 ----------------------------------------------------------------------------
    def find_best_matches(self, query: str, entity_type: Literal["product", "customer"]) -> List[Tuple[str, float]]:
        ent = self._entities["product" if entity_type == "product" else "customer"]
        query_normalized = self.normalize_string(query)
        exact_match = self._check_exact_match(query_normalized, ent)
        if exact_match:
            logger.info(f"Exact match found for '{query}': '{exact_match}'")
            return [(exact_match, 1.0)]
        top_n = self.top_n
        n = len(ent.keys)
        if n == 0:
            return []
        if top_n is not None and top_n < 0:              # a negative slice drops from the end: rank everything
            k = n
        else:
            k = n if top_n is None else min(top_n, n)
        # a key equal to the query scores exactly 1.0 (app.py:1019-1020); it only gets here when its
        # original string is falsy (the `if exact_match:` above)
        equal_pos = ent.keys.index(query_normalized) if query_normalized in ent.mapping else -1
        tab, vals, min_rank = rank_table(self.threshold)
        matches: List[Tuple[str, float]] = []
        kk = k + (1 if equal_pos >= 0 else 0)
        if kk > _lib.MAX_K:
            # the reference would return every passing candidate; the device top-k holds FZ_MAX_K.  Same
            # behaviour as FuzzySearch / Corpus.topk: an explicit error, never a silently shorter list.
            raise _lib.FuzzyHipError(f"top_n={top_n!r} over {n} candidates needs k={kk}, beyond FZ_MAX_K={_lib.MAX_K}")
        if kk > 0:
            qa_cps, qa_offs = pack_prepared([query_normalized])
            qb_cps, qb_offs = pack_prepared([full_process(query_normalized)])
            idx = np.zeros((1, kk), dtype=np.uint64)
            rank = np.zeros((1, kk), dtype=np.int32)
            cnt = np.zeros(1, dtype=np.uint32)
            _lib.check(self._lib.fz_query_topk_pair(
                ent.raw._handle(), _lib.PARTIAL, _u32(qa_cps), _u64(qa_offs),
                ent.processed._handle(), _lib.TOKEN_SET, _u32(qb_cps), _u64(qb_offs), 1,
                tab.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), int(min_rank), int(kk),
                _u64(idx), rank.ctypes.data_as(_lib.c_i32p), _u32(cnt)))
            matches = [(ent.originals[int(idx[0, j])], vals[int(rank[0, j]) - 1], int(idx[0, j]))
                       for j in range(int(cnt[0]))]
        if equal_pos >= 0:
            matches = [m for m in matches if m[2] != equal_pos]
            if 1.0 >= self.threshold:
                matches.insert(0, (ent.originals[equal_pos], 1.0, equal_pos))
        out = [(m[0], m[1]) for m in matches]
        return out[:top_n]

    def close(self) -> None:
        for e in self._entities.values():
            e.close()
        self._entities = {}
