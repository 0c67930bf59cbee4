"""Drop-in for the reference's ``src/fuzzy_search.py``: same class, same two signatures, same
return values -- the candidate-scoring loop (reference ``src/fuzzy_search.py:22-38``) runs on an
MI355X through ``libfuzzyhip.so``.

    reference                                     here
    FuzzySearch.__init__            :10-14        stores the lists, uploads one Corpus per list
    find_best_matches               :16-38        prepare query -> fz_query_topk -> map idx -> item
      query.lower() / item.lower()  :19,26        preprocess.prepare (query per call, items once)
      fuzz.token_set_ratio          :26           fz_score kernels (bit-exact integer score)
      score/100.0 >= threshold      :29-31        integer min_score computed with Python floats
      stable sort desc + [:TOP_N]   :35-38        fz_select_topk: (score desc, list index asc)

There is no CPU path: construction raises if the library or a GPU is missing."""
from __future__ import annotations

import logging
from typing import List, Literal, Optional, Sequence, Tuple

from . import _lib
from .config import FUZZY_SEARCH_THRESHOLD_FLOAT, TOP_N_MATCHES_INT
from .corpus import Corpus
from .preprocess import min_score_for_threshold, prepare

logger = logging.getLogger(__name__)


class FuzzySearch:
    """A system for performing fuzzy matching on product and customer names."""

    def __init__(self, products: List[str], customers: List[str],
                 threshold: float = FUZZY_SEARCH_THRESHOLD_FLOAT, *, device: int = 0,
                 top_n: Optional[int] = None):
        self.products = products
        self.customers = customers
        self.threshold = threshold
        self._device = device
        self._top_n = TOP_N_MATCHES_INT if top_n is None else int(top_n)
        self._corpora = {}
        self._corpus_for("product")
        self._corpus_for("customer")
        logger.info(f"FuzzySearch initialized with {len(products)} products and {len(customers)} customers.")

    # the reference re-reads self.products / self.customers on every call; rebuild if the list
    # object was swapped or changed size, otherwise the upload made at construction is reused
    def _corpus_for(self, kind: str) -> Tuple[Sequence[str], Corpus]:
        items = self.products if kind == "product" else self.customers
        ent = self._corpora.get(kind)
        if ent is None or ent[0] is not items or ent[1].n != len(items):
            if ent is not None:
                ent[1].close()
            ent = (items, Corpus.from_strings(items, device=self._device))
            self._corpora[kind] = ent
        return ent

    def find_best_matches(self, query: str, entity_type: Literal["product", "customer"]) -> List[Tuple[str, float]]:
        """Find the best matching items for a query using a more advanced ratio."""
        return self.find_best_matches_many([query], entity_type)[0]

    def find_best_matches_many(self, queries: Sequence[str],
                               entity_type: Literal["product", "customer"]) -> List[List[Tuple[str, float]]]:
        """All entities of one user message in a single pass over the corpus;
        element i equals ``find_best_matches(queries[i], entity_type)``."""
        kind = "product" if entity_type == "product" else "customer"   # reference :18
        items, corpus = self._corpus_for(kind)
        prepared = [prepare(q) for q in queries]
        min_score = min_score_for_threshold(self.threshold)
        idx, score, cnt = corpus.topk(prepared, min_score, max(self._top_n, 0), _lib.TOKEN_SET)
        out = []
        for qi in range(len(prepared)):
            out.append([(items[int(idx[qi, j])], int(score[qi, j]) / 100.0) for j in range(int(cnt[qi]))])
        return out

    def close(self) -> None:
        for _, corpus in self._corpora.values():
            corpus.close()
        self._corpora = {}
