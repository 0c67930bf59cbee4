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
import threading
from typing import List, Literal, Optional, Sequence, Tuple

from . import _lib
from .config import FUZZY_SEARCH_THRESHOLD_FLOAT, TOP_N_MATCHES_INT
from .corpus import Corpus, LeasedResource, list_fingerprint
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
        self._build_lock = threading.Lock()
        for kind in ("product", "customer"):
            self._lease(kind)[2].release()
        logger.info(f"FuzzySearch initialized with {len(products)} products and {len(customers)} customers.")

    # The reference re-reads self.products / self.customers on every call (:17-18).  Here the upload made
    # at construction is reused while the list is unchanged (corpus.list_fingerprint: swapped object,
    # changed length, edited items) and rebuilt otherwise; a corpus that another thread is still
    # querying is closed only when that query returns (corpus.LeasedResource).
    def _lease(self, kind: str):
        items = self.products if kind == "product" else self.customers
        fp = list_fingerprint(items)
        with self._build_lock:
            ent = self._corpora.get(kind)
            corpus = ent[2].lease() if ent is not None and ent[0] == fp else None
            if corpus is None:
                if ent is not None:
                    ent[2].retire()
                # the strings are copied: results map back to the items the scores were computed from
                snapshot = list(items)
                ent = (fp, snapshot, LeasedResource(Corpus.from_strings(snapshot, device=self._device)))
                self._corpora[kind] = ent
                ent[2].lease()
            return ent[1], ent[2].resource, ent[2]

    def refresh(self) -> None:
        """Re-upload both lists now (after an in-place edit of a list longer than 4 096 items, which
        the per-call fingerprint only samples)."""
        with self._build_lock:
            for ent in self._corpora.values():
                ent[2].retire()
            self._corpora = {}
        for kind in ("product", "customer"):
            self._lease(kind)[2].release()

    def find_best_matches(self, query: str, entity_type: Literal["product", "customer"]) -> List[Tuple[str, float]]:
        """Find the best matching items for a query using a more advanced ratio."""
        return self.find_best_matches_many([query], entity_type)[0]

    def find_best_matches_many(self, queries: Sequence[str],
                               entity_type: Literal["product", "customer"]) -> List[List[Tuple[str, float]]]:
        """All entities of one user message in a single pass over the corpus;
        element i equals ``find_best_matches(queries[i], entity_type)``."""
        kind = "product" if entity_type == "product" else "customer"   # reference :18
        items, corpus, lease = self._lease(kind)
        try:
            prepared = [prepare(q) for q in queries]
            min_score = min_score_for_threshold(self.threshold)
            idx, score, cnt = corpus.topk(prepared, min_score, max(self._top_n, 0), _lib.TOKEN_SET)
        finally:
            lease.release()
        out = []
        for qi in range(len(prepared)):
            out.append([(items[int(idx[qi, j])], int(score[qi, j]) / 100.0) for j in range(int(cnt[qi]))])
        return out

    def close(self) -> None:
        with self._build_lock:
            for ent in self._corpora.values():
                ent[2].retire()
            self._corpora = {}
