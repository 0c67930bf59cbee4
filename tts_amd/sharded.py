"""Multi-GPU path: one process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI),
the candidate list sharded by contiguous global index ranges, queries replicated.

The path has exactly one exchange step: each rank's top-k keys -- nq*k u64, a few
hundred KB at most, latency-bound -- are all-gathered, then every rank runs the same deterministic
merge (``fz_merge_topk``: descending packed key = score desc, global index asc), so 1/2/4/8-GPU
results are identical by construction.  The reference has no counterpart (it is single-process)."""
from __future__ import annotations

from typing import List, Literal, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .config import FUZZY_SEARCH_THRESHOLD_FLOAT, TOP_N_MATCHES_INT
from .corpus import Corpus, merge_topk
from .preprocess import min_score_for_threshold, prepare


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous original-index range of `rank`: base_g = g * ceil(n / G)  (SURVEY.md 8e)."""
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def gather_and_merge(keys, nq: int, k: int):
    """keys: this rank's [nq, k] packed keys as an int64 torch tensor (CPU for gloo, device for
    nccl).  All-gather over the default process group, then the library's merge."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    flat = keys.reshape(-1).contiguous()
    if dist.get_backend() == "gloo" and flat.is_cuda:        # CPU rehearsal of the exchange
        flat = flat.cpu()
    gathered = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(gathered, flat)
    host = gathered.cpu().numpy().view(np.uint64)
    return merge_topk(host, world, nq, k)


class ShardedFuzzySearch:
    """``FuzzySearch`` semantics over a candidate list sharded across the ranks of the default
    process group; every rank passes the FULL lists and keeps only its index range on its GPU."""

    def __init__(self, products: List[str], customers: List[str],
                 threshold: float = FUZZY_SEARCH_THRESHOLD_FLOAT, *, device: Optional[int] = None,
                 top_n: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.products, self.customers, self.threshold = products, customers, threshold
        self._top_n = TOP_N_MATCHES_INT if top_n is None else int(top_n)
        self._rank, self._world = dist.get_rank(), dist.get_world_size()
        self._device = torch.cuda.current_device() if device is None else device
        self._corpora = {}
        for kind, items in (("product", products), ("customer", customers)):
            lo, hi = shard_bounds(len(items), self._world, self._rank)
            self._corpora[kind] = Corpus.from_strings(items[lo:hi], device=self._device, base_index=lo)

    def find_best_matches_many(self, queries: Sequence[str], entity_type: Literal["product", "customer"]):
        import torch
        kind = "product" if entity_type == "product" else "customer"
        items = self.products if kind == "product" else self.customers
        corpus = self._corpora[kind]
        prepared = [prepare(q) for q in queries]
        nq, k = len(prepared), max(self._top_n, 0)
        if nq == 0 or k == 0:
            return [[] for _ in prepared]
        keys = torch.zeros(nq * k, dtype=torch.int64, device=torch.device("cuda", self._device))
        corpus.topk_keys_device(prepared, min_score_for_threshold(self.threshold), k, keys.data_ptr(), 0,
                                _lib.TOKEN_SET, sync=True)
        idx, score, cnt = gather_and_merge(keys, nq, k)
        return [[(items[int(idx[q, j])], int(score[q, j]) / 100.0) for j in range(int(cnt[q]))] for q in range(nq)]

    def find_best_matches(self, query: str, entity_type: Literal["product", "customer"]):
        return self.find_best_matches_many([query], entity_type)[0]

    def close(self):
        for c in self._corpora.values():
            c.close()
        self._corpora = {}
