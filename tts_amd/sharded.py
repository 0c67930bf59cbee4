"""Multi-GPU path: one process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI),
the candidate list sharded by contiguous global index ranges, queries replicated.

The path has exactly one exchange step: each rank's top-k keys -- nq*k u64, a few
hundred KB at most, latency-bound -- are all-gathered, then every rank runs the same deterministic
merge (descending packed key = score desc, global index asc), so 1/2/4/8-GPU results are identical by
construction.  The reference has no counterpart (it is single-process).

Two carriers of that step:
  exchange="rccl"   (default) inside the library (``LibraryExchange``): ``ncclAllGather`` on the corpus' stream and
                    a device merge kernel; torch is only the launcher that hands the 128-byte communicator
                    id to the ranks (or no torch at all: ``comm_id_via_file``)
  exchange="torch"  ``torch.distributed.all_gather_into_tensor`` + the host merge ``fz_merge_topk`` (also
                    what the gloo tests on the CPU run)"""
from __future__ import annotations

import ctypes
import os
import time
from typing import Callable, List, Literal, Optional, Sequence, Tuple

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


def comm_id_via_file(path: str, rank: int, timeout_s: float = 120.0) -> Callable[[Optional[bytes]], bytes]:
    """A torch-free way to hand rank 0's communicator id to the other ranks: rank 0 writes `path`
    (temporary name + rename), the others wait for it.  Returns the `share` callable LibraryExchange takes."""
    def share(my_id: Optional[bytes]) -> bytes:
        if rank == 0:
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(my_id)
            os.replace(tmp, path)
            return my_id
        t0 = time.monotonic()
        while True:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == _lib.COMM_ID_BYTES:
                    return data
            except OSError:
                pass
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"no communicator id at {path} after {timeout_s} s")
            time.sleep(0.01)
    return share


def comm_id_via_torch() -> Callable[[Optional[bytes]], bytes]:
    """Rank 0's communicator id broadcast over the default torch.distributed group (launch time only)."""
    def share(my_id: Optional[bytes]) -> bytes:
        import torch.distributed as dist
        box = [my_id]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    return share


class LibraryExchange:
    """An RCCL communicator owned by libfuzzyhip (``fz_comm_*``): one per process, on the GPU of its shard.
    `share(id_or_None) -> id` moves rank 0's 128-byte id to every rank (comm_id_via_torch / comm_id_via_file)."""

    def __init__(self, rank: int, world: int, device: int, share: Callable[[Optional[bytes]], bytes]):
        self._lib = _lib.load()
        _lib.assert_single_runtime("LibraryExchange")
        self.rank, self.world, self.device = rank, world, device
        self._h = ctypes.c_void_p()
        my_id = None
        if rank == 0:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            try:
                _lib.check(self._lib.fz_comm_unique_id(buf))
            except _lib.FuzzyHipError:
                share(b"")                                   # the other ranks must not wait for an id that never comes
                raise
            my_id = buf.raw
        comm_id = share(my_id)
        if not isinstance(comm_id, (bytes, bytearray)) or len(comm_id) != _lib.COMM_ID_BYTES:
            raise _lib.FuzzyHipError("the shared communicator id must be %d bytes" % _lib.COMM_ID_BYTES)
        h = ctypes.c_void_p()
        _lib.check(self._lib.fz_comm_create(ctypes.c_char_p(bytes(comm_id)), world, rank, device, ctypes.byref(h)))
        self._h = h

    @property
    def handle(self) -> int:
        if not self._h:
            raise _lib.FuzzyHipError("communicator is closed")
        return self._h.value

    def close(self) -> None:
        if self._h:
            _lib.check(self._lib.fz_comm_destroy(self._h))
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedFuzzySearch:
    """``FuzzySearch`` semantics over a candidate list sharded across the ranks of the default
    process group; every rank passes the FULL lists and keeps only its index range on its GPU."""

    def __init__(self, products: List[str], customers: List[str],
                 threshold: float = FUZZY_SEARCH_THRESHOLD_FLOAT, *, device: Optional[int] = None,
                 top_n: Optional[int] = None, exchange: str = "rccl", comm: Optional[LibraryExchange] = None):
        """exchange="rccl" (default, the exchange BASELINE.json's north_star names): the all-gather and the merge
        run inside the library; pass `comm` (a LibraryExchange made by the launcher, e.g. with comm_id_via_file) to
        keep torch out of the process altogether, otherwise the id is broadcast once over the default
        torch.distributed group (any backend; gloo will do: no torch.cuda call is made here).
        exchange="torch": module docstring.  `device` defaults to LOCAL_RANK."""
        if exchange not in ("torch", "rccl"):
            raise ValueError("exchange must be 'torch' or 'rccl'")
        self.products, self.customers, self.threshold = products, customers, threshold
        self._top_n = TOP_N_MATCHES_INT if top_n is None else int(top_n)
        self._exchange, self._comm = exchange, comm
        if comm is not None:
            self._exchange = "rccl"
            self._rank, self._world = comm.rank, comm.world
            self._device = comm.device if device is None else device
        else:
            import torch.distributed as dist
            self._rank, self._world = dist.get_rank(), dist.get_world_size()
            self._device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
            if exchange == "rccl":
                self._comm = LibraryExchange(self._rank, self._world, self._device, comm_id_via_torch())
        self._corpora = {}
        for kind, items in (("product", products), ("customer", customers)):
            lo, hi = shard_bounds(len(items), self._world, self._rank)
            self._corpora[kind] = Corpus.from_strings(items[lo:hi], device=self._device, base_index=lo)

    def find_best_matches_many(self, queries: Sequence[str], entity_type: Literal["product", "customer"]):
        kind = "product" if entity_type == "product" else "customer"
        items = self.products if kind == "product" else self.customers
        corpus = self._corpora[kind]
        prepared = [prepare(q) for q in queries]
        nq, k = len(prepared), max(self._top_n, 0)
        if nq == 0 or k == 0:
            return [[] for _ in prepared]
        if self._exchange == "rccl":                      # no torch on this path
            idx, score, cnt = corpus.topk_sharded(self._comm.handle, prepared, min_score_for_threshold(self.threshold), k)
            return [[(items[int(idx[q, j])], int(score[q, j]) / 100.0) for j in range(int(cnt[q]))] for q in range(nq)]
        import torch
        _lib.assert_single_runtime("ShardedFuzzySearch(exchange='torch')")     # a torch tensor's pointer goes to the library
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
        if self._comm is not None and self._exchange == "rccl":
            self._comm.close()
