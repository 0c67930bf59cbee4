"""``Corpus``: a candidate list resident in HBM behind an opaque ``fz_corpus*`` handle.

Thin, typed wrapper over the C ABI (``include/fuzzyhip.h``): all scoring happens in
``libfuzzyhip.so`` on the GPU.  Used by ``FuzzySearch`` (the reference's class, one corpus per entity
list) and by ``bench.py`` / the tests directly."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .preprocess import pack_prepared, prepare, prepare_and_pack


def _u32(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_u32p)


def _u64(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_u64p)


class Corpus:
    """Candidates of one shard, tokenised, canonicalised and uploaded once (``fz_corpus_create``)."""

    def __init__(self, cps: np.ndarray, offs: np.ndarray, device: int = 0, base_index: int = 0):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        cps = np.ascontiguousarray(cps, dtype=np.uint32)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        if offs.ndim != 1 or offs.size < 1:
            raise ValueError("offs must hold n+1 offsets")
        self.n = int(offs.size - 1)
        self.device = int(device)
        self.base_index = int(base_index)
        if cps.size == 0:
            cps = np.zeros(1, dtype=np.uint32)
        _lib.check(self._lib.fz_corpus_create(_u32(cps), _u64(offs), self.n, self.device,
                                              self.base_index, ctypes.byref(self._h)))

    # -- construction helpers ------------------------------------------------------------------
    @classmethod
    def from_strings(cls, items: Sequence[str], device: int = 0, base_index: int = 0) -> "Corpus":
        """Raw application strings: applies the reference's lower-casing + thefuzz ``full_process``."""
        cps, offs = prepare_and_pack(items)
        return cls(cps, offs, device=device, base_index=base_index)

    @classmethod
    def from_prepared(cls, items: Sequence[str], device: int = 0, base_index: int = 0) -> "Corpus":
        cps, offs = pack_prepared(items)
        return cls(cps, offs, device=device, base_index=base_index)

    # -- packed corpus file ----------------------------------------------------
    def save(self, path: str) -> None:
        """Write the built device layout to `path` (``fz_corpus_save``)."""
        _lib.check(self._lib.fz_corpus_save(self._handle(), str(path).encode()))

    @classmethod
    def load(cls, path: str, device: int = 0, base_index: Optional[int] = None) -> "Corpus":
        """Read + upload a file written by ``save``; ``base_index=None`` keeps the saved one."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.device = int(device)
        _lib.check(self._lib.fz_corpus_load(str(path).encode(), self.device,
                                            (1 << 64) - 1 if base_index is None else int(base_index),
                                            ctypes.byref(self._h)))
        inf = self.info()
        self.n = int(inf["n"])
        self.base_index = int(inf["base_index"])
        return self

    # -- lifetime ------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.fz_corpus_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _handle(self):
        if not self._h.value:
            raise RuntimeError("corpus is closed")
        return self._h

    # -- queries -------------------------------------------------------------------------------
    @staticmethod
    def _pack_queries(queries) -> Tuple[np.ndarray, np.ndarray, int]:
        if isinstance(queries, tuple) and len(queries) == 2 and isinstance(queries[0], np.ndarray):
            q_cps = np.ascontiguousarray(queries[0], dtype=np.uint32)
            q_offs = np.ascontiguousarray(queries[1], dtype=np.uint64)
            if q_cps.size == 0:
                q_cps = np.zeros(1, dtype=np.uint32)
            return q_cps, q_offs, int(q_offs.size - 1)
        q_cps, q_offs = pack_prepared(list(queries))
        return q_cps, q_offs, int(q_offs.size - 1)

    def topk(self, queries, min_score: int, k: int, scorer: int = _lib.TOKEN_SET, noalloc: bool = False):
        """Prepared queries (list of str, or (cps, offs)) -> (idx[nq,k] u64, score[nq,k] i32, cnt[nq]).
        ``noalloc=True`` goes through ``fz_query_topk_noalloc``: same results, no device allocation
        (static scratch array; sizes beyond its limits raise)."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        idx = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        score = np.zeros((nq, max(k, 1)), dtype=np.int32)
        cnt = np.zeros(max(nq, 1), dtype=np.uint32)
        fn = self._lib.fz_query_topk_noalloc if noalloc else self._lib.fz_query_topk
        _lib.check(fn(self._handle(), _u32(q_cps), _u64(q_offs), nq, int(scorer), int(min_score), int(k),
                      _u64(idx), score.ctypes.data_as(_lib.c_i32p), _u32(cnt)))
        return idx[:, :k], score[:, :k], cnt[:nq]

    def topk_submit(self, queries, min_score: int, k: int, scorer: int = _lib.TOKEN_SET) -> "TopkTicket":
        """First half of ``topk`` (``fz_query_topk_submit``): tables built, every kernel enqueued, nothing waited for.
        At most ``MAX_IN_FLIGHT`` tickets per corpus; fetch each with ``topk_wait``.  The ticket keeps the query arrays
        alive (the library reads them again if a fused list overflows)."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        t = ctypes.c_int(-1)
        _lib.check(self._lib.fz_query_topk_submit(self._handle(), _u32(q_cps), _u64(q_offs), nq, int(scorer),
                                                  int(min_score), int(k), ctypes.byref(t)))
        return TopkTicket(t.value, nq, int(k), q_cps, q_offs)

    def topk_wait(self, ticket: "TopkTicket"):
        """Second half (``fz_query_topk_wait``): blocks until the call's result is there; same return value as ``topk``."""
        if ticket.done:
            raise RuntimeError("ticket already fetched")
        nq, k = ticket.nq, ticket.k
        idx = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        score = np.zeros((nq, max(k, 1)), dtype=np.int32)
        cnt = np.zeros(max(nq, 1), dtype=np.uint32)
        ticket.done = True
        _lib.check(self._lib.fz_query_topk_wait(self._handle(), ticket.slot, _u64(idx), score.ctypes.data_as(_lib.c_i32p), _u32(cnt)))
        ticket.q_cps = ticket.q_offs = None
        return idx[:, :k], score[:, :k], cnt[:nq]

    def topk_sharded(self, comm_handle: int, queries, min_score: int, k: int, scorer: int = _lib.TOKEN_SET):
        """Collective: top-k over the WHOLE sharded list (this corpus is one shard).  The exchange -- RCCL
        all-gather of the packed keys and the device merge -- runs inside the library
        (``fz_query_topk_sharded``); every rank passes the same queries and gets the same global result."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        idx = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        score = np.zeros((nq, max(k, 1)), dtype=np.int32)
        cnt = np.zeros(max(nq, 1), dtype=np.uint32)
        _lib.check(self._lib.fz_query_topk_sharded(self._handle(), ctypes.c_void_p(comm_handle), _u32(q_cps), _u64(q_offs),
                                                   nq, int(scorer), int(min_score), int(k), _u64(idx),
                                                   score.ctypes.data_as(_lib.c_i32p), _u32(cnt)))
        return idx[:, :k], score[:, :k], cnt[:nq]

    def topk_keys_device(self, queries, min_score: int, k: int, keys_ptr: int, cnt_ptr: int = 0,
                         scorer: int = _lib.TOKEN_SET, sync: bool = True) -> int:
        """Per-shard top-k as packed keys written to DEVICE memory (for the RCCL all-gather)."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        _lib.assert_single_runtime("Corpus.topk_keys_device (a device pointer made by other code)")
        _lib.check(self._lib.fz_query_topk_keys(self._handle(), _u32(q_cps), _u64(q_offs), nq,
                                                int(scorer), int(min_score), int(k),
                                                ctypes.c_void_p(keys_ptr),
                                                ctypes.c_void_p(cnt_ptr) if cnt_ptr else None,
                                                1 if sync else 0))
        return nq

    def score_all(self, queries, scorer: int = _lib.TOKEN_SET) -> np.ndarray:
        """Every integer score, original candidate order: uint8 [nq, n]."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        out = np.zeros((max(nq, 1), max(self.n, 1)), dtype=np.uint8)
        _lib.check(self._lib.fz_score_all(self._handle(), _u32(q_cps), _u64(q_offs), nq, int(scorer),
                                          out.ctypes.data_as(_lib.c_u8p)))
        return out[:nq, :self.n]

    def distance_all(self, queries, kind: int = 0) -> np.ndarray:
        """Raw distances against every candidate as given (no canonicalisation), original order:
        uint16 [nq, n].  kind 0 = InDel, 1 = Levenshtein (Myers bit-vector kernel), 2 = common tokens,
        3 = substring, 4 = matching characters of difflib.SequenceMatcher (0xFFFF: candidate beyond 64 symbols)."""
        q_cps, q_offs, nq = self._pack_queries(queries)
        out = np.zeros((max(nq, 1), max(self.n, 1)), dtype=np.uint16)
        _lib.check(self._lib.fz_distance_all(self._handle(), _u32(q_cps), _u64(q_offs), nq, int(kind),
                                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))))
        return out[:nq, :self.n]

    # -- introspection -------------------------------------------------------------------------
    def info(self) -> dict:
        inf = _lib.FzInfo()
        _lib.check(self._lib.fz_corpus_info(self._handle(), ctypes.byref(inf)))
        return {name: getattr(inf, name) for name, _ in inf._fields_}

    def timing(self) -> dict:
        t = _lib.FzTiming()
        _lib.check(self._lib.fz_last_timing(self._handle(), ctypes.byref(t)))
        return {name: getattr(t, name) for name, _ in t._fields_}

    def set_stream(self, stream_ptr: Optional[int]) -> None:
        """Later kernels run on the caller's hipStream_t (None: the corpus' own).  The stream must belong to the HIP
        runtime the library is bound to: with torch in the process, torch must have been imported first (_lib._RULE)."""
        if stream_ptr:
            _lib.assert_single_runtime("Corpus.set_stream")
        _lib.check(self._lib.fz_set_stream(self._handle(), ctypes.c_void_p(stream_ptr or 0)))

    def set_workspace_limit(self, nbytes: int) -> None:
        _lib.check(self._lib.fz_set_workspace_limit(self._handle(), int(nbytes)))


MAX_IN_FLIGHT = 2          # FZ_MAX_IN_FLIGHT (include/fuzzyhip.h)


class TopkTicket:
    """A submitted, not yet fetched ``Corpus.topk_submit`` call."""
    __slots__ = ("slot", "nq", "k", "q_cps", "q_offs", "done")

    def __init__(self, slot: int, nq: int, k: int, q_cps, q_offs):
        self.slot, self.nq, self.k, self.q_cps, self.q_offs, self.done = slot, nq, k, q_cps, q_offs, False


def list_fingerprint(items: Sequence[str]):
    """What a drop-in compares to decide whether the list it uploaded is still the list it is handed.
    The reference re-reads its lists on every call (src/fuzzy_search.py:17-18), so an edit made in place
    must be seen: lists of up to 4 096 items are hashed in full (Python caches each string's hash, so
    this costs tens of microseconds), longer ones by identity, length and 256 evenly spaced items --
    after an in-place edit of a longer list call the owner's ``refresh()``."""
    n = len(items)
    if n <= 4096:
        return (id(items), n, hash(tuple(items)))
    step = max(1, n // 256)
    return (id(items), n, hash(tuple(items[i] for i in range(0, n, step))), hash(items[-1]))


class LeasedResource:
    """A device resource (anything with ``close()``) shared by concurrent callers: ``lease()`` hands it
    out counted, ``retire()`` closes it once the last lease is back.  This is what lets a drop-in
    replace a corpus whose list changed while another session thread is still inside a query on it
    (the reference's object is shared by every Streamlit session, utils/helper_components.py:59-64)."""

    def __init__(self, resource):
        import threading
        self.resource = resource
        self._lock = threading.Lock()
        self._users = 0
        self._retired = False

    def lease(self):
        with self._lock:
            if self._retired:
                return None
            self._users += 1
            return self.resource

    def release(self) -> None:
        with self._lock:
            self._users -= 1
            close_now = self._retired and self._users == 0
        if close_now:
            self.resource.close()

    def retire(self) -> None:
        with self._lock:
            already = self._retired
            self._retired = True
            close_now = not already and self._users == 0
        if close_now:
            self.resource.close()


def distance_batch(a: Sequence[str], b: Sequence[str], kind: int = 0, device: int = 0) -> np.ndarray:
    """Distances of explicit pairs (a[i], b[i]) through ``fz_distance_batch``: uint32 [n]."""
    if len(a) != len(b):
        raise ValueError("a and b must pair up")
    lib = _lib.load()
    a_cps, a_offs = pack_prepared(list(a))
    b_cps, b_offs = pack_prepared(list(b))
    out = np.zeros(max(len(a), 1), dtype=np.uint32)
    _lib.check(lib.fz_distance_batch(int(device), int(kind), _u32(a_cps), _u64(a_offs), _u32(b_cps),
                                     _u64(b_offs), len(a), _u32(out)))
    return out[:len(a)]


def merge_topk(keys: np.ndarray, n_lists: int, nq: int, k: int):
    """Deterministic merge of gathered per-shard key lists [n_lists, nq, k] (``fz_merge_topk``)."""
    lib = _lib.load()
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
    if keys.size != n_lists * nq * k:
        raise ValueError("keys must hold n_lists*nq*k entries")
    if keys.size == 0:
        keys = np.zeros(1, dtype=np.uint64)
    idx = np.zeros((nq, max(k, 1)), dtype=np.uint64)
    score = np.zeros((nq, max(k, 1)), dtype=np.int32)
    cnt = np.zeros(max(nq, 1), dtype=np.uint32)
    _lib.check(lib.fz_merge_topk(_u64(keys), n_lists, nq, k, _u64(idx),
                                 score.ctypes.data_as(_lib.c_i32p), _u32(cnt)))
    return idx[:, :k], score[:, :k], cnt[:nq]
