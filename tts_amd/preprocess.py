"""Host-side string preparation for the hot path: what the reference and the thefuzz wrapper do to
each string before the scorer sees it (reference ``src/fuzzy_search.py:19,26``; SURVEY.md rows
a2, a4, a5).  It stays in Python because it depends on Python's own Unicode tables
(``str.lower``, ``str.isalnum``); the C ABI receives code points that are already processed.

ASSUMPTION, isolated in ``_NON_TOKEN``: rapidfuzz's ``default_process`` replaces every character
that is not ``str.isalnum()`` -- including ``'_'`` -- by a space (the compiled wheel's table);
SURVEY.md section 8c lists this as the low-confidence item."""
from __future__ import annotations

import re
from typing import Iterable, List, Sequence, Tuple

import numpy as np

# thefuzz utils.ascii_only: str.translate with {i: None for i in range(128, 256)}
_DROP_LATIN1 = {i: None for i in range(128, 256)}
# \W is "not (isalnum or '_')" for str patterns, so [\W_] is exactly "not isalnum"
_NON_TOKEN = re.compile(r"[\W_]")

# one-shot byte table for pure-ASCII input: lower-case letters and digits stay, A-Z fold, rest -> ' '
_ASCII_TABLE = bytes(
    (c + 32) if 65 <= c <= 90 else (c if (48 <= c <= 57 or 97 <= c <= 122) else 32) for c in range(256))


def full_process(s: str) -> str:
    """thefuzz ``full_process(s, force_ascii=True)``: drop U+0080..U+00FF, non-alphanumerics to
    spaces, lower-case, strip."""
    return _NON_TOKEN.sub(" ", s.translate(_DROP_LATIN1)).lower().strip()


def prepare(s: str) -> str:
    """A string as the scorer sees it on the reference path: ``s.lower()`` by the reference
    (src/fuzzy_search.py:19,26), then thefuzz's ``full_process``."""
    if s.isascii():
        return s.encode("ascii").translate(_ASCII_TABLE).decode("ascii").strip()
    return full_process(s.lower())


def min_score_for_threshold(threshold: float) -> int:
    """Smallest integer score s with ``s / 100.0 >= threshold`` evaluated in Python floats, exactly
    the comparison of src/fuzzy_search.py:29-31; 101 if no score in 0..100 passes."""
    for s in range(0, 101):
        if s / 100.0 >= threshold:
            return s
    return 101


def pack_prepared(strings: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Already-prepared strings -> (UTF-32 code points uint32, offsets uint64[n+1])."""
    n = len(strings)
    offs = np.zeros(n + 1, dtype=np.uint64)
    if n:
        lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=n)
        np.cumsum(lens, out=offs[1:])
    joined = "".join(strings)
    if joined.isascii():
        buf = np.frombuffer(joined.encode("ascii"), dtype=np.uint8).astype(np.uint32)
    else:
        buf = np.frombuffer(joined.encode("utf-32-le"), dtype=np.uint32).copy()
    if buf.size == 0:
        buf = np.zeros(1, dtype=np.uint32)
    return np.ascontiguousarray(buf), offs


def prepare_and_pack(items: Iterable[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Prepare every candidate and pack the lot.  Pure-ASCII lists (the common case for product and
    customer names) go through one ``bytes.translate`` over the joined text; anything else falls
    back to per-string ``prepare`` so that Python's Unicode semantics hold exactly."""
    items = list(items)
    n = len(items)
    joined = "\n".join(items)
    if joined.isascii():
        # ASCII lower-casing keeps lengths, so offsets follow from the original lengths; the
        # separator (and any newline inside an item) becomes a space, which tokenisation ignores
        lens = np.fromiter((len(s) for s in items), dtype=np.uint64, count=n)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            np.cumsum(lens + np.uint64(1), out=offs[1:])
        raw = (joined + "\n").encode("ascii").translate(_ASCII_TABLE)
        buf = np.frombuffer(raw, dtype=np.uint8).astype(np.uint32)
        if n == 0:
            buf = np.zeros(1, dtype=np.uint32)
        return np.ascontiguousarray(buf), offs
    return pack_prepared([prepare(s) for s in items])
