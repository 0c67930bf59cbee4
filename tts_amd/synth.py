"""Seeded synthetic corpora and queries of BASELINE.md section 3 (there is no network for real data).

Names are 1-5 tokens of length 2-9 over ``[a-z0-9]``, tokens drawn Zipf(1.2) from a 50k-token
vocabulary (so token intersections really occur), joined by single spaces and clipped to the
configuration's maximum length.  Queries are corpus names with 1-3 random edits.  Everything is
generated directly in prepared form (lower-case alphanumerics and spaces), as packed UTF-32 code
points + offsets, so a 10^6..10^8-name corpus never exists as Python strings."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

CORPUS_SEED = 20251017
QUERY_SEED = 20251018
_ALPHABET = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)


def make_vocab(rng: np.random.Generator, size: int = 50_000, min_len: int = 2, max_len: int = 9):
    lens = rng.integers(min_len, max_len + 1, size=size)
    chars = _ALPHABET[rng.integers(0, _ALPHABET.size, size=(size, max_len))]
    return chars, lens


def zipf_ranks(rng: np.random.Generator, n: int, vocab: int, a: float = 1.2) -> np.ndarray:
    p = np.arange(1, vocab + 1, dtype=np.float64) ** (-a)
    cdf = np.cumsum(p / p.sum())
    return np.minimum(np.searchsorted(cdf, rng.random(n)), vocab - 1)


def make_corpus(n: int, max_chars: int = 64, seed: int = CORPUS_SEED, vocab_size: int = 50_000,
                min_tokens: int = 1, max_tokens: int = 5, tok_min: int = 2, tok_max: int = 9
                ) -> Tuple[np.ndarray, np.ndarray]:
    """-> (cps uint32, offs uint64[n+1]); string i = cps[offs[i]:offs[i+1]]."""
    rng = np.random.default_rng(seed)
    vchars, vlens = make_vocab(rng, vocab_size, tok_min, tok_max)
    ntok = rng.integers(min_tokens, max_tokens + 1, size=n)
    ranks = zipf_ranks(rng, n * max_tokens, vocab_size).reshape(n, max_tokens)
    tlen = vlens[ranks]                                          # [n, T]
    # clip: drop trailing tokens until the name fits max_chars
    for t in range(max_tokens):
        tlen[:, t] = np.where(t < ntok, tlen[:, t], 0)
    while True:
        used = (tlen > 0).sum(axis=1)
        total = tlen.sum(axis=1) + np.maximum(used - 1, 0)
        over = (total > max_chars) & (used > 1)
        if not over.any():
            break
        last = used - 1
        tlen[over, last[over]] = 0
    used = (tlen > 0).sum(axis=1)
    # a single over-long token cannot happen for tok_max <= max_chars
    total = tlen.sum(axis=1) + np.maximum(used - 1, 0)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(total, out=offs[1:])
    cps = np.full(int(offs[-1]), 32, dtype=np.uint32)            # spaces; tokens overwrite
    start = offs[:-1].astype(np.int64).copy()
    for t in range(max_tokens):
        lt = tlen[:, t]
        for j in range(tok_max):
            m = lt > j
            if not m.any():
                break
            cps[start[m] + j] = vchars[ranks[m, t], j]
        start += lt + (lt > 0)
    return cps, offs


def make_corpus_blocks(b0: int, b1: int, per_block: int, max_chars: int = 64, seed: int = CORPUS_SEED,
                       workers: int = 8, **gen) -> Tuple[np.ndarray, np.ndarray]:
    """Blocks b0..b1-1 of a corpus defined block by block (block b = make_corpus(per_block, seed = seed +
    1000 + b)), concatenated.  The whole corpus is the same whatever range a rank asks for, which is what
    lets bench.py shard ONE 10^8-candidate corpus (C3) over any number of ranks without any rank ever
    generating all of it.  Blocks are generated on a few threads (numpy releases the GIL in its bulk loops)."""
    from concurrent.futures import ThreadPoolExecutor
    if b1 <= b0:
        return np.zeros(0, dtype=np.uint32), np.zeros(1, dtype=np.uint64)
    with ThreadPoolExecutor(max(1, min(workers, b1 - b0))) as ex:
        parts = list(ex.map(lambda b: make_corpus(per_block, max_chars=max_chars, seed=seed + 1000 + b, **gen), range(b0, b1)))
    cps = np.concatenate([p[0] for p in parts])
    offs = np.zeros((b1 - b0) * per_block + 1, dtype=np.uint64)
    base = 0
    for i, (c, o) in enumerate(parts):
        offs[i * per_block + 1:(i + 1) * per_block + 1] = o[1:] + np.uint64(base)
        base += int(o[-1])
    return cps, offs


def to_strings(cps: np.ndarray, offs: np.ndarray, lo: int = 0, hi: int = None) -> List[str]:
    hi = len(offs) - 1 if hi is None else hi
    raw = cps.astype(np.uint8).tobytes().decode("ascii") if cps.max(initial=0) < 128 else None
    out = []
    for i in range(lo, hi):
        a, b = int(offs[i]), int(offs[i + 1])
        out.append(raw[a:b] if raw is not None else "".join(map(chr, cps[a:b])))
    return out


def make_queries(cps: np.ndarray, offs: np.ndarray, nq: int, seed: int = QUERY_SEED,
                 max_chars: int = 64) -> List[str]:
    """Corpus names with 1-3 random edits: char substitute / insert / delete, token drop, token swap."""
    rng = np.random.default_rng(seed)
    n = len(offs) - 1
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    out = []
    picks = rng.integers(0, n, size=nq)
    for qi in range(nq):
        a, b = int(offs[picks[qi]]), int(offs[picks[qi] + 1])
        s = "".join(map(chr, cps[a:b]))
        for _ in range(int(rng.integers(1, 4))):
            op = int(rng.integers(0, 5))
            toks = s.split(" ")
            if op == 0 and s:                                  # substitute
                i = int(rng.integers(0, len(s)))
                if s[i] != " ":
                    s = s[:i] + alpha[int(rng.integers(0, 36))] + s[i + 1:]
            elif op == 1 and len(s) < max_chars:               # insert
                i = int(rng.integers(0, len(s) + 1))
                s = s[:i] + alpha[int(rng.integers(0, 36))] + s[i:]
            elif op == 2 and len(s) > 2:                       # delete
                i = int(rng.integers(0, len(s)))
                if s[i] != " ":
                    s = s[:i] + s[i + 1:]
            elif op == 3 and len(toks) > 1:                    # token drop
                toks.pop(int(rng.integers(0, len(toks))))
                s = " ".join(toks)
            elif op == 4 and len(toks) > 1:                    # token swap
                i = int(rng.integers(0, len(toks) - 1))
                toks[i], toks[i + 1] = toks[i + 1], toks[i]
                s = " ".join(toks)
        out.append(" ".join(s.split())[:max_chars])
    return out


def algorithmic_bytes_per_query(info: dict) -> int:
    """SURVEY.md 8d / BASELINE.md section 4: per (query, candidate) pair the candidate's UTF-32 code points
    (4 B each) + one 4-B offset word + 4 B per token id, counted once per pair regardless of query
    batching.  Summed over the corpus this is the bytes one query 'reads'."""
    return 4 * int(info["total_symbols"]) + 4 * int(info["n"]) + 4 * int(info["total_tokens"])
