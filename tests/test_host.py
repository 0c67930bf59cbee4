"""Host-side logic and the C-ABI surface, without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from tts_amd import _lib, merge_topk
from tts_amd import preprocess as P

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fuzzyhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fz_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fuzzyhip.h but not exported"
    assert declared == set(_lib.exported_symbols())
    assert lib.fz_version() == 100


def test_no_cpu_fallback_without_device():
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.fz_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    from tts_amd import Corpus, FuzzySearch
    with pytest.raises(_lib.FuzzyHipError):
        Corpus.from_prepared(["abc"])
    with pytest.raises(_lib.FuzzyHipError):
        FuzzySearch(["a"], ["b"])


def test_product_preprocess_matches_oracle_restatement(oracle):
    with open(os.path.join(HERE, "golden", "preprocess.json")) as f:
        cases = json.load(f)
    for c in cases:
        assert P.prepare(c["raw"]) == c["prepared"], c
        assert P.prepare(c["raw"]).split() == oracle.full_process(c["raw"].lower()).split()
    # ASCII fast path == general path
    for s in ["Coke Zero", "A_b-C.d", "x  y", "", "\tTAB\n", "UPPER lower 123", "~!@#$%^&*()"]:
        assert P.prepare(s) == P.full_process(s.lower())
    # the two restatements agree on a sweep of the BMP (token view)
    chunk = "".join(chr(c) for c in range(32, 0x3100) if not (0xD800 <= c < 0xE000))
    for i in range(0, len(chunk), 97):
        s = "a" + chunk[i:i + 97] + "z"
        assert P.prepare(s).split() == oracle.full_process(s.lower()).split(), hex(32 + i)


def test_prepare_and_pack_bulk_paths():
    items = ["Coke Zero 330ml", "A-B", "", "multi\nline", "Sprite"]
    cps, offs = P.prepare_and_pack(items)
    for i, it in enumerate(items):
        s = "".join(map(chr, cps[int(offs[i]):int(offs[i + 1])]))
        assert s.split() == P.prepare(it).split()
    items2 = items + ["Café مياه"]
    cps, offs = P.prepare_and_pack(items2)
    for i, it in enumerate(items2):
        s = "".join(map(chr, cps[int(offs[i]):int(offs[i + 1])]))
        assert s == P.prepare(it)
    cps, offs = P.prepare_and_pack([])
    assert offs.tolist() == [0]


def test_min_score(oracle):
    for thr in (0.0, 0.35, 0.57, 0.58, 0.8, 1.0, 1.5):
        assert P.min_score_for_threshold(thr) == oracle.min_score_for_threshold(thr)


def _key(score, idx):
    return (score << 40) | (((1 << 40) - 1) - idx)


def test_merge_topk_is_deterministic_and_shard_independent():
    rng = np.random.default_rng(3)
    nq, k, n = 7, 5, 400
    scores = rng.integers(30, 40, size=(nq, n))                 # heavy ties
    full = []
    for q in range(nq):
        order = sorted(range(n), key=lambda i: (-scores[q, i], i))[:k]
        full.append([(i, int(scores[q, i])) for i in order])
    for shards in (1, 2, 3, 8):
        bounds = np.linspace(0, n, shards + 1).astype(int)
        lists = np.zeros((shards, nq, k), dtype=np.uint64)
        for s in range(shards):
            for q in range(nq):
                loc = sorted(range(bounds[s], bounds[s + 1]), key=lambda i: (-scores[q, i], i))[:k]
                for j, i in enumerate(loc):
                    lists[s, q, j] = _key(int(scores[q, i]), i)
        for perm in (np.arange(shards), np.arange(shards)[::-1]):
            idx, sc, cnt = merge_topk(lists[perm], shards, nq, k)
            for q in range(nq):
                assert [(int(idx[q, j]), int(sc[q, j])) for j in range(int(cnt[q]))] == full[q]
    # short lists / padding
    lists = np.zeros((2, 1, 4), dtype=np.uint64)
    lists[0, 0, 0] = _key(50, 3)
    lists[1, 0, 0] = _key(50, 1)
    idx, sc, cnt = merge_topk(lists, 2, 1, 4)
    assert cnt.tolist() == [2] and idx[0, :2].tolist() == [1, 3]
