"""GPU parity of the FUSED threshold + top-k path (sample tiles -> per-row limit -> keys appended by the EMIT
kernels -> fz_merge_lists): the code that stands in for the threshold, stable sort and slice of reference
src/fuzzy_search.py:31-38 on every batch whose score matrix would be >= 64 MiB -- the path bench.py times.

Three kinds of cases: the bench's own C2 batch (1000 queries x 10^6 candidates) against the oracle and against
the score-matrix path; small corpora with the switch forced (FUZZYHIP_FUSED_MIN_MB=0 before the corpus is
made: the knob is read at creation) for every scorer and kernel family; and a corpus built so that the
appended lists overflow and the chunk is redone through the matrix."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tts_amd import Corpus, _lib, synth  # noqa: E402

THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


@pytest.fixture()
def fast_oracle(oracle):
    oracle.lib().orc_set_bitpar(1)
    yield oracle
    oracle.lib().orc_set_bitpar(0)


def _rows(idx, score, cnt):
    return [[(int(idx[q, j]), int(score[q, j])) for j in range(int(cnt[q]))] for q in range(idx.shape[0])]


def _oracle_topk_rows(oracle, scorer, queries, buf, offs, min_score, k):
    with ThreadPoolExecutor(THREADS) as ex:
        return list(ex.map(lambda q: oracle.topk(scorer, q, buf, offs, min_score, k), queries))


def _force_fused(monkeypatch, **extra):
    monkeypatch.setenv("FUZZYHIP_FUSED_MIN_MB", "0")
    for name, val in extra.items():
        monkeypatch.setenv(name, str(val))


def test_c2_bench_batch_fused_vs_oracle_and_matrix(fast_oracle, monkeypatch):
    """bench.py's C2 step itself: same corpus and query seeds, 1000 queries, k = 10, min_score 35.  The call
    must take the fused path; 64 rows (incl. a one-token query with ~10^5 ties at 100, an exact name and a
    query no candidate reaches) are compared with the oracle's loop over all 10^6 candidates, all 1000 rows with
    the score-matrix path (which test_gpu_fullsize.py checks against the oracle matrix)."""
    n, nq, k, min_score = 1_000_000, 1000, 10, 35
    cps, offs = synth.make_corpus(n, max_chars=64)
    names = synth.to_strings(cps, offs, 0, 64)
    queries = synth.make_queries(cps, offs, nq, seed=synth.QUERY_SEED, max_chars=64)
    queries[61] = names[3].split()[0]
    queries[62] = names[17]
    queries[63] = "zzqq0 qqzz1 zqzq2"
    with Corpus(cps, offs) as c:
        got = c.topk(queries, min_score, k)
        t = c.timing()
        assert t["fused"] == 1 and t["fused_fallback_chunks"] == 0, t
        got64 = c.topk(queries, 100, 64)                         # k at the fused limit, threshold at the top
        assert c.timing()["fused"] == 1
    exp = _oracle_topk_rows(fast_oracle, fast_oracle.TOKEN_SET, queries[:64], cps, offs, min_score, k)
    g = _rows(*got)
    for qi in range(64):
        assert g[qi] == exp[qi], (queries[qi], g[qi][:4], exp[qi][:4])
    assert g[62][0][1] == 100 and len(g[61]) == k
    exp64 = _oracle_topk_rows(fast_oracle, fast_oracle.TOKEN_SET, queries[56:64], cps, offs, 100, 64)
    g64 = _rows(*got64)
    for j, qi in enumerate(range(56, 64)):
        assert g64[qi] == exp64[j], (queries[qi], g64[qi][:4], exp64[j][:4])
    monkeypatch.setenv("FUZZYHIP_NO_FUSED", "1")
    with Corpus(cps, offs) as c:
        ref = c.topk(queries, min_score, k)
        assert c.timing()["fused"] == 0
        ref64 = c.topk(queries, 100, 64)
    for a, b in zip(got, ref):
        assert (a == b).all()
    for a, b in zip(got64, ref64):
        assert (a == b).all()


def test_forced_fused_all_scorers_vs_oracle(oracle, monkeypatch):
    """Fused forced onto a 20 000-candidate corpus (313 tiles, default 1-in-16 sample): token_set and ratio
    over the packed / one-word / two-word stream kernels, token_sort, partial_ratio over the general kernel,
    k = 1, 10, 64, thresholds 0, 35, 100; then the same batch through fz_query_topk_noalloc."""
    _force_fused(monkeypatch)
    cps_a, offs_a = synth.make_corpus(12_000, max_chars=14, seed=501, vocab_size=400, max_tokens=4)
    cps_b, offs_b = synth.make_corpus(8_000, max_chars=64, seed=502, vocab_size=400)
    cand = synth.to_strings(cps_a, offs_a) + synth.to_strings(cps_b, offs_b) + ["", "x" * 70, "ab " * 30]
    rng = np.random.default_rng(503)
    cand = [cand[int(i)] for i in rng.permutation(len(cand))]
    queries = synth.make_queries(cps_a, offs_a, 14, seed=504) + synth.make_queries(cps_b, offs_b, 14, seed=505)
    queries += [cand[0].split()[0] if cand[0] else "a", cand[11], "", "y" * 70, "zz qq"]
    buf, o = oracle.pack(cand)
    cases = ((_lib.TOKEN_SET, oracle.TOKEN_SET, 35, 10), (_lib.TOKEN_SET, oracle.TOKEN_SET, 0, 64),
             (_lib.TOKEN_SET, oracle.TOKEN_SET, 100, 1), (_lib.RATIO, oracle.RATIO, 50, 10),
             (_lib.TOKEN_SORT, oracle.TOKEN_SORT, 40, 7), (_lib.PARTIAL, oracle.PARTIAL, 60, 10))
    with Corpus.from_prepared(cand) as c:
        assert c.info()["n_tiles"] >= 256
        for sc, orc, ms, k in cases:
            qs = queries[:8] + queries[-5:] if sc == _lib.PARTIAL else queries
            got = c.topk(qs, ms, k, sc)
            t = c.timing()
            assert t["fused"] == 1, (sc, t)
            exp = _oracle_topk_rows(oracle, orc, qs, buf, o, ms, k)
            g = _rows(*got)
            for qi, q in enumerate(qs):
                assert g[qi] == exp[qi], (sc, ms, k, q, g[qi][:4], exp[qi][:4])
            if sc != _lib.PARTIAL:
                again = c.topk(qs, ms, k, sc, noalloc=True)
                assert c.timing()["fused"] == 1
                for a, b in zip(again, got):
                    assert (a == b).all(), (sc, ms, k, "noalloc")


def test_forced_fused_wide_alphabet(oracle, monkeypatch):
    """> 254 distinct code points: 16-bit symbols, general kernel with the EMIT epilogue"""
    from test_emul_sanitizer import _wide_corpus
    _force_fused(monkeypatch)
    cand, vocab = _wide_corpus(17_000, 511)
    queries = [cand[3], cand[10] + " " + vocab[5], vocab[7], "", "zz", " ".join(vocab[10:18])]
    buf, o = oracle.pack(cand)
    with Corpus.from_prepared(cand) as c:
        assert c.info()["alphabet_size"] > 256 and c.info()["n_tiles"] >= 256
        for sc, orc, ms, k in ((_lib.TOKEN_SET, oracle.TOKEN_SET, 35, 10), (_lib.RATIO, oracle.RATIO, 30, 64)):
            got = c.topk(queries, ms, k, sc)
            assert c.timing()["fused"] == 1
            exp = _oracle_topk_rows(oracle, orc, queries, buf, o, ms, k)
            g = _rows(*got)
            for qi, q in enumerate(queries):
                assert g[qi] == exp[qi], (sc, q, g[qi][:4], exp[qi][:4])


def test_fused_list_overflow_falls_back(oracle, monkeypatch):
    """Every candidate has the same canonical length and token count, so the physical order is the list order
    and the sample tiles are tiles 8, 24, 40, ...: 3 000 candidates that hold the query's token sit in other
    tiles only.  The sample then finds nothing, the row's limit stays at the threshold, 3 000 keys arrive for a
    list of 2 048: the chunk must be redone through the score matrix, with the oracle's result."""
    _force_fused(monkeypatch)
    rng = np.random.default_rng(521)
    n = 64 * 320
    fill = ["".join(chr(97 + int(x)) for x in rng.integers(0, 20, 6)) for _ in range(n)]
    cand = ["yy " + f for f in fill]
    slots = [i for i in range(n) if (i // 64) % 16 != 8 and (i // 64) != n // 64 - 1]
    for i in rng.choice(len(slots), 3000, replace=False):
        cand[slots[int(i)]] = "zz " + fill[slots[int(i)]]
    queries = ["zz", "zz " + fill[slots[5]], "yy", "qq"]
    buf, o = oracle.pack(cand)
    with Corpus.from_prepared(cand) as c:
        got = c.topk(queries, 100, 64)
        t = c.timing()
        assert t["fused"] == 1 and t["fused_fallback_chunks"] >= 1 and t["fused_fallback_rows"] == 1, t    # only the 'zz' row
        exp = _oracle_topk_rows(oracle, oracle.TOKEN_SET, queries, buf, o, 100, 64)
        g = _rows(*got)
        for qi, q in enumerate(queries):
            assert g[qi] == exp[qi], (q, g[qi][:4], exp[qi][:4])
        # more overflowed rows than are redone one by one: the chunk goes through the matrix as a whole
        many = c.topk(["zz"] * 300, 100, 64)
        t = c.timing()
        assert t["fused"] == 1 and t["fused_fallback_rows"] == 300, t
        assert all(r == exp[0] for r in _rows(*many))
        # the same rows when nothing overflows (k below what the sample leaves): no fallback
        got = c.topk(queries[2:], 35, 10)
        assert c.timing()["fused_fallback_chunks"] == 0
        exp = _oracle_topk_rows(oracle, oracle.TOKEN_SET, queries[2:], buf, o, 35, 10)
        assert _rows(*got) == exp


def test_submit_wait_two_in_flight(oracle, monkeypatch):
    """fz_query_topk_submit / fz_query_topk_wait: two batches in flight on one corpus, fetched in either order, give
    what fz_query_topk gives -- fused path (a 200 000-candidate shard, 600 queries) and matrix path (3 queries);
    a third submit is refused; a fused call whose list overflows is repaired by its wait while another call is in
    flight behind it (the corpus of test_fused_list_overflow_falls_back)."""
    n, k, ms = 200_000, 10, 35
    cps, offs = synth.make_corpus(n, max_chars=64, seed=531)
    qa = synth.make_queries(cps, offs, 600, seed=532, max_chars=64)
    qb = synth.make_queries(cps, offs, 500, seed=533, max_chars=64)
    with Corpus(cps, offs) as c:
        ref_a, ref_b, ref_s = c.topk(qa, ms, k), c.topk(qb, ms, k), c.topk(qa[:3], ms, k)
        assert c.topk(qa, ms, k) and c.timing()["fused"] == 1
        ta = c.topk_submit(qa, ms, k)
        tb = c.topk_submit(qb, ms, k)
        with pytest.raises(_lib.FuzzyHipError):
            c.topk_submit(qa[:3], ms, k)
        got_b = c.topk_wait(tb)                                    # the later call first
        ts = c.topk_submit(qa[:3], ms, k)                          # its slot is free again; matrix path (3 rows)
        got_a = c.topk_wait(ta)
        got_s = c.topk_wait(ts)
        with pytest.raises(RuntimeError):
            c.topk_wait(ts)
        for got, ref in ((got_a, ref_a), (got_b, ref_b), (got_s, ref_s)):
            for a, b in zip(got, ref):
                assert (a == b).all()
        # a queue of batches, two in flight throughout
        prev, outs = None, []
        for i in range(6):
            t = c.topk_submit(qa if i % 2 == 0 else qb, ms, k)
            if prev is not None:
                outs.append(c.topk_wait(prev))
            prev = t
        outs.append(c.topk_wait(prev))
        for i, got in enumerate(outs):
            for a, b in zip(got, ref_a if i % 2 == 0 else ref_b):
                assert (a == b).all(), i
    _force_fused(monkeypatch)
    rng = np.random.default_rng(521)
    n = 64 * 320
    fill = ["".join(chr(97 + int(x)) for x in rng.integers(0, 20, 6)) for _ in range(n)]
    cand = ["yy " + f for f in fill]
    slots = [i for i in range(n) if (i // 64) % 16 != 8 and (i // 64) != n // 64 - 1]
    for i in rng.choice(len(slots), 3000, replace=False):
        cand[slots[int(i)]] = "zz " + fill[slots[int(i)]]
    queries = ["zz", "zz " + fill[slots[5]], "yy", "qq"]
    buf, o = oracle.pack(cand)
    exp = _oracle_topk_rows(oracle, oracle.TOKEN_SET, queries, buf, o, 100, 64)
    with Corpus.from_prepared(cand) as c:
        t1 = c.topk_submit(queries, 100, 64)                       # 'zz' overflows its list
        t2 = c.topk_submit(queries[2:], 100, 64)                   # queued behind it; reuses the workspace
        g1 = _rows(*c.topk_wait(t1))                               # redoes row 0 behind t2's kernels
        g2 = _rows(*c.topk_wait(t2))
        assert g1 == exp and g2 == exp[2:]


def test_submit_wait_from_two_threads():
    """two threads, one corpus, each with one ticket at a time (so never more than FZ_MAX_IN_FLIGHT): a submit that
    finds the other thread's call running only builds its tables and is sent out by whichever wait sees the GPU free --
    possibly the other thread's.  Every result equals the synchronous call's."""
    cps, offs = synth.make_corpus(150_000, max_chars=64, seed=541)
    batches = [synth.make_queries(cps, offs, 300 + 50 * i, seed=542 + i, max_chars=64) for i in range(2)]
    with Corpus(cps, offs) as c:
        refs = [c.topk(b, 35, 10) for b in batches]
        errors = []

        def worker(t):
            try:
                for _ in range(15):
                    got = c.topk_wait(c.topk_submit(batches[t], 35, 10))
                    for a, b in zip(got, refs[t]):
                        assert (a == b).all()
            except Exception as e:      # noqa: BLE001 -- reported by the main thread
                errors.append((t, repr(e)))

        import threading
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        assert not any(t.is_alive() for t in ts) and not errors, errors


def test_fused_sharded_exchange(monkeypatch):
    """fz_query_topk_sharded with fused local top-k calls: 2 and 3 ranks as threads over the RCCL stand-in on
    shards of >= 256 tiles (tests/comm_case.py 'threads_fused'), results equal to the single-corpus top-k"""
    from test_gpu_comm import HERE, _run
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake)
    _run("threads_fused", {"FUZZYHIP_RCCL_LIB": fake, "FUZZYHIP_FUSED_MIN_MB": "0"})
    # a shard whose fused lists overflow: its status word sends every rank round once more (no host synchronisation
    # stands between the local step and the all-gather, so the overflow is only known afterwards)
    _run("threads_overflow", {"FUZZYHIP_RCCL_LIB": fake, "FUZZYHIP_FUSED_MIN_MB": "0"})
