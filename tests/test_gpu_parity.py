"""GPU parity: libfuzzyhip.so (through the C ABI) against the CPU oracle, bit-exact.

Every test here needs an MI355X; the oracle is the checker only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tts_amd import Corpus, FuzzySearch, _lib, synth  # noqa: E402
from tts_amd.preprocess import min_score_for_threshold, prepare  # noqa: E402


def _oracle_scores(oracle, queries, cand):
    buf, offs = oracle.pack(cand)
    return np.stack([oracle.score_all(oracle.TOKEN_SET, q, buf, offs) for q in queries]).astype(np.uint8)


def _check_topk(oracle, corpus, queries, cand, min_score, k):
    idx, score, cnt = corpus.topk(queries, min_score, k)
    buf, offs = oracle.pack(cand)
    for qi, q in enumerate(queries):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf, offs, min_score, k)
        got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
        assert got == exp, (qi, q, got[:5], exp[:5])


def test_score_matrix_synthetic_short(oracle):
    cps, offs = synth.make_corpus(3000, max_chars=64, seed=1)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 40, seed=2) + ["", "zz", cand[5], cand[7] + " extra tok"]
    with Corpus(cps, offs) as c:
        got = c.score_all(queries)
    exp = _oracle_scores(oracle, queries, cand)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]


def test_packed_short_candidates(oracle):
    """candidates of at most 15 symbols: the packed stream kernel (two tiles per wave), mixed with
    longer ones that stay in the plain stream kernel; token_set, ratio and token_sort"""
    cps_a, offs_a = synth.make_corpus(2500, max_chars=14, seed=31, vocab_size=200, max_tokens=4)
    cps_b, offs_b = synth.make_corpus(900, max_chars=64, seed=32, vocab_size=200)
    cand = synth.to_strings(cps_a, offs_a) + synth.to_strings(cps_b, offs_b) + ["", "a", "abcdefghijklmn", "a b c d"]
    rng = np.random.default_rng(33)
    cand = [cand[int(i)] for i in rng.permutation(len(cand))]
    queries = synth.make_queries(cps_a, offs_a, 30, seed=34) + synth.make_queries(cps_b, offs_b, 20, seed=35)
    queries += ["", "zz", cand[5], cand[7] + " extra tok", "a b c d e f g h i"]
    buf, offs = oracle.pack(cand)
    with Corpus.from_prepared(cand) as c:
        for sk, orc in ((_lib.TOKEN_SET, oracle.TOKEN_SET), (_lib.RATIO, oracle.RATIO), (_lib.TOKEN_SORT, oracle.TOKEN_SORT)):
            got = c.score_all(queries, sk)
            exp = np.stack([oracle.score_all(orc, q, buf, offs) for q in queries]).astype(np.uint8)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (sk, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])
        _check_topk(oracle, c, queries, cand, 35, 10)


def test_topk_order_under_ties(oracle):
    # tiny vocabulary => massive score ties; order must be (score desc, original index asc)
    cps, offs = synth.make_corpus(5000, max_chars=32, seed=3, vocab_size=12)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 12, seed=4, max_chars=32) + [cand[0].split()[0]]
    with Corpus(cps, offs) as c:
        for k, ms in ((10, 35), (1, 0), (64, 90), (5, 100), (1024, 50)):
            _check_topk(oracle, c, queries, cand, ms, k)


def test_ties_clustered_far_into_the_list(oracle):
    """The holders of a token that only one region of the list uses (a 10^8-name list read region by region): tens of
    thousands of ties at 100 whose first original index is beyond any walk from 0.  One block per row (20 queries),
    several per row (2 queries), and the sample rows of a fused call -- the radix select of fz_select_topk."""
    cps, offs = synth.make_corpus(300_000, max_chars=32, seed=131, vocab_size=3000)
    names = synth.to_strings(cps, offs)
    for i in range(240_000, 290_000):
        if i % 2:
            names[i] = "qq7 " + names[i]
    queries = ["qq7", "qq7 zz", names[240_001]] + synth.make_queries(cps, offs, 17, seed=132, max_chars=32)
    buf, o = oracle.pack(names)
    oracle.lib().orc_set_bitpar(1)
    try:
        with Corpus.from_prepared(names) as c:
            for qs, k, ms in ((queries, 10, 35), (queries[:2], 64, 100), (queries * 12, 10, 35)):
                idx, score, cnt = c.topk(qs, ms, k)
                for qi in [q_ for q_ in (0, 1, 2, 5) if q_ < len(qs)]:
                    exp = oracle.topk(oracle.TOKEN_SET, qs[qi], buf, o, ms, k)
                    got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
                    assert got == exp, (len(qs), k, ms, qs[qi], got[:4], exp[:4])
            assert c.timing()["fused"] == 1            # 240 rows x 300 032 = 72 MB: the last call was a fused one
    finally:
        oracle.lib().orc_set_bitpar(0)


def test_edge_cases(oracle):
    cand = ["", "a", "a a a", "b a", "a b", "   ", "abc def", "def abc", "abc", "x" * 70, "ab " * 30,
            "coke 330ml 1x24 np sleek can", "coke sleek can", "fuzzy fuzzy was a bear"]
    queries = ["", "a", "a b", "abc", "coke sleek can", "fuzzy was a bear", "x" * 70, "q" * 5, "ab"]
    with Corpus.from_prepared(cand) as c:
        got = c.score_all(queries)
        exp = _oracle_scores(oracle, queries, cand)
        assert (got == exp).all(), (got.tolist(), exp.tolist())
        for ms in (0, 1, 50, 100, 101):
            _check_topk(oracle, c, queries, cand, ms, 5)
        idx, score, cnt = c.topk(queries, 0, 0)
        assert cnt.sum() == 0
    with Corpus.from_prepared([]) as c:
        idx, score, cnt = c.topk(["abc"], 0, 5)
        assert cnt.tolist() == [0]


def test_multiword_patterns(oracle):
    rng = np.random.default_rng(7)
    cps, offs = synth.make_corpus(700, max_chars=512, seed=5, min_tokens=8, max_tokens=60, vocab_size=300)
    cand = synth.to_strings(cps, offs)
    queries = [cand[int(i)] for i in rng.integers(0, len(cand), 6)]
    queries += synth.make_queries(cps, offs, 6, seed=6, max_chars=512)
    queries += [" ".join(cand[3].split()[:k]) for k in (5, 9, 17)]
    with Corpus(cps, offs) as c:
        got = c.score_all(queries)
        exp = _oracle_scores(oracle, queries, cand)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, [(queries[q][:40], cand[i][:40], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]
        _check_topk(oracle, c, queries[:4], cand, 35, 10)


def test_fuzzysearch_dropin(oracle):
    products = ["Coke Zero 330ml", "COKE 330ML 1x24 NP Sleek Can", "Sprite Zero", "Fanta Orange 1.5L",
                "Café Crème", "Coke-Zero (330ml)", "مياه معدنية 500ml", "coke_zero", "İstanbul Cola", ""]
    customers = ["FATHIMA GROCERY", "Al Madina Supermarket", "fathima grocery llc", "Carrefour"]
    fs = FuzzySearch(products, customers, threshold=0.35)
    for q in ("coke sleek can", "coke zero", "sprite", "مياه", "istanbul", "zzz", ""):
        got = fs.find_best_matches(q, "product")
        exp = oracle.c_find_best_matches(products, q, 0.35, 5)
        assert got == exp, (q, got, exp)
    got = fs.find_best_matches("fatima grocery", "anything-else-means-customers")
    assert got == oracle.c_find_best_matches(customers, "fatima grocery", 0.35, 5)
    many = fs.find_best_matches_many(["coke zero", "fanta"], "product")
    assert many[0] == fs.find_best_matches("coke zero", "product")
    fs.close()


def test_golden_pairs_on_gpu():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "pairs.json")) as f:
        recs = json.load(f)
    cands = [r["c"] for r in recs]
    with Corpus.from_prepared(cands) as c:
        for lo in range(0, len(recs), 64):
            chunk = recs[lo:lo + 64]
            got = c.score_all([r["q"] for r in chunk])
            for j, r in enumerate(chunk):
                assert int(got[j, lo + j]) == r["token_set"], r


def test_full_size_properties():
    """C2-sized corpus: properties that need no oracle pass over 10^6 candidates."""
    n = 1_000_000
    cps, offs = synth.make_corpus(n, max_chars=64)
    queries = synth.make_queries(cps, offs, 48)
    picks = [0, 17, n // 2, n - 1]
    exact = synth.to_strings(cps, offs, 0, 18) + synth.to_strings(cps, offs, n // 2, n // 2 + 1) + synth.to_strings(cps, offs, n - 1, n)
    exact = [exact[0], exact[17], exact[18], exact[19]]
    k = 10
    with Corpus(cps, offs) as c:
        idx, score, cnt = c.topk(queries + exact, 35, k)
        # (1) a candidate queried by its own text scores 100; rows sorted by (score desc, index asc)
        for j in range(4):
            row = len(queries) + j
            assert score[row, 0] == 100
        for r in range(idx.shape[0]):
            keys = [(-int(score[r, j]), int(idx[r, j])) for j in range(int(cnt[r]))]
            assert keys == sorted(keys) and len(set(keys)) == len(keys)
        # (2) token order and repetition in the query do not matter (token SET)
        shuffled = [" ".join(reversed(q.split())) + " " + q.split()[0] if q else q for q in queries]
        idx2, score2, cnt2 = c.topk(shuffled, 35, k)
        assert (idx2 == idx[:len(queries)]).all() and (score2 == score[:len(queries)]).all()
        # (3) sharding invariance: top-k of the whole == merge of the top-k of two halves
        full_keys = None
    half = n // 2
    from tts_amd import merge_topk
    lists = np.zeros((2, len(queries), k), dtype=np.uint64)
    for s, (lo, hi) in enumerate(((0, half), (half, n))):
        sub_offs = offs[lo:hi + 1] - offs[lo]
        sub_cps = cps[int(offs[lo]):int(offs[hi])]
        with Corpus(sub_cps, sub_offs, base_index=lo) as c:
            i3, s3, c3 = c.topk(queries, 35, k)
            for q in range(len(queries)):
                for j in range(int(c3[q])):
                    lists[s, q, j] = (int(s3[q, j]) << 40) | (((1 << 40) - 1) - int(i3[q, j]))
    mi, ms, mc = merge_topk(lists, 2, len(queries), k)
    assert (mi == idx[:len(queries)]).all() and (ms == score[:len(queries)]).all()


def test_sharded_two_ranks_one_gpu():
    """2 ranks sharing the one GPU, gloo for the exchange (ShardedFuzzySearch exchange="torch", keys handed over
    as a torch tensor's data_ptr): sharded result == single-shard result.  Ranks are processes of their own
    (tests/dist_case.py), torch imported first there."""
    from test_dist_gloo import run_ranks
    run_ranks("sharded_gpu", 2, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})


def test_packed_corpus_file_roundtrip(oracle, tmp_path):
    """Corpus.save / Corpus.load (fz_corpus_save / fz_corpus_load): identical top-k, base index kept or rebased."""
    import time
    cps, offs = synth.make_corpus(200_000, max_chars=64, seed=91)
    queries = synth.make_queries(cps, offs, 16, seed=92)
    path = str(tmp_path / "shard0.fzc")
    t0 = time.perf_counter()
    with Corpus(cps, offs, base_index=1000) as c:
        t_ingest = time.perf_counter() - t0
        exp = c.topk(queries, 35, 10)
        exp_ratio = c.score_all(queries[:2], _lib.RATIO)      # builds the raw view, which is then saved too
        c.save(path)
    t0 = time.perf_counter()
    with Corpus.load(path) as c:
        t_load = time.perf_counter() - t0
        assert c.n == 200_000 and c.base_index == 1000
        got = c.topk(queries, 35, 10)
        for a, b in zip(got, exp):
            assert (a == b).all()
        assert (c.score_all(queries[:2], _lib.RATIO) == exp_ratio).all()
    with Corpus.load(path, base_index=0) as c:
        idx, score, cnt = c.topk(queries, 35, 10)
        assert (idx + np.uint64(1000) == exp[0])[exp[1] > 0].all() and (score == exp[1]).all()
    print(f"ingest {t_ingest:.3f} s, load {t_load:.3f} s for 200k candidates")
    with open(path, "r+b") as f:
        f.truncate(100_000)
    with pytest.raises(_lib.FuzzyHipError):
        Corpus.load(path)


def test_workspace_chunking_and_caller_stream(oracle):
    """A tiny workspace limit forces many query chunks; a caller-provided stream carries the work:
    results must not change."""
    # a stream of the caller's own, made with the very HIP runtime the library is bound to (fz_runtime_paths).
    # A torch stream is the other legitimate source and is covered by tests/test_gpu_runtime.py, in processes
    # that import torch first: one HIP runtime per process (tts_amd/_lib.py _RULE).
    import ctypes
    hip = ctypes.CDLL(_lib.runtime_paths()["hip"])
    st = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(st)) == 0
    cps, offs = synth.make_corpus(20_000, max_chars=64, seed=101)
    queries = synth.make_queries(cps, offs, 37, seed=102)
    with Corpus(cps, offs) as c:
        ref = c.topk(queries, 35, 10)
        ref_all = c.score_all(queries[:5])
        c.set_workspace_limit(3 * 20_032)          # three score rows at a time
        got = c.topk(queries, 35, 10)
        for a, b in zip(got, ref):
            assert (a == b).all()
        assert (c.score_all(queries[:5]) == ref_all).all()
        c.set_stream(st.value)
        got = c.topk(queries, 35, 10)
        for a, b in zip(got, ref):
            assert (a == b).all()
        c.set_stream(None)
        assert hip.hipStreamDestroy(st) == 0
        c.set_workspace_limit(4 << 30)
        got = c.topk(queries, 35, 10, _lib.PARTIAL)
        cand = synth.to_strings(cps, offs)
        buf, o = oracle.pack(cand)
        e = oracle.topk(oracle.PARTIAL, queries[0], buf, o, 35, 10)
        assert [(int(got[0][0, j]), int(got[1][0, j])) for j in range(int(got[2][0]))] == e


def test_concurrent_callers_one_handle(oracle):
    """The agent object is shared by every Streamlit session thread (reference
    utils/helper_components.py:59-64): concurrent find_best_matches on one FuzzySearch."""
    import threading
    cps, offs = synth.make_corpus(30_000, max_chars=48, seed=111, vocab_size=300)
    items = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 24, seed=112, max_chars=48)
    fs = FuzzySearch(items, items[:100], threshold=0.35)
    expected = [fs.find_best_matches(q, "product") for q in queries]
    errors = []

    def worker(tid):
        try:
            for rep in range(3):
                for qi in range(tid, len(queries), 4):
                    if fs.find_best_matches(queries[qi], "product") != expected[qi]:
                        errors.append((tid, qi))
        except Exception as exc:      # noqa: BLE001
            errors.append((tid, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    fs.close()
    assert not errors, errors[:3]
    assert expected[0] == oracle.c_find_best_matches(items, queries[0], 0.35, 5)
