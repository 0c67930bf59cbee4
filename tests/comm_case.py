"""Helper of tests/test_gpu_comm.py (run as a script in its own process, because the library opens ONE
librccl per process): the sharded entry point fz_query_topk_sharded against the single-corpus top-k.

    python tests/comm_case.py threads   ranks = threads of this process sharing GPU 0, FUZZYHIP_RCCL_LIB must
                                        name tests/fake_rccl/libfake_rccl.so (2 and 3 ranks)
    python tests/comm_case.py real      one rank, the real librccl (communicator of size 1)
    python tests/comm_case.py threads_fused   as `threads`, on shards of >= 256 tiles with FUZZYHIP_FUSED_MIN_MB=0
                                        in the environment: every local top-k takes the fused path (asserted);
                                        expected results come from a score-matrix corpus (FUZZYHIP_NO_FUSED=1)
    python tests/comm_case.py threads_fail    two ranks; the local step of rank 1 fails (it is handed a query beyond
                                        FZ_MAX_PATTERN): rank 1 raises its own error, rank 0 is told that a peer failed,
                                        nobody waits in the collective, and the communicator serves the next call"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tts_amd import Corpus, _lib, synth  # noqa: E402
from tts_amd.sharded import LibraryExchange, shard_bounds  # noqa: E402


def corpus_and_queries(n=6000):
    cps, offs = synth.make_corpus(n, max_chars=64, seed=91, vocab_size=300)
    names = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 12, seed=92) + [names[5].split()[0], names[77], "zz qq", ""]
    return names, queries


def expected(names, queries, cases):
    out = {}
    with Corpus.from_prepared(names) as c:
        for sc, ms, k in cases:
            out[(sc, ms, k)] = c.topk(queries, ms, k, sc)
    return out


def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def run_threads(n_ranks, names, queries, cases, exp, want_fused=False, want_overflow=()):
    box, ready = {}, threading.Event()

    def share(my_id):
        if my_id is not None:
            box["id"] = my_id
            ready.set()
            return my_id
        assert ready.wait(60)
        return box["id"]

    errors, results = [], {}

    def rank_main(rank):
        try:
            lo, hi = shard_bounds(len(names), n_ranks, rank)
            with Corpus.from_prepared(names[lo:hi], device=0, base_index=lo) as shard:
                comm = LibraryExchange(rank, n_ranks, 0, share)
                for case in cases:
                    sc, ms, k = case
                    results[(rank, case)] = shard.topk_sharded(comm.handle, queries, ms, k, sc)
                    if want_fused:
                        assert shard.timing()["fused"] == 1, (rank, case, shard.timing())
                    if case in want_overflow:
                        assert shard.timing()["fused_fallback_rows"] >= 1, (rank, case, shard.timing())
                comm.close()
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(n_ranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    assert not any(t.is_alive() for t in ts), "a rank did not finish"
    for r in range(n_ranks):
        for case in cases:
            assert same(results[(r, case)], exp[case]), (n_ranks, r, case)


def main():
    mode = sys.argv[1]
    cases = [(_lib.TOKEN_SET, 35, 10), (_lib.TOKEN_SET, 0, 64), (_lib.RATIO, 50, 7), (_lib.TOKEN_SET, 100, 5)]
    if mode == "threads_fused":
        assert "fake_rccl" in os.environ.get("FUZZYHIP_RCCL_LIB", "") and os.environ.get("FUZZYHIP_FUSED_MIN_MB") == "0"
        names, queries = corpus_and_queries(3 * 17_000)
        os.environ["FUZZYHIP_NO_FUSED"] = "1"              # read when a corpus is created
        exp = expected(names, queries, cases)
        del os.environ["FUZZYHIP_NO_FUSED"]
        for n_ranks in (2, 3):
            run_threads(n_ranks, names, queries, cases, exp, want_fused=True)
        print("comm case ok:", mode)
        return
    if mode == "threads_overflow":
        # the corpus of tests/test_gpu_fused.py::test_fused_list_overflow_falls_back, once per shard: in every shard
        # 3 000 holders of the token 'zz' sit outside the sample tiles, so every rank's 'zz' row overflows its list
        assert "fake_rccl" in os.environ.get("FUZZYHIP_RCCL_LIB", "") and os.environ.get("FUZZYHIP_FUSED_MIN_MB") == "0"
        rng = np.random.default_rng(521)
        per = 64 * 320
        names = []
        for shard in range(2):
            fill = ["".join(chr(97 + int(x)) for x in rng.integers(0, 20, 6)) for _ in range(per)]
            part = ["yy " + f for f in fill]
            slots = [i for i in range(per) if (i // 64) % 16 != 8 and (i // 64) != per // 64 - 1]
            for i in rng.choice(len(slots), 3000, replace=False):
                part[slots[int(i)]] = "zz " + fill[slots[int(i)]]
            names += part
        queries = ["zz", "yy", "qq", "zz " + names[3][3:], "yy " + names[per + 7][3:], "", "zz zz", "yy zz"]
        cases = [(_lib.TOKEN_SET, 100, 64), (_lib.TOKEN_SET, 35, 10)]
        os.environ["FUZZYHIP_NO_FUSED"] = "1"
        exp = expected(names, queries, cases)
        del os.environ["FUZZYHIP_NO_FUSED"]
        run_threads(2, names, queries, cases, exp, want_fused=True, want_overflow={cases[0]})
        print("comm case ok:", mode)
        return
    if mode == "threads_fail":
        assert "fake_rccl" in os.environ.get("FUZZYHIP_RCCL_LIB", "")
        names, queries = corpus_and_queries()
        exp = expected(names, queries, cases[:1])
        box, ready, errors, outcome = {}, threading.Event(), [], {}

        def share(my_id):
            if my_id is not None:
                box["id"] = my_id
                ready.set()
                return my_id
            assert ready.wait(60)
            return box["id"]

        def rank_main(rank):
            try:
                lo, hi = shard_bounds(len(names), 2, rank)
                with Corpus.from_prepared(names[lo:hi], device=0, base_index=lo) as shard:
                    comm = LibraryExchange(rank, 2, 0, share)
                    bad = list(queries)
                    if rank == 1:
                        bad[3] = " ".join("t%03d" % i for i in range(300))   # 1 500 code points of distinct tokens: beyond FZ_MAX_PATTERN, on this rank only
                    try:
                        shard.topk_sharded(comm.handle, bad, 35, 10)
                        outcome[rank] = "returned"
                    except _lib.FuzzyHipError as e:
                        outcome[rank] = str(e)
                    outcome[(rank, "next")] = shard.topk_sharded(comm.handle, queries, 35, 10)
                    comm.close()
            except Exception as e:      # noqa: BLE001
                errors.append((rank, repr(e)))

        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        assert not any(t.is_alive() for t in ts), "a rank is still waiting in the collective"
        assert not errors, errors
        assert "another rank failed" in outcome[0], outcome[0]
        assert outcome[1] != "returned" and "another rank" not in outcome[1], outcome[1]
        for r in range(2):
            assert same(outcome[(r, "next")], exp[cases[0]]), r
        print("comm case ok:", mode)
        return
    names, queries = corpus_and_queries()
    exp = expected(names, queries, cases)
    if mode == "threads":
        assert "fake_rccl" in os.environ.get("FUZZYHIP_RCCL_LIB", "")
        for n_ranks in (2, 3):
            run_threads(n_ranks, names, queries, cases, exp)
    else:
        with Corpus.from_prepared(names) as c:
            comm = LibraryExchange(0, 1, 0, lambda my_id: my_id)
            for case in cases:
                sc, ms, k = case
                assert same(c.topk_sharded(comm.handle, queries, ms, k, sc), exp[case]), case
            comm.close()
    print("comm case ok:", mode)


if __name__ == "__main__":
    main()
