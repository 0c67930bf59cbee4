"""Kernel + host sources of libfuzzyhip under AddressSanitizer/UBSan on the CPU.

GPU AddressSanitizer is not available on the MI355X pool, and a kernel that faults there takes the
whole box down.  tests/emul compiles the UNMODIFIED tts_amd/csrc sources with g++ over a small HIP
stand-in (one OS thread per work-item) and this file runs tiny cases of every kernel through the C
ABI under the sanitizers, checking the results against the oracle as well.  This is a checker of
the sources, not a product path: tts_amd never loads the emulated build."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))

from tts_amd import synth  # noqa: E402
from tts_amd.preprocess import pack_prepared  # noqa: E402

MAGIC = 0x465A454D554C0001


@pytest.fixture(scope="module")
def runner():
    import build_emul
    return build_emul.build()


def _run(runner, tmp_path, op, sk, cand, queries, min_score=0, k=0, env=None):
    cps, offs = pack_prepared(list(cand)) if not isinstance(cand, tuple) else cand
    qcps, qoffs = pack_prepared(list(queries))
    n, nq = len(offs) - 1, len(qoffs) - 1
    ncps, nqcps = int(offs[-1]), int(qoffs[-1])
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    with open(case, "wb") as f:
        np.array([MAGIC, op, sk, np.int64(min_score).astype(np.uint64), k, n, ncps, nq, nqcps], dtype=np.uint64).tofile(f)
        np.ascontiguousarray(cps[:ncps], dtype=np.uint32).tofile(f)
        np.ascontiguousarray(offs, dtype=np.uint64).tofile(f)
        np.ascontiguousarray(qcps[:nqcps], dtype=np.uint32).tofile(f)
        np.ascontiguousarray(qoffs, dtype=np.uint64).tofile(f)
    e = dict(os.environ)
    e["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
    e["UBSAN_OPTIONS"] = "print_stacktrace=1"
    e.update(env or {})
    r = subprocess.run([runner, case, out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    _run.last_stderr = r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    raw = np.fromfile(out, dtype=np.uint8)
    if op in (0, 4):
        return raw.reshape(nq, n)
    if op == 1:
        return raw.view(np.uint16).reshape(nq, n)
    idx = raw[:nq * k * 8].view(np.uint64).reshape(nq, k)
    score = raw[nq * k * 8:nq * k * 12].view(np.int32).reshape(nq, k)
    cnt = raw[nq * k * 12:].view(np.uint32)
    return idx, score, cnt


def _oracle_scores(oracle, scorer, queries, cand):
    buf, offs = oracle.pack(cand)
    return np.stack([oracle.score_all(scorer, q, buf, offs) for q in queries]).astype(np.uint8)


@pytest.mark.parametrize("no_stream", ["0", "1"])
def test_token_set_scores_short(oracle, runner, tmp_path, no_stream):
    cps, offs = synth.make_corpus(330, max_chars=64, seed=11, vocab_size=60)
    cand = synth.to_strings(cps, offs) + ["", "a", "x" * 64, "ab " * 21 + "c"]
    queries = synth.make_queries(cps, offs, 9, seed=12) + ["", cand[5], "zz top", "y" * 70]
    got = _run(runner, tmp_path, 0, 0, cand, queries, env={"FUZZYHIP_NO_STREAM": no_stream})
    exp = _oracle_scores(oracle, oracle.TOKEN_SET, queries, cand)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]]


def test_packed_short_tiles(oracle, runner, tmp_path):
    """short candidates go to the packed stream kernel, two tiles per wave (7 tiles: three pairs, the
    odd one stays with the one-tile kernel); every streamed scorer against the oracle"""
    cps, offs = synth.make_corpus(420, max_chars=14, seed=21, vocab_size=60, max_tokens=4)
    cand = synth.to_strings(cps, offs) + ["", "a", "abcdefghijklmn", "ab cd ef gh", "a b c d"]
    queries = synth.make_queries(cps, offs, 9, seed=22) + ["", cand[5], "zz top", "y" * 70, "a b c d e f",
                                                            cand[17] + " " + cand[40]]
    for sk, orc in ((0, oracle.TOKEN_SET), (1, oracle.RATIO), (2, oracle.TOKEN_SORT)):
        got = _run(runner, tmp_path, 0, sk, cand, queries, env={"FUZZYHIP_TRACE": "1"})
        assert "7 tiles" in _run.last_stderr and "-> packed 6, stream1 1," in _run.last_stderr, _run.last_stderr[-500:]
        exp = _oracle_scores(oracle, orc, queries, cand)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (sk, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])


def test_packed_pairs_of_unequal_tiles(oracle, runner, tmp_path):
    """tiles sorted by length and paired short with long (split S = longest candidate of the short tile):
    lengths 0..29 give pairs with many different splits; token_set and ratio against the oracle"""
    import random
    rnd = random.Random(41)
    words = ["".join(rnd.choice("abcdefg") for _ in range(rnd.randint(1, 9))) for _ in range(40)]
    cand = []
    for _ in range(700):
        toks, target = [], rnd.randint(0, 29)
        while len(toks) < 4:
            w = rnd.choice(words)
            if len(" ".join(toks + [w])) > target: break
            toks.append(w)
        cand.append(" ".join(toks))
    queries = [cand[3], cand[500], "abc de", "gfedcba abc", "", " ".join(words[:4]), words[7] + " " + words[8] + " zz"]
    for sk, orc in ((0, oracle.TOKEN_SET), (1, oracle.RATIO)):
        got = _run(runner, tmp_path, 0, sk, cand, queries, env={"FUZZYHIP_TRACE": "1"})
        packed = int(_run.last_stderr.split("-> packed ")[1].split(",")[0])
        assert packed >= 8, _run.last_stderr[-300:]
        exp = _oracle_scores(oracle, orc, queries, cand)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (sk, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])


def test_token_set_multiword(oracle, runner, tmp_path):
    cps, offs = synth.make_corpus(150, max_chars=300, seed=13, min_tokens=6, max_tokens=40, vocab_size=80)
    cand = synth.to_strings(cps, offs)
    queries = [cand[3], cand[77], " ".join(cand[9].split()[:7]), "zz"] + synth.make_queries(cps, offs, 3, seed=14, max_chars=300)
    got = _run(runner, tmp_path, 0, 0, cand, queries)
    exp = _oracle_scores(oracle, oracle.TOKEN_SET, queries, cand)
    assert (got == exp).all()


def test_ratio_and_token_sort(oracle, runner, tmp_path):
    cps, offs = synth.make_corpus(200, max_chars=64, seed=15, vocab_size=40)
    cand = synth.to_strings(cps, offs) + ["", "q" * 90]
    queries = synth.make_queries(cps, offs, 5, seed=16) + ["", cand[2], "w" * 80]
    for sk, orc in ((1, oracle.RATIO), (2, oracle.TOKEN_SORT)):
        got = _run(runner, tmp_path, 0, sk, cand, queries)
        exp = _oracle_scores(oracle, orc, queries, cand)
        assert (got == exp).all(), sk


def test_distances(oracle, runner, tmp_path):
    import random
    rnd = random.Random(17)
    cand = ["", "a", "ab"] + ["".join(rnd.choice("abcd ") for _ in range(rnd.randint(0, 140))) for _ in range(90)]
    queries = ["", "abc"] + ["".join(rnd.choice("abcd ") for _ in range(n)) for n in (31, 32, 33, 64, 65, 130)]
    for kind, fn in ((0, oracle.indel), (1, oracle.levenshtein)):
        got = _run(runner, tmp_path, 1, kind, cand, queries)
        for qi, q in enumerate(queries):
            for i, s in enumerate(cand):
                assert int(got[qi, i]) == fn(q, s), (kind, q, s)


def _difflib_matches(a, b):
    from difflib import SequenceMatcher
    return sum(blk.size for blk in SequenceMatcher(None, a, b, autojunk=False).get_matching_blocks())


def test_sequence_matcher_matches(runner, tmp_path):
    """fz_distance_all kind 4 (fz_ro_matches) against the stdlib's difflib.SequenceMatcher itself: small alphabets
    (many equally long blocks: the earliest-in-a, earliest-in-b rule decides), repeats, one- and two-word patterns,
    empty strings, candidates beyond 64 symbols (0xFFFF: left to difflib)"""
    import random
    rnd = random.Random(131)
    def word(alpha, lo, hi):
        return "".join(rnd.choice(alpha) for _ in range(rnd.randint(lo, hi)))
    cand = [word("ab", 0, 64) for _ in range(90)] + [word("abc", 0, 40) for _ in range(90)] + [word("abcdefghij", 1, 64) for _ in range(120)]
    cand += ["", "a" * 64, "ab" * 32, "abc" * 21, "x" * 65, "ab" * 50, "coke330ml", "cokezero330mlsleekcan", "b" * 63 + "a"]
    queries = ["", "a", "ab" * 16, "abcabcabc", "a" * 32, "ba" * 20, word("ab", 33, 64), word("abc", 20, 32), word("abcdefghij", 40, 64),
               "cokezero330ml", "a" * 64, "ab" * 32]
    got = _run(runner, tmp_path, 1, 4, cand, queries)
    for qi, q in enumerate(queries):
        for i, c in enumerate(cand):
            exp = 0xFFFF if len(c) > 64 else _difflib_matches(q, c)
            assert int(got[qi, i]) == exp, (q, c, int(got[qi, i]), exp)


def test_topk_under_ties(oracle, runner, tmp_path):
    cps, offs = synth.make_corpus(700, max_chars=32, seed=18, vocab_size=9)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 5, seed=19, max_chars=32) + [cand[0].split()[0], ""]
    buf, o = oracle.pack(cand)
    for k, ms in ((10, 35), (1, 0), (5, 100), (300, 50)):
        idx, score, cnt = _run(runner, tmp_path, 2, 0, cand, queries, min_score=ms, k=k)
        for qi, q in enumerate(queries):
            exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, ms, k)
            got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
            assert got == exp, (k, ms, q, got[:5], exp[:5])


def test_sliced_select_few_rows(oracle, runner, tmp_path):
    """a few rows over a 'large' shard (FUZZYHIP_SELECT_SLICE=256 makes 700 candidates large): several blocks per row,
    the last one to finish merges; ties in every run, the walk in original order inside a run, the first-passing order"""
    env = {"FUZZYHIP_SELECT_SLICE": "256"}
    cps, offs = synth.make_corpus(700, max_chars=32, seed=18, vocab_size=9)
    cand = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 5, seed=19, max_chars=32) + [cand[0].split()[0], ""]
    buf, o = oracle.pack(cand)
    for k, ms in ((10, 35), (1, 0), (5, 100), (300, 50)):
        idx, score, cnt = _run(runner, tmp_path, 2, 0, cand, queries, min_score=ms, k=k, env=env)
        for qi, q in enumerate(queries):
            exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, ms, k)
            got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
            assert got == exp, (k, ms, q, got[:5], exp[:5])
    cand2 = ["aa bb"] * 2500 + ["aa"] * 3 + ["cc"] * 20
    buf2, o2 = oracle.pack(cand2)
    idx, score, cnt = _run(runner, tmp_path, 2, 0, cand2, ["aa bb", "cc dd"], min_score=35, k=12, env=env)
    for qi, q in enumerate(["aa bb", "cc dd"]):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf2, o2, 35, 12)
        assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == exp
    idx, score, cnt = _run(runner, tmp_path, 2, 0 | 0x100, cand2, ["aa", "cc"], min_score=100, k=4, env=env)
    for qi, q in enumerate(["aa", "cc"]):
        exp = [i for i, s_ in enumerate(cand2) if oracle.score(oracle.TOKEN_SET, q, s_) >= 100][:4]
        assert [int(idx[qi, j]) for j in range(int(cnt[qi]))] == exp


def test_fused_topk_sample_and_lists(oracle, runner, tmp_path):
    """the fused top-k path (sample tiles -> limit per row -> appended lists -> merge) and the slab-major
    item order, forced onto a small corpus: every 2nd tile is a sample tile, slabs of 1 KB; token_set over
    the stream kernels, partial_ratio over the general kernel, a many-ties row, k up to the fused limit"""
    env = {"FUZZYHIP_SAMPLE_MIN_TILES": "4", "FUZZYHIP_SAMPLE_EVERY": "2", "FUZZYHIP_SLAB_KB": "1", "FUZZYHIP_FUSED_MIN_MB": "0",
           "FUZZYHIP_TRACE": "1"}
    cps_a, offs_a = synth.make_corpus(420, max_chars=14, seed=61, vocab_size=40, max_tokens=4)
    cps_b, offs_b = synth.make_corpus(330, max_chars=64, seed=62, vocab_size=40)
    cand = synth.to_strings(cps_a, offs_a) + synth.to_strings(cps_b, offs_b) + ["x" * 70, "ab " * 30]
    queries = synth.make_queries(cps_a, offs_a, 4, seed=63) + synth.make_queries(cps_b, offs_b, 3, seed=64)
    queries += [cand[0].split()[0], "", "y" * 70]
    buf, o = oracle.pack(cand)
    for sk, orc, k, ms in ((0, oracle.TOKEN_SET, 10, 35), (0, oracle.TOKEN_SET, 64, 0), (3, oracle.PARTIAL, 7, 50)):
        idx, score, cnt = _run(runner, tmp_path, 2, sk, cand, queries, min_score=ms, k=k, env=env)
        assert "fused top-k over %d rows" % len(queries) in _run.last_stderr, _run.last_stderr[-500:]
        for qi, q in enumerate(queries):
            exp = oracle.topk(orc, q, buf, o, ms, k)
            got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
            assert got == exp, (sk, k, ms, q, got[:5], exp[:5])


def test_submit_wait_two_in_flight(oracle, runner, tmp_path):
    """fz_query_topk_submit / fz_query_topk_wait (runner op 5: two calls in flight, a refused third, waits in reverse
    order, a ticket fetched twice refused): the matrix path, the fused path, and a fused call whose list overflows
    (the hits sit outside the sample tiles, so the sample limit stays at the threshold and the row's 2 048 slots do
    not hold them: the wait redoes that row through the score matrix)"""
    cps_a, offs_a = synth.make_corpus(420, max_chars=14, seed=81, vocab_size=40, max_tokens=4)
    cps_b, offs_b = synth.make_corpus(330, max_chars=64, seed=82, vocab_size=40)
    cand = synth.to_strings(cps_a, offs_a) + synth.to_strings(cps_b, offs_b) + ["x" * 70, "ab " * 30]
    queries = synth.make_queries(cps_a, offs_a, 4, seed=83) + synth.make_queries(cps_b, offs_b, 3, seed=84) + [cand[0].split()[0], "", "y" * 70]
    buf, o = oracle.pack(cand)
    fused = {"FUZZYHIP_SAMPLE_MIN_TILES": "4", "FUZZYHIP_SAMPLE_EVERY": "2", "FUZZYHIP_FUSED_MIN_MB": "0", "FUZZYHIP_TRACE": "1"}
    for env, k, ms in (({}, 10, 35), (fused, 10, 35), (fused, 64, 0)):
        idx, score, cnt = _run(runner, tmp_path, 5, 0, cand, queries, min_score=ms, k=k, env=env)
        assert ("fused top-k over" in _run.last_stderr) == bool(env)
        for qi, q in enumerate(queries):
            exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, ms, k)
            got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
            assert got == exp, (k, ms, q, got[:5], exp[:5])
    # overflow: 80 tiles of one canonical length (physical order = list order); the sample is every 2nd tile (the odd
    # ones), the 2 560 candidates that hold the query's token sit in the even ones: the sample finds nothing
    rng = np.random.default_rng(85)
    fill = ["".join(chr(97 + int(x)) for x in rng.integers(0, 20, 6)) for _ in range(64 * 80)]
    cand = [("zz " if (i // 64) % 2 == 0 else "yy ") + f for i, f in enumerate(fill)]
    queries = ["zz", "zz " + fill[7], "qq", "zz"]
    buf, o = oracle.pack(cand)
    idx, score, cnt = _run(runner, tmp_path, 5, 0, cand, queries, min_score=100, k=5, env=fused)
    assert "rows overflowed their lists" in _run.last_stderr, _run.last_stderr[-800:]
    for qi, q in enumerate(queries):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, 100, 5)
        got = [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))]
        assert got == exp, (q, got[:5], exp[:5])


def test_query_tables_built_by_threads(oracle, runner, tmp_path):
    """>= 256 queries in one chunk: the host builds the query tables on several threads (stream rows handed
    out first, per-thread batches appended with their offsets moved); token_set over the stream kernels and
    partial_ratio over the general kernel (Peq tables and token triples of every part) against the oracle"""
    cps, offs = synth.make_corpus(70, max_chars=40, seed=71, vocab_size=30)
    cand = synth.to_strings(cps, offs) + ["", "x" * 70]
    base = synth.make_queries(cps, offs, 37, seed=72, max_chars=40) + ["", "zz top", "y" * 70, cand[3]]
    queries = [base[i % len(base)] for i in range(300)]
    for sk, orc in ((0, oracle.TOKEN_SET), (3, oracle.PARTIAL)):
        got = _run(runner, tmp_path, 0, sk, cand, queries, env={"FUZZYHIP_PREP_THREADS": "4"})
        exp = _oracle_scores(oracle, orc, base, cand)[np.arange(300) % len(base)]
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (sk, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])


def test_topk_many_ties_walk(oracle, runner, tmp_path):
    # > SEL_TIECAP equal scores at the cut: the original-order walk of fz_select_topk
    cand = ["aa bb"] * 2500 + ["aa"] * 3 + ["cc"] * 20
    idx, score, cnt = _run(runner, tmp_path, 2, 0, cand, ["aa bb", "cc dd"], min_score=35, k=12)
    buf, o = oracle.pack(cand)
    for qi, q in enumerate(["aa bb", "cc dd"]):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, 35, 12)
        assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == exp


def test_topk_ties_far_into_the_list(oracle, runner, tmp_path):
    """> SEL_TIECAP ties at the cut, all of them beyond the walk's budget (32 x 256 candidates): radix select over
    the original indices; one block per row and several (FUZZYHIP_SELECT_SLICE)"""
    cand = ["cc dd %d" % (i % 7) for i in range(8300)] + ["aa bb"] * 2500 + ["aa"] * 3
    buf, o = oracle.pack(cand)
    for env in (None, {"FUZZYHIP_SELECT_SLICE": "2048"}):
        for k, ms in ((12, 35), (1, 100)):
            idx, score, cnt = _run(runner, tmp_path, 2, 0, cand, ["aa bb", "cc dd"], min_score=ms, k=k, env=env)
            for qi, q in enumerate(["aa bb", "cc dd"]):
                exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, ms, k)
                assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == exp, (env, k, ms, q)
    idx, score, cnt = _run(runner, tmp_path, 2, 0 | 0x100, cand, ["aa"], min_score=100, k=5)
    exp = [i for i, s_ in enumerate(cand) if oracle.score(oracle.TOKEN_SET, "aa", s_) >= 100][:5]
    assert [int(idx[0, j]) for j in range(int(cnt[0]))] == exp


def test_partial_ratio(oracle, runner, tmp_path):
    import random
    rnd = random.Random(23)
    cps, offs = synth.make_corpus(150, max_chars=48, seed=21, vocab_size=30)
    cand = synth.to_strings(cps, offs) + ["", "a", "ab", "abc abc", "x" * 70]
    cand += ["".join(rnd.choice("abc ") for _ in range(rnd.randint(1, 12))) for _ in range(60)]
    queries = ["", "a", "abc", "b a c", cand[3], cand[3][:9], "abcabcabca", "x" * 33, "xy" * 40] + synth.make_queries(cps, offs, 3, seed=22, max_chars=48)
    got = _run(runner, tmp_path, 0, 3, cand, queries)
    exp = _oracle_scores(oracle, oracle.PARTIAL, queries, cand)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:8]]


def _wide_corpus(n, seed, max_tok=5):
    """names over > 254 distinct code points (CJK, Arabic, Cyrillic, ASCII): forces 16-bit symbols"""
    import random
    rnd = random.Random(seed)
    alpha = [chr(c) for c in range(0x4E00, 0x4E00 + 400)] + [chr(c) for c in range(0x0621, 0x064A)] + \
            [chr(c) for c in range(0x0430, 0x0450)] + list("abcdefghij0123456789")
    vocab = ["".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 6))) for _ in range(220)]
    return [" ".join(rnd.choice(vocab) for _ in range(rnd.randint(1, max_tok))) for _ in range(n)], vocab


def test_wide_alphabet_all_scorers(oracle, runner, tmp_path):
    cand, vocab = _wide_corpus(260, 31)
    cand += ["", vocab[0], "x" * 70, " ".join(vocab[:30])]
    assert len(set("".join(cand))) > 300
    queries = [cand[3], cand[10] + " " + vocab[5], vocab[7], "", "zz", "丁丂 abc", " ".join(vocab[10:32])]
    for sk, orc in ((0, oracle.TOKEN_SET), (1, oracle.RATIO), (2, oracle.TOKEN_SORT), (3, oracle.PARTIAL)):
        got = _run(runner, tmp_path, 0, sk, cand, queries)
        exp = _oracle_scores(oracle, orc, queries, cand)
        bad = np.argwhere(got != exp)
        assert bad.size == 0, (sk, [(queries[q], cand[i], int(got[q, i]), int(exp[q, i])) for q, i in bad[:5]])
    for kind, fn in ((0, oracle.indel), (1, oracle.levenshtein)):
        got = _run(runner, tmp_path, 1, kind, cand[:80], queries)
        for qi, q in enumerate(queries):
            for i, s in enumerate(cand[:80]):
                assert int(got[qi, i]) == fn(q, s), (kind, q, s)
    buf, o = oracle.pack(cand)
    idx, score, cnt = _run(runner, tmp_path, 2, 0, cand, queries, min_score=35, k=6)
    for qi, q in enumerate(queries):
        exp = oracle.topk(oracle.TOKEN_SET, q, buf, o, 35, 6)
        assert [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] == exp


def _run_pair(runner, tmp_path, cand_a, cand_b, q_a, q_b, tab, min_rank, k):
    """op 3: fz_query_topk_pair(PARTIAL on corpus a, TOKEN_SET on corpus b)."""
    ca, oa = pack_prepared(list(cand_a)); cb, ob = pack_prepared(list(cand_b))
    qa, qoa = pack_prepared(list(q_a)); qb, qob = pack_prepared(list(q_b))
    n, nq = len(oa) - 1, len(qoa) - 1
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    with open(case, "wb") as f:
        np.array([MAGIC, 3, 3 | (0 << 8), min_rank, k, n, int(oa[-1]), nq, int(qoa[-1])], dtype=np.uint64).tofile(f)
        ca[:int(oa[-1])].astype(np.uint32).tofile(f); oa.tofile(f); qa[:int(qoa[-1])].astype(np.uint32).tofile(f); qoa.tofile(f)
        np.array([int(ob[-1]), int(qob[-1])], dtype=np.uint64).tofile(f)
        cb[:int(ob[-1])].astype(np.uint32).tofile(f); ob.tofile(f); qb[:int(qob[-1])].astype(np.uint32).tofile(f); qob.tofile(f)
        np.ascontiguousarray(tab, dtype=np.uint16).tofile(f)
    e = dict(os.environ); e["ASAN_OPTIONS"] = "detect_leaks=0"
    r = subprocess.run([runner, case, out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    raw = np.fromfile(out, dtype=np.uint8)
    idx = raw[:nq * k * 8].view(np.uint64).reshape(nq, k)
    rank = raw[nq * k * 8:nq * k * 12].view(np.int32).reshape(nq, k)
    cnt = raw[nq * k * 12:].view(np.uint32)
    return idx, rank, cnt


def test_weighted_pair_topk(oracle, runner, tmp_path):
    """fz_query_topk_pair + rank table against the reference's weighted loop (app.py:995-1002)."""
    from tts_amd.preprocess import full_process
    from tts_amd.weighted_search import rank_table
    cps, offs = synth.make_corpus(420, max_chars=40, seed=41, vocab_size=25)
    keys = list(dict.fromkeys(synth.to_strings(cps, offs) + ["coke 1x24 (np)", "coke-zero 330ml", "a&b"]))
    queries = synth.make_queries(cps, offs, 4, seed=42, max_chars=40) + ["coke zero", keys[3][:5], "zz"]
    for thr, k in ((0.70, 10), (0.30, 7), (0.0, 300)):
        tab, vals, min_rank = rank_table(thr)
        idx, rank, cnt = _run_pair(runner, tmp_path, keys, [full_process(s) for s in keys], queries,
                                   [full_process(q) for q in queries], tab, min_rank, k)
        for qi, q in enumerate(queries):
            exp = [(i, oracle.weighted_score(q, s)) for i, s in enumerate(keys)]
            exp = [(i, v) for i, v in exp if v >= thr and q != keys[i]]
            exp.sort(key=lambda x: x[1], reverse=True)
            got = [(int(idx[qi, j]), vals[int(rank[qi, j]) - 1]) for j in range(int(cnt[qi]))]
            got = [(i, v) for i, v in got]
            exp_k = [(i, v) for i, v in [(i, oracle.weighted_score(q, s)) for i, s in enumerate(keys)] if v >= thr]
            exp_k.sort(key=lambda x: x[1], reverse=True)
            # the device scores a key equal to the query like any other (the host class special-cases it)
            if q not in keys:
                assert got == exp_k[:k], (thr, k, q, got[:4], exp_k[:4])


def test_first_passing_index(oracle, runner, tmp_path):
    """FZ_ORDER_BY_INDEX: the first candidates in list order whose ratio passes (app.py:1009-1011)."""
    cand = ["abcdefgh" * 13 + "x", "zzz", "abcdefgh" * 13 + "y", "abcdefgh" * 13, "abcdefgh" * 13 + "yy"] * 3
    q = "abcdefgh" * 13 + "z"
    idx, score, cnt = _run(runner, tmp_path, 2, 1 | 0x100, cand, [q, "nothing"], min_score=99, k=4)
    exp = [i for i, s in enumerate(cand) if oracle.score(oracle.RATIO, q, s) >= 99][:4]
    assert [int(idx[0, j]) for j in range(int(cnt[0]))] == exp and int(cnt[1]) == 0


def test_weighted_pair_many_ties(oracle, runner, tmp_path):
    """> SEL_TIECAP candidates with the same rank at the cut: the in-order walk of fz_select16_topk"""
    from tts_amd.weighted_search import rank_table
    keys = ["k%05d" % i for i in range(2600)] + ["zz top", "zz"]
    tab, vals, min_rank = rank_table(0.0)
    idx, rank, cnt = _run_pair(runner, tmp_path, keys, keys, ["zz", "qq"], ["zz", "qq"], tab, min_rank, 12)
    for qi, q in enumerate(["zz", "qq"]):
        exp = [(i, oracle.weighted_score(q, s)) for i, s in enumerate(keys) if q != s or True]
        exp = [(i, min(v, 0.99) if q == keys[i] else v) for i, v in exp]
        exp.sort(key=lambda x: x[1], reverse=True)
        got = [(int(idx[qi, j]), vals[int(rank[qi, j]) - 1]) for j in range(int(cnt[qi]))]
        assert got == exp[:12], (q, got, exp[:12])


def test_save_load_roundtrip(oracle, runner, tmp_path):
    """packed corpus file: scores of the loaded corpus == oracle; a truncated file is refused"""
    cps, offs = synth.make_corpus(300, max_chars=64, seed=51, vocab_size=40)
    cand = synth.to_strings(cps, offs) + ["", "x" * 90, "a b c"]
    queries = synth.make_queries(cps, offs, 4, seed=52) + ["", cand[1]]
    for sk, orc in ((0, oracle.TOKEN_SET), (2, oracle.TOKEN_SORT), (3, oracle.PARTIAL)):
        got = _run(runner, tmp_path, 4, sk, cand, queries, min_score=1 if sk == 0 else 0)
        assert (got == _oracle_scores(oracle, orc, queries, cand)).all(), sk
    wide, vocab = _wide_corpus(120, 53)
    got = _run(runner, tmp_path, 4, 0, wide, [wide[3], vocab[2]])
    assert (got == _oracle_scores(oracle, oracle.TOKEN_SET, [wide[3], vocab[2]], wide)).all()


def test_common_tokens_and_substring(oracle, runner, tmp_path):
    """fz_distance_all kind 2 (set intersection size) and kind 3 (Shift-And substring search)"""
    import random
    rnd = random.Random(61)
    toks = ["ab", "bc", "cd", "de", "ef", "zz", "q1", "w2", "abc", "bcd"]
    cand = [" ".join(rnd.choice(toks) for _ in range(rnd.randint(0, 12))) for _ in range(150)] + ["", "ab ab ab"]
    queries = ["ab bc cd", "", "zz", "q1 q1 w2 nope", " ".join(toks)]
    got = _run(runner, tmp_path, 1, 2, cand, queries)
    for qi, q in enumerate(queries):
        for i, s in enumerate(cand):
            assert int(got[qi, i]) == len(set(q.split()) & set(s.split())), (q, s)
    text = ["".join(rnd.choice("abc") for _ in range(rnd.randint(0, 90))) for _ in range(150)] + ["", "abc", "\x01abc"]
    pats = ["", "a", "abc", "cab", "abca" * 9, "\x01ab", "c" * 33, "ab" * 40, "x"]
    got = _run(runner, tmp_path, 1, 3, text, pats)
    for qi, q in enumerate(pats):
        for i, s in enumerate(text):
            assert int(got[qi, i]) == (1 if q in s else 0), (q, s)


def test_limits_long_strings(oracle, runner, tmp_path):
    """the longest supported shapes: 32-word patterns, candidates of thousands of symbols"""
    import random
    rnd = random.Random(71)
    words = ["".join(rnd.choice("abcdefgh") for _ in range(rnd.randint(2, 7))) for _ in range(400)]
    long_c = " ".join(rnd.choice(words) for _ in range(700))            # ~3500 chars
    long_q = " ".join(rnd.choice(words) for _ in range(170))[:1000].strip()
    cand = [long_c, long_q, "ab cd", "", " ".join(words[:150])] + [" ".join(rnd.choice(words) for _ in range(rnd.randint(1, 60))) for _ in range(70)]
    queries = [long_q, "ab cd", " ".join(words[:150])]
    got = _run(runner, tmp_path, 0, 0, cand, queries)
    assert (got == _oracle_scores(oracle, oracle.TOKEN_SET, queries, cand)).all()
    raw_q = long_q.replace(" ", "")[:1000]
    for kind, fn in ((0, oracle.indel), (1, oracle.levenshtein)):
        got = _run(runner, tmp_path, 1, kind, cand[:12], [raw_q, "abc"])
        for qi, q in enumerate([raw_q, "abc"]):
            for i, s in enumerate(cand[:12]):
                assert int(got[qi, i]) == fn(q, s), (kind, len(q), len(s))
