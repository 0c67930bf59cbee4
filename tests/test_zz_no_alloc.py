"""fz_query_topk_noalloc (the entry point that allocates nothing: static device scratch) against
fz_query_topk, through the C ABI, bit for bit -- two calls in a row, the second with a smaller input,
then a batch that needs several chunks of the scratch, then the limits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tts_amd import Corpus, _lib, synth  # noqa: E402


def _same(a, b):
    return all((x == y).all() for x, y in zip(a, b))


def test_noalloc_matches_topk_two_calls_in_a_row():
    cps, offs = synth.make_corpus(300_000, max_chars=64, seed=401)
    big = synth.make_queries(cps, offs, 48, seed=402)
    small = synth.make_queries(cps, offs, 3, seed=403) + [""]
    with Corpus(cps, offs) as c:
        # the second call is smaller in every dimension: fewer queries, smaller k; stale rows, keys and
        # counters of the first call are still in the array
        for queries, min_score, k in ((big, 35, 10), (small, 60, 4)):
            got = c.topk(queries, min_score, k, noalloc=True)
            exp = c.topk(queries, min_score, k)
            assert _same(got, exp), (len(queries), k)
        # 150 rows x 300 032 padded candidates = 45 MB of score rows: more than the array, several chunks
        many = synth.make_queries(cps, offs, 150, seed=404)
        assert _same(c.topk(many, 35, 10, noalloc=True), c.topk(many, 35, 10))
        # massive ties (a one-token query) and another scorer whose view an ordinary call has built
        tie_q = [synth.to_strings(cps, offs, 5, 6)[0].split()[0]]
        assert _same(c.topk(tie_q, 100, 64, noalloc=True), c.topk(tie_q, 100, 64))
        with pytest.raises(_lib.FuzzyHipError, match="not built"):
            c.topk(small, 50, 5, _lib.RATIO, noalloc=True)
        exp = c.topk(small, 50, 5, _lib.RATIO)
        assert _same(c.topk(small, 50, 5, _lib.RATIO, noalloc=True), exp)
    # a smaller corpus right after, on the same GPU: the same array again
    cps2, offs2 = synth.make_corpus(5_000, max_chars=32, seed=405)
    q2 = synth.make_queries(cps2, offs2, 7, seed=406, max_chars=32)
    with Corpus(cps2, offs2) as c2:
        assert _same(c2.topk(q2, 35, 5, noalloc=True), c2.topk(q2, 35, 5))


def test_noalloc_limits_are_refused():
    cps, offs = synth.make_corpus(2_000, max_chars=32, seed=411)
    with Corpus(cps, offs) as c:
        q = synth.make_queries(cps, offs, 40, seed=412, max_chars=32)
        with pytest.raises(_lib.FuzzyHipError, match="top-k keys"):
            c.topk(q, 0, 1000, noalloc=True)           # 40 x 1000 x 8 bytes of keys > FZ_NOALLOC_KEYS_BYTES
        with pytest.raises(_lib.FuzzyHipError, match="32768"):
            c.topk([""] * 40_000, 0, 1, noalloc=True)
    # one score row beyond the array: 40 million padded candidates would be needed; checked on a small
    # corpus through the message path instead (a 32 MiB row cannot be built in a unit test's time)
