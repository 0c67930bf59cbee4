"""Drop-in for the reference's ``AdvancedFuzzyMatcher`` (reference ``app.py:265-460``; the monolith's
live caller is ``app.py:1847``): the best of six weighted strategies per candidate, with the exact
and unit-conversion shortcuts in front.  Every per-candidate string algorithm runs on the MI355X:

    strategy (app.py)                          device work (fz_distance_all on a resident corpus)
    levenshtein_distance       :296-317        kind 1, Myers/Hyyro bit-vector Levenshtein
    substring_match_score      :337-345        kind 3, Shift-And search of query_norm
    prefix_match_score         :320-334        kind 3 with a sentinel in front of both sides, on the
                                               normalised text and on the normalised words
    jaccard_similarity (n=2,3) :282-293        kind 2, |A & B| over the distinct n-grams
    SequenceMatcher.ratio      :421-424        kind 0 (InDel) gives 2*LCS/T, an upper bound of the
                                               Ratcliff/Obershelp ratio; difflib itself (stdlib) then
                                               runs only for candidates the bound cannot rule out

The fp64 arithmetic of the reference (weights, ``max``, threshold, stable sort, slice) is evaluated
on the host from the device's integers, with the same expressions, so scores are bit-identical.
Exact match (:399-405) is a dict lookup; the unit-conversion shortcut (:348-376, :408-415) is regex
work done once per candidate at ingest and vectorised per query.

There is no CPU path for the per-candidate loops: building an index raises without the library or a
GPU."""
from __future__ import annotations

import re
import threading
from collections import OrderedDict
from difflib import SequenceMatcher
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .corpus import Corpus, LeasedResource, list_fingerprint

SENTINEL = "\x01"            # never produced by normalize_text ([a-z0-9] only)

# reference app.py:253-262
FUZZY_MATCH_CONFIG = {
    'threshold': 0.7,
    'top_n': 5,
    'prefix_weight': 1.0,
    'substring_weight': 0.9,
    'sequence_weight': 0.85,
    'jaccard_weight': 0.8,
    'levenshtein_weight': 0.8,
    'unit_conversion_weight': 0.95,
}

# reference app.py:116-128, built from (family members, size of one unit in the family's base unit)
def _conversion_table() -> Dict[str, Dict[str, float]]:
    table: Dict[str, Dict[str, float]] = {}
    litre = {'ml': 1000, 'l': 1, 'lt': 1, 'ltr': 1}
    table['ml'] = {'ml': 1, 'l': 0.001, 'lt': 0.001, 'ltr': 0.001}
    for u in ('l', 'lt', 'ltr'):
        table[u] = dict(litre)
    table['g'] = {'g': 1, 'kg': 0.001}
    table['kg'] = {'g': 1000, 'kg': 1}
    for u in ('pcs', 'pc'):
        table[u] = {'pcs': 1, 'pc': 1, 'piece': 1, 'pieces': 1}
    return table


UNIT_CONVERSIONS = _conversion_table()

_QTY = re.compile(r'(\d+\.?\d*)\s*([a-zA-Z]+)')
_QTY_STRIP = re.compile(r'\d+\.?\d*\s*[a-zA-Z]+')
_NOT_ALNUM = re.compile(r'[^a-z0-9]')


class UnitConverter:
    """reference app.py:130-250 (the parts the matcher uses)."""

    @staticmethod
    def extract_quantity_and_unit(text: str) -> Optional[Tuple[float, str]]:
        m = _QTY.search(text.lower())
        if not m:
            return None
        try:
            return float(m.group(1)), m.group(2).lower()
        except (ValueError, AttributeError):
            return None

    @staticmethod
    def normalize_unit(unit: str) -> Optional[str]:
        u = unit.lower()
        if u in UNIT_CONVERSIONS:
            return u
        for standard, conv in UNIT_CONVERSIONS.items():
            if u in conv:
                return standard
        return None

    @staticmethod
    def convert_quantity(quantity: float, from_unit: str, to_unit: str) -> Optional[float]:
        f, t = UnitConverter.normalize_unit(from_unit), UnitConverter.normalize_unit(to_unit)
        if not f or not t:
            return None
        if f not in UNIT_CONVERSIONS or t not in UNIT_CONVERSIONS[f]:
            return None
        return quantity * UNIT_CONVERSIONS[f][t]

    @staticmethod
    def are_quantities_equivalent(text1: str, text2: str, tolerance: float = 0.01) -> bool:
        a, b = UnitConverter.extract_quantity_and_unit(text1), UnitConverter.extract_quantity_and_unit(text2)
        if not a or not b:
            return False
        (q1, u1), (q2, u2) = a, b
        if not UnitConverter.normalize_unit(u1) or not UnitConverter.normalize_unit(u2):
            return False
        converted = UnitConverter.convert_quantity(q1, u1, u2)
        if converted is None:
            return False
        return abs(converted - q2) <= (q2 * tolerance)


def normalize_text(text: str) -> str:
    """reference app.py:271-273"""
    return _NOT_ALNUM.sub('', text.lower())


def _ngrams(norm: str, n: int) -> set:
    return set(norm[i:i + n] for i in range(len(norm) - n + 1))


class CandidateIndex:
    """One candidate list, preprocessed once and resident in HBM in four forms."""

    def __init__(self, candidates: Sequence[str], device: int = 0):
        self.candidates = candidates
        self.n = len(candidates)
        lower = [c.lower() for c in candidates]
        norm = [_NOT_ALNUM.sub('', s) for s in lower]
        self.lower_index: Dict[str, List[int]] = {}
        for i, s in enumerate(lower):
            self.lower_index.setdefault(s, []).append(i)
        self.len_norm = np.fromiter((len(s) for s in norm), dtype=np.int64, count=self.n)
        bi = [_ngrams(s, 2) for s in norm]
        tri = [_ngrams(s, 3) for s in norm]
        self.n_bi = np.fromiter((len(s) for s in bi), dtype=np.int64, count=self.n)
        self.n_tri = np.fromiter((len(s) for s in tri), dtype=np.int64, count=self.n)
        self.norm = norm
        self.c_norm = Corpus.from_prepared([SENTINEL + s for s in norm], device=device)
        self.c_words = Corpus.from_prepared(
            [SENTINEL + SENTINEL.join(_NOT_ALNUM.sub('', w) for w in s.split()) for s in lower], device=device)
        self.c_bi = Corpus.from_prepared([" ".join(sorted(s)) for s in bi], device=device)
        self.c_tri = Corpus.from_prepared([" ".join(sorted(s)) for s in tri], device=device)
        # unit-conversion shortcut: quantity and normalised unit of every candidate that has one
        qty = np.zeros(self.n, dtype=np.float64)
        unit: List[Optional[str]] = [None] * self.n
        for i, c in enumerate(candidates):
            qu = UnitConverter.extract_quantity_and_unit(c)
            if qu:
                qty[i] = qu[0]
                unit[i] = UnitConverter.normalize_unit(qu[1]) or ""      # "": has a quantity, unknown unit
        self.qty = qty
        self.unit = unit
        self.has_qty = np.fromiter((u is not None for u in unit), dtype=bool, count=self.n)
        self.units_present = sorted({u for u in unit if u})
        self.unit_id = np.fromiter((self.units_present.index(u) if u else -1 for u in unit), dtype=np.int64, count=self.n)

    def close(self) -> None:
        for c in (self.c_norm, self.c_words, self.c_bi, self.c_tri):
            c.close()


class AdvancedFuzzyMatcher:
    """Advanced Fuzzy Matching Engine with unit conversion support (reference app.py:265-460)."""

    _indexes: "OrderedDict[int, Tuple[tuple, LeasedResource]]" = OrderedDict()
    _max_indexes = 8
    _lock = threading.Lock()
    device = 0
    SEQUENCE_ON_GPU_FROM = 48        # survivors from which SequenceMatcher's matches are computed by the library (one pass
                                     # over the list costs about as much as this many difflib runs)

    normalize_text = staticmethod(normalize_text)

    @staticmethod
    def get_ngrams(text: str, n: int) -> set:
        return _ngrams(normalize_text(text), n)

    # -- index cache: the reference passes the list on every call (app.py:1847) ---------------------
    @classmethod
    def _lease_index(cls, candidates: Sequence[str]):
        """-> (index, lease).  The index of a list is reused while corpus.list_fingerprint says the list is
        unchanged; an index that was replaced or evicted is closed only once no call is still using it."""
        key = id(candidates)
        fp = list_fingerprint(candidates)
        with cls._lock:
            ent = cls._indexes.get(key)
            ix = ent[1].lease() if ent is not None and ent[0] == fp else None
            if ix is not None:
                cls._indexes.move_to_end(key)
                return ix, ent[1]
            if ent is not None:
                ent[1].retire()
            res = LeasedResource(CandidateIndex(list(candidates), device=cls.device))
            cls._indexes[key] = (fp, res)
            cls._indexes.move_to_end(key)
            while len(cls._indexes) > cls._max_indexes:
                _, (_, old) = cls._indexes.popitem(last=False)
                old.retire()
            return res.lease(), res

    @classmethod
    def index_for(cls, candidates: Sequence[str]) -> CandidateIndex:
        ix, lease = cls._lease_index(candidates)
        lease.release()
        return ix

    @classmethod
    def clear_cache(cls) -> None:
        with cls._lock:
            for _, res in cls._indexes.values():
                res.retire()
            cls._indexes.clear()

    # -- unit conversion shortcut (app.py:348-376), vectorised over the candidates ------------------
    @staticmethod
    def _unit_scores(query: str, ix: CandidateIndex) -> Dict[int, float]:
        qu = UnitConverter.extract_quantity_and_unit(query)
        if not qu:
            return {}
        q1, u1 = qu
        f = UnitConverter.normalize_unit(u1)
        if not f:
            return {}
        out: Dict[int, float] = {}
        for uid, t in enumerate(ix.units_present):
            if t not in UNIT_CONVERSIONS[f]:
                continue
            converted = q1 * UNIT_CONVERSIONS[f][t]
            sel = np.nonzero(ix.unit_id == uid)[0]
            q2 = ix.qty[sel]
            hit = sel[np.abs(converted - q2) <= (q2 * 0.01)]
            if hit.size == 0:
                continue
            query_no_qty = _QTY_STRIP.sub('', query).strip()
            for i in hit.tolist():
                target_no_qty = _QTY_STRIP.sub('', ix.candidates[i]).strip()
                if query_no_qty.lower() == target_no_qty.lower():
                    score = 1.0
                else:
                    similarity = SequenceMatcher(None, normalize_text(query_no_qty), normalize_text(target_no_qty)).ratio()
                    score = similarity if similarity > 0.6 else 0.0
                if score > 0:
                    out[i] = score
        return out

    # -- app.py:378-460 -----------------------------------------------------------------------------
    @classmethod
    def fuzzy_search(cls, query: str, candidates: List[str], threshold: float = None, top_n: int = None) -> List[Dict]:
        if threshold is None:
            threshold = FUZZY_MATCH_CONFIG['threshold']
        if top_n is None:
            top_n = FUZZY_MATCH_CONFIG['top_n']
        if not query or not candidates:
            return []
        ix, lease = cls._lease_index(candidates)
        try:
            return cls._search(query, candidates, ix, threshold, top_n)
        finally:
            lease.release()

    @classmethod
    def _search(cls, query: str, candidates: List[str], ix: CandidateIndex, threshold: float, top_n: int) -> List[Dict]:
        results: List[Tuple[int, Dict]] = []

        special = np.zeros(ix.n, dtype=bool)
        for i in ix.lower_index.get(query.lower(), ()):
            results.append((i, {'value': ix.candidates[i], 'score': 1.0, 'method': 'exact'}))
            special[i] = True
        for i, unit_score in cls._unit_scores(query, ix).items():
            if special[i]:
                continue
            results.append((i, {'value': ix.candidates[i],
                                'score': unit_score * FUZZY_MATCH_CONFIG['unit_conversion_weight'],
                                'method': 'unit_conversion'}))
            special[i] = True

        qn = normalize_text(query)
        la = len(qn)
        q_bi, q_tri = _ngrams(qn, 2), _ngrams(qn, 3)
        lev = ix.c_norm.distance_all([SENTINEL + qn], 1)[0].astype(np.int64)
        indel = ix.c_norm.distance_all([SENTINEL + qn], 0)[0].astype(np.int64)
        starts = ix.c_norm.distance_all([SENTINEL + qn], 3)[0] != 0
        inside = ix.c_norm.distance_all([qn], 3)[0] != 0
        word_starts = ix.c_words.distance_all([SENTINEL + qn], 3)[0] != 0
        inter2 = ix.c_bi.distance_all([" ".join(sorted(q_bi))], 2)[0].astype(np.int64)
        inter3 = ix.c_tri.distance_all([" ".join(sorted(q_tri))], 2)[0].astype(np.int64)

        lb = ix.len_norm
        todo = ~special
        if la == 0 and bool(np.any(todo & (lb == 0))):
            raise ZeroDivisionError("division by zero")       # len(query_norm) / len(target_norm), app.py:343
        with np.errstate(divide='ignore', invalid='ignore'):
            prefix = np.where(starts, 1.0, np.where(word_starts, 0.9, 0.0))
            substring = np.where(inside, la / np.where(lb == 0, 1, lb), 0.0)

            def jaccard(inter, na, nb):
                union = na + nb - inter
                return np.where((na > 0) & (nb > 0) & (union > 0), inter / np.where(union == 0, 1, union), 0.0)

            j2 = jaccard(inter2, len(q_bi), ix.n_bi)
            j3 = jaccard(inter3, len(q_tri), ix.n_tri)
            max_len = np.maximum(la, lb)
            lev_score = np.where(max_len > 0, 1 - (lev / np.where(max_len == 0, 1, max_len)), 0.0)
            total = la + lb
            lcs = (total - indel) // 2
            seq_bound = np.where(total > 0, 2.0 * lcs / np.where(total == 0, 1, total), 1.0)   # >= SequenceMatcher.ratio()
        cfg = FUZZY_MATCH_CONFIG
        others = np.maximum.reduce([prefix * cfg['prefix_weight'], substring * cfg['substring_weight'],
                                    j2 * cfg['jaccard_weight'], j3 * 0.75, lev_score * cfg['levenshtein_weight']])
        seq_ub = seq_bound * cfg['sequence_weight']
        could_pass = todo & ((others >= threshold) | (seq_ub >= threshold))
        # Only out[:top_n] is returned (app.py:459-460: sort by score, slice).  `others` is a LOWER bound of a candidate's
        # final score and max(others, seq_ub) an UPPER bound; the exact, exact-key and unit-conversion results have their
        # score already.  A candidate whose upper bound lies strictly below the top_n-th largest lower bound ranks behind
        # top_n candidates with strictly larger scores whatever difflib says: it needs no SequenceMatcher run.  (A short,
        # popular query otherwise sends thousands of candidates through difflib: 167 ms against 2.7 ms p50.)
        if top_n is not None and top_n > 0:
            sure = np.concatenate([np.asarray([r['score'] for _, r in results], dtype=np.float64),
                                   others[todo & (others >= threshold)]])
            if sure.size >= top_n:
                cut = np.partition(sure, sure.size - top_n)[sure.size - top_n]
                could_pass &= np.maximum(others, seq_ub) >= cut
        survivors = np.nonzero(could_pass)[0].tolist()
        # many survivors (a short, popular query): the matching characters M of SequenceMatcher come from the GPU
        # (fz_distance_all kind 4, Ratcliff/Obershelp with difflib's tie rules; ratio() = 2.0 * M / (|a| + |b|), difflib's own
        # expression).  The candidate side carries the sentinel, which the query lacks, so it matches nothing.  0xFFFF: a
        # candidate beyond the kernel's 64 symbols -- difflib itself.  A handful of survivors: difflib is quicker than a pass
        seq_m = None
        if len(survivors) > cls.SEQUENCE_ON_GPU_FROM and 0 < la <= 64:
            seq_m = ix.c_norm.distance_all([qn], 4)[0]
        seq_of = None
        if seq_m is not None:
            # ... and with thousands of survivors the final scores are taken for all of them at once (the same fp64
            # operations, element by element), so that the per-candidate Python below only runs for those that can be returned
            sv = np.asarray(survivors, dtype=np.int64)
            mm = seq_m[sv].astype(np.int64)
            seq = np.zeros(sv.size, dtype=np.float64)
            ok = mm != 0xFFFF
            seq[ok] = 2.0 * mm[ok] / (la + lb[sv][ok])
            for t in np.nonzero(~ok)[0].tolist():
                seq[t] = SequenceMatcher(None, qn, ix.norm[int(sv[t])]).ratio()
            final = np.maximum.reduce([prefix[sv] * cfg['prefix_weight'], substring[sv] * cfg['substring_weight'],
                                       seq * cfg['sequence_weight'], j2[sv] * cfg['jaccard_weight'], j3[sv] * 0.75,
                                       lev_score[sv] * cfg['levenshtein_weight']])
            pick = np.nonzero(final >= threshold)[0]
            if top_n is not None and top_n > 0 and pick.size > top_n:     # out[:top_n]: score descending, candidate order among equals
                pick = np.sort(pick[np.argsort(-final[pick], kind='stable')[:top_n]])
            survivors = sv[pick].tolist()
            seq_of = dict(zip(survivors, seq[pick].tolist()))
        for i in survivors:
            if seq_of is not None:
                sequence_score = seq_of[i]
            else:
                sequence_score = SequenceMatcher(None, qn, ix.norm[i]).ratio()
            final_score = max(prefix[i] * cfg['prefix_weight'], substring[i] * cfg['substring_weight'],
                              sequence_score * cfg['sequence_weight'], j2[i] * cfg['jaccard_weight'],
                              j3[i] * 0.75, lev_score[i] * cfg['levenshtein_weight'])
            if final_score >= threshold:
                method_scores = {'prefix': prefix[i], 'substring': substring[i], 'sequence': sequence_score,
                                 'jaccard': max(j2[i], j3[i]), 'levenshtein': lev_score[i]}
                best_method = max(method_scores, key=method_scores.get)
                results.append((i, {'value': ix.candidates[i], 'score': float(final_score), 'method': best_method}))

        results.sort(key=lambda t: t[0])                       # the reference appends in candidate order
        out = [r for _, r in results]
        out.sort(key=lambda x: x['score'], reverse=True)
        return out[:top_n]
