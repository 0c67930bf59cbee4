/*
 * fuzz_oracle.c -- CPU restatement of the scorer arithmetic behind the reference's
 * candidate-scoring loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (tts_amd/, libfuzzyhip.so) never does.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or fixtures for this
 * path (reference tests/tests.py is a 0-byte file) and its scorer lives in third-party packages
 * that are absent from /root/reference and from this image:
 *     thefuzz   0.22.1  (reference uv.lock:4634-4635)
 *     rapidfuzz 3.14.1  (reference uv.lock:3842-3843)
 * The functions below restate the *published* algorithm of those two packages and are anchored
 * on the reference's call sites:
 *     src/fuzzy_search.py:26   fuzz.token_set_ratio(query_lower, item.lower())
 *     app.py:1010              fuzz.ratio(...)
 *     app.py:1023              fuzz.partial_ratio(...)
 *     app.py:1026              fuzz.token_set_ratio(...)
 *     app.py:296-317           AdvancedFuzzyMatcher.levenshtein_distance (in-tree Wagner-Fischer)
 * Distances (LCS / InDel / Levenshtein) are definitional; they are computed here by the plain
 * O(n*m) dynamic programme, and a bit-parallel variant (what rapidfuzz itself runs) is provided
 * beside it and checked against the DP in tests/.
 *
 * All strings are UTF-32 code points (uint32_t) + length.  Plain C, no dependencies.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t cp_t;

/* ------------------------------------------------------------------------------------------ */
/* whitespace: the set Python's str.split() / rapidfuzz's is_space() split on                  */
/* ------------------------------------------------------------------------------------------ */
int orc_is_space(cp_t c)
{
    if (c == 0x20) return 1;
    if (c >= 0x09 && c <= 0x0D) return 1;
    if (c >= 0x1C && c <= 0x1F) return 1;
    if (c == 0x85 || c == 0xA0 || c == 0x1680) return 1;
    if (c >= 0x2000 && c <= 0x200A) return 1;
    if (c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* distances                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* Longest common subsequence length, textbook two-row DP. */
int64_t orc_lcs_dp(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    if (na == 0 || nb == 0) return 0;
    int64_t *prev = (int64_t *)calloc((size_t)(nb + 1), sizeof(int64_t));
    int64_t *cur = (int64_t *)calloc((size_t)(nb + 1), sizeof(int64_t));
    for (int64_t i = 1; i <= na; ++i) {
        cur[0] = 0;
        for (int64_t j = 1; j <= nb; ++j) {
            if (a[i - 1] == b[j - 1])
                cur[j] = prev[j - 1] + 1;
            else
                cur[j] = prev[j] > cur[j - 1] ? prev[j] : cur[j - 1];
        }
        int64_t *t = prev; prev = cur; cur = t;
    }
    int64_t r = prev[nb];
    free(prev); free(cur);
    return r;
}

/* Same value by Hyyro's bit-parallel LCS recurrence:
 *   S = ~0;  for each text char c:  u = S & Peq[c];  S = (S + u) | (S - u);  LCS = popcount(~S)
 * multi-word: the + carries low->high across 64-bit words, the - never borrows (u is a subset of S).
 * Pattern = a, text = b.  Peq is looked up by linear search over the pattern's distinct code points. */
int64_t orc_lcs_bitpar(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    if (na == 0 || nb == 0) return 0;
    int64_t W = (na + 63) / 64;
    /* distinct code points of the pattern */
    cp_t *alpha = (cp_t *)malloc((size_t)na * sizeof(cp_t));
    int64_t nalpha = 0;
    for (int64_t i = 0; i < na; ++i) {
        int64_t k = 0;
        for (; k < nalpha; ++k) if (alpha[k] == a[i]) break;
        if (k == nalpha) alpha[nalpha++] = a[i];
    }
    uint64_t *peq = (uint64_t *)calloc((size_t)(nalpha * W), sizeof(uint64_t));
    for (int64_t i = 0; i < na; ++i) {
        int64_t k = 0;
        for (; k < nalpha; ++k) if (alpha[k] == a[i]) break;
        peq[k * W + i / 64] |= (uint64_t)1 << (i % 64);
    }
    uint64_t *S = (uint64_t *)malloc((size_t)W * sizeof(uint64_t));
    for (int64_t w = 0; w < W; ++w) S[w] = ~(uint64_t)0;
    for (int64_t j = 0; j < nb; ++j) {
        int64_t k = 0;
        for (; k < nalpha; ++k) if (alpha[k] == b[j]) break;
        if (k == nalpha) continue;            /* char not in pattern: Peq = 0, S unchanged */
        uint64_t carry = 0;
        for (int64_t w = 0; w < W; ++w) {
            uint64_t s = S[w];
            uint64_t u = s & peq[k * W + w];
            uint64_t x = s + carry;
            uint64_t c1 = x < s;
            uint64_t y = x + u;
            uint64_t c2 = y < x;
            carry = c1 | c2;
            S[w] = y | (s - u);
        }
    }
    int64_t lcs = 0;
    for (int64_t w = 0; w < W; ++w) {
        uint64_t z = ~S[w];
        if (w == W - 1 && (na % 64) != 0) z &= (((uint64_t)1 << (na % 64)) - 1);
        lcs += __builtin_popcountll(z);
    }
    free(alpha); free(peq); free(S);
    return lcs;
}

static int g_use_bitpar = 0;
/* tests flip this to run every scorer over either LCS implementation */
void orc_set_bitpar(int on) { g_use_bitpar = on; }

static int64_t lcs_len(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    return g_use_bitpar ? orc_lcs_bitpar(a, na, b, nb) : orc_lcs_dp(a, na, b, nb);
}

/* InDel distance = |a| + |b| - 2*LCS  (rapidfuzz Indel). */
int64_t orc_indel_distance(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    return na + nb - 2 * lcs_len(a, na, b, nb);
}

/* Uniform-cost Levenshtein, Wagner-Fischer two-row DP -- the algorithm of reference
 * app.py:296-317 (AdvancedFuzzyMatcher.levenshtein_distance). */
int64_t orc_levenshtein(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    if (na == 0) return nb;
    if (nb == 0) return na;
    int64_t *prev = (int64_t *)malloc((size_t)(nb + 1) * sizeof(int64_t));
    int64_t *cur = (int64_t *)malloc((size_t)(nb + 1) * sizeof(int64_t));
    for (int64_t j = 0; j <= nb; ++j) prev[j] = j;
    for (int64_t i = 1; i <= na; ++i) {
        cur[0] = i;
        for (int64_t j = 1; j <= nb; ++j) {
            int64_t ins = prev[j] + 1, del = cur[j - 1] + 1;
            int64_t sub = prev[j - 1] + (a[i - 1] != b[j - 1]);
            int64_t m = ins < del ? ins : del;
            cur[j] = m < sub ? m : sub;
        }
        int64_t *t = prev; prev = cur; cur = t;
    }
    int64_t r = prev[nb];
    free(prev); free(cur);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* normalisation -- the two fp64 op orders                           */
/* ------------------------------------------------------------------------------------------ */

/* rapidfuzz detail::norm_distance<100>(dist, lensum): 100 - 100*dist/lensum, lensum==0 -> 100.
 * Used by token_set_ratio (row a6). */
double orc_norm_distance_100(int64_t dist, int64_t lensum)
{
    volatile double max = 100.0;
    if (lensum <= 0) return max;
    volatile double num = max * (double)dist;
    volatile double quo = num / (double)lensum;
    volatile double res = max - quo;
    return res;
}

/* rapidfuzz fuzz::ratio = indel_normalized_similarity * 100 = (1.0 - dist/lensum) * 100,
 * lensum==0 -> 100 (row a8). */
double orc_norm_sim_100(int64_t dist, int64_t lensum)
{
    volatile double nd = (lensum > 0) ? (double)dist / (double)lensum : 0.0;
    volatile double ns = 1.0 - nd;
    volatile double res = ns * 100.0;
    return res;
}

/* thefuzz wrapper: int(round(x)) -- Python round() on a float is round-half-to-even (row a4). */
int orc_py_round(double x)
{
    return (int)nearbyint(x);   /* default rounding mode FE_TONEAREST = ties to even */
}

/* ------------------------------------------------------------------------------------------ */
/* tokens                                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const cp_t *p; int64_t n; } tok_t;

static int tok_cmp(const void *x, const void *y)
{
    const tok_t *a = (const tok_t *)x, *b = (const tok_t *)y;
    int64_t n = a->n < b->n ? a->n : b->n;
    for (int64_t i = 0; i < n; ++i) {
        if (a->p[i] != b->p[i]) return a->p[i] < b->p[i] ? -1 : 1;
    }
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    return 0;
}

/* split on whitespace, sort by code point; if dedupe, drop repeats.  returns token count */
static int64_t split_sorted(const cp_t *s, int64_t n, int dedupe, tok_t **out)
{
    tok_t *t = (tok_t *)malloc((size_t)(n / 2 + 2) * sizeof(tok_t));
    int64_t nt = 0, i = 0;
    while (i < n) {
        while (i < n && orc_is_space(s[i])) ++i;
        if (i >= n) break;
        int64_t st = i;
        while (i < n && !orc_is_space(s[i])) ++i;
        t[nt].p = s + st; t[nt].n = i - st; ++nt;
    }
    qsort(t, (size_t)nt, sizeof(tok_t), tok_cmp);
    if (dedupe && nt > 1) {
        int64_t k = 1;
        for (int64_t j = 1; j < nt; ++j)
            if (tok_cmp(&t[j], &t[k - 1]) != 0) t[k++] = t[j];
        nt = k;
    }
    *out = t;
    return nt;
}

/* ' '.join(tokens) */
static int64_t join_tokens(const tok_t *t, int64_t nt, cp_t *dst)
{
    int64_t n = 0;
    for (int64_t i = 0; i < nt; ++i) {
        if (i) dst[n++] = 0x20;
        memcpy(dst + n, t[i].p, (size_t)t[i].n * sizeof(cp_t));
        n += t[i].n;
    }
    return n;
}

/* Writes ' '.join(sorted(set(s.split()))) (dedupe=1) or ' '.join(sorted(s.split())) (dedupe=0)
 * into dst (capacity >= n); returns its length.  Exposed so tests can check the product's ingest. */
int64_t orc_canonical(const cp_t *s, int64_t n, int dedupe, cp_t *dst)
{
    tok_t *t; int64_t nt = split_sorted(s, n, dedupe, &t);
    int64_t r = join_tokens(t, nt, dst);
    free(t);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* scorers (each returns the un-rounded double, as rapidfuzz does; thefuzz rounds after)        */
/* ------------------------------------------------------------------------------------------ */

/* rapidfuzz.fuzz.ratio (reference call site app.py:1010). */
double orc_ratio(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    return orc_norm_sim_100(orc_indel_distance(a, na, b, nb), na + nb);
}

/* rapidfuzz.fuzz.token_sort_ratio: ratio of the sorted-token joins (duplicates kept). */
double orc_token_sort_ratio(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    cp_t *ja = (cp_t *)malloc((size_t)(na + 1) * sizeof(cp_t));
    cp_t *jb = (cp_t *)malloc((size_t)(nb + 1) * sizeof(cp_t));
    int64_t la = orc_canonical(a, na, 0, ja), lb = orc_canonical(b, nb, 0, jb);
    double r = orc_ratio(ja, la, jb, lb);
    free(ja); free(jb);
    return r;
}

/* rapidfuzz.fuzz.token_set_ratio (reference call site src/fuzzy_search.py:26).
 * Optional detail outputs (may be NULL): det[0]=sect_len det[1]=ab_len det[2]=ba_len det[3]=indel
 * det[4]=lensum0, or all -1 when an early return fired. */
double orc_token_set_ratio_detail(const cp_t *a, int64_t na, const cp_t *b, int64_t nb, int64_t *det)
{
    if (det) for (int i = 0; i < 5; ++i) det[i] = -1;
    tok_t *ta, *tb;
    int64_t nta = split_sorted(a, na, 1, &ta), ntb = split_sorted(b, nb, 1, &tb);
    double result = 0.0;
    if (nta == 0 || ntb == 0) { free(ta); free(tb); return 0.0; }

    tok_t *sect = (tok_t *)malloc((size_t)(nta + 1) * sizeof(tok_t));
    tok_t *dab = (tok_t *)malloc((size_t)(nta + 1) * sizeof(tok_t));
    tok_t *dba = (tok_t *)malloc((size_t)(ntb + 1) * sizeof(tok_t));
    int64_t ns = 0, nab = 0, nba = 0;
    for (int64_t i = 0; i < nta; ++i) {
        int found = 0;
        for (int64_t j = 0; j < ntb; ++j) if (tok_cmp(&ta[i], &tb[j]) == 0) { found = 1; break; }
        if (found) sect[ns++] = ta[i]; else dab[nab++] = ta[i];
    }
    for (int64_t j = 0; j < ntb; ++j) {
        int found = 0;
        for (int64_t i = 0; i < nta; ++i) if (tok_cmp(&ta[i], &tb[j]) == 0) { found = 1; break; }
        if (!found) dba[nba++] = tb[j];
    }
    if (ns > 0 && (nab == 0 || nba == 0)) {
        result = 100.0;                       /* one sentence is part of the other one */
    } else {
        cp_t *sa = (cp_t *)malloc((size_t)(na + 1) * sizeof(cp_t));
        cp_t *sb = (cp_t *)malloc((size_t)(nb + 1) * sizeof(cp_t));
        int64_t ab_len = join_tokens(dab, nab, sa);
        int64_t ba_len = join_tokens(dba, nba, sb);
        int64_t sect_len = 0;
        for (int64_t i = 0; i < ns; ++i) sect_len += sect[i].n + (i ? 1 : 0);
        int64_t e = sect_len != 0;
        int64_t sect_ab_len = sect_len + e + ab_len;
        int64_t sect_ba_len = sect_len + e + ba_len;
        int64_t dist = orc_indel_distance(sa, ab_len, sb, ba_len);
        result = orc_norm_distance_100(dist, sect_ab_len + sect_ba_len);
        if (det) { det[0] = sect_len; det[1] = ab_len; det[2] = ba_len; det[3] = dist;
                   det[4] = sect_ab_len + sect_ba_len; }
        if (sect_len) {
            double r1 = orc_norm_distance_100(e + ab_len, sect_len + sect_ab_len);
            double r2 = orc_norm_distance_100(e + ba_len, sect_len + sect_ba_len);
            if (r1 > result) result = r1;
            if (r2 > result) result = r2;
        }
        free(sa); free(sb);
    }
    free(sect); free(dab); free(dba); free(ta); free(tb);
    return result;
}

double orc_token_set_ratio(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    return orc_token_set_ratio_detail(a, na, b, nb, NULL);
}

/* rapidfuzz.fuzz.partial_ratio (reference call site app.py:1023), restated by its definition:
 * with s the shorter and l the longer string, the best ratio(s, w) over the windows w of l that
 * rapidfuzz considers: every full window l[i:i+|s|], every prefix l[:i] (i<|s|) and every suffix
 * l[i:] (i > |l|-|s|).  Equal lengths: rapidfuzz evaluates both directions (windows of b against a,
 * then windows of a against b) and keeps the better one; the two window sets differ, so both are
 * evaluated here too.  Empty s: 100 if both empty else 0.
 * LOW-CONFIDENCE restatement: window policy recalled from the published sources. */
static double partial_one_way(const cp_t *s, int64_t ns, const cp_t *l, int64_t nl)
{
    double best = 0.0;
    for (int64_t i = 1; i < ns; ++i) {                       /* prefixes shorter than |s| */
        double r = orc_ratio(s, ns, l, i);
        if (r > best) best = r;
    }
    for (int64_t i = 0; i + ns <= nl; ++i) {                 /* full windows */
        double r = orc_ratio(s, ns, l + i, ns);
        if (r > best) best = r;
    }
    for (int64_t i = nl - ns + 1; i < nl; ++i) {             /* suffixes shorter than |s| */
        double r = orc_ratio(s, ns, l + i, nl - i);
        if (r > best) best = r;
    }
    return best;
}

double orc_partial_ratio(const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    const cp_t *s = a, *l = b; int64_t ns = na, nl = nb;
    if (ns > nl) { s = b; ns = nb; l = a; nl = na; }
    if (ns == 0) return nl == 0 ? 100.0 : 0.0;
    double best = partial_one_way(s, ns, l, nl);
    if (ns == nl) {
        double r = partial_one_way(l, nl, s, ns);
        if (r > best) best = r;
    }
    return best;
}

/* ------------------------------------------------------------------------------------------ */
/* the loop of reference src/fuzzy_search.py:22-38 over already-preprocessed strings             */
/* ------------------------------------------------------------------------------------------ */

/* scorer ids shared with tests */
enum { ORC_TOKEN_SET = 0, ORC_RATIO = 1, ORC_TOKEN_SORT = 2, ORC_PARTIAL = 3 };

int orc_score_int(int scorer, const cp_t *a, int64_t na, const cp_t *b, int64_t nb)
{
    double x;
    switch (scorer) {
    case ORC_TOKEN_SET: x = orc_token_set_ratio(a, na, b, nb); break;
    case ORC_RATIO: x = orc_ratio(a, na, b, nb); break;
    case ORC_TOKEN_SORT: x = orc_token_sort_ratio(a, na, b, nb); break;
    case ORC_PARTIAL: x = orc_partial_ratio(a, na, b, nb); break;
    default: return -1;
    }
    return orc_py_round(x);
}

/* all scores of one query against a packed corpus (cps + offs[n+1]) */
void orc_score_all(int scorer, const cp_t *q, int64_t nq, const cp_t *cps, const uint64_t *offs,
                   int64_t n, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = orc_score_int(scorer, q, nq, cps + offs[i], (int64_t)(offs[i + 1] - offs[i]));
}

/* score every candidate, keep score >= min_score, stable sort descending, slice k
 * (src/fuzzy_search.py:31-38: ties keep ascending candidate-list order).  returns count. */
int64_t orc_topk(int scorer, const cp_t *q, int64_t nq, const cp_t *cps, const uint64_t *offs,
                 int64_t n, int32_t min_score, int64_t k, int64_t *out_idx, int32_t *out_score)
{
    int64_t cnt = 0;
    if (k <= 0) return 0;
    for (int64_t i = 0; i < n; ++i) {
        int s = orc_score_int(scorer, q, nq, cps + offs[i], (int64_t)(offs[i + 1] - offs[i]));
        if (s < min_score) continue;
        /* insertion position: after every kept entry with score >= s (stability) */
        int64_t pos = cnt;
        while (pos > 0 && out_score[pos - 1] < s) --pos;
        if (pos >= k) continue;
        int64_t last = cnt < k ? cnt : k - 1;
        for (int64_t j = last; j > pos; --j) { out_idx[j] = out_idx[j - 1]; out_score[j] = out_score[j - 1]; }
        out_idx[pos] = i; out_score[pos] = s;
        if (cnt < k) ++cnt;
    }
    return cnt;
}
