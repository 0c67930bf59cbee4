/*
 * fuzzyhip.h -- C ABI of libfuzzyhip.so: the MI355X (gfx950) implementation of the reference's
 * candidate-scoring loop, FuzzySearch.find_best_matches (reference src/fuzzy_search.py:16-38).
 *
 * The reference has no FFI of its own for this path: its boundary is the Python class
 * `FuzzySearch` (src/fuzzy_search.py:8-38).  tts_amd/fuzzy_search.py keeps that class and its two
 * signatures byte for byte and binds the entry points below with ctypes; INTEGRATION.md shows the
 * binding.  Each declaration cites the reference lines it stands in for.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 on success or a negative FZ_ERR_*
 *     code, and fz_last_error() then returns a message for the calling thread.
 *   - strings are packed UTF-32 code points + offsets (string i = cps[offs[i] .. offs[i+1]));
 *     they are ALREADY preprocessed by the host language (lower-casing / full_process depend on
 *     Python's Unicode tables and stay in Python: SURVEY.md 8b).
 *   - candidate indices are 64-bit GLOBAL indices: base_index + position in this shard.
 *   - there is no CPU fallback: every scoring entry point needs a gfx950 device.
 */
#ifndef FUZZYHIP_H
#define FUZZYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZ_VERSION 100 /* 0.1.0 */

enum {
    FZ_OK = 0,
    FZ_ERR_INVALID = -1,   /* bad argument */
    FZ_ERR_HIP = -2,       /* a HIP runtime call failed (message has the HIP error string) */
    FZ_ERR_ALPHABET = -3,  /* corpus has more distinct code points than the symbol table holds */
    FZ_ERR_TOO_LONG = -4,  /* query pattern / candidate longer than the kernels support */
    FZ_ERR_LIMIT = -5,     /* k or another size beyond a documented limit */
    FZ_ERR_NO_DEVICE = -6, /* no usable GPU */
    FZ_ERR_COMM = -7       /* librccl could not be opened, or an RCCL call failed (message has its error string) */
};

/* scorers; names follow thefuzz.fuzz (the library the reference calls) */
typedef enum {
    FZ_TOKEN_SET = 0,   /* fuzz.token_set_ratio  -- reference src/fuzzy_search.py:26, app.py:1026 */
    FZ_RATIO = 1,       /* fuzz.ratio            -- reference app.py:1010 */
    FZ_TOKEN_SORT = 2,  /* fuzz.token_sort_ratio -- BASELINE.json config 5 */
    FZ_PARTIAL = 3      /* fuzz.partial_ratio    -- reference app.py:1023 */
} fz_scorer;
/* The uniform-cost Levenshtein of reference app.py:296-317 is not a 0..100 scorer (app.py:429-432 turns the distance
 * into an unrounded fp64 similarity on the host): its distances come from fz_distance_all / fz_distance_batch (kind 1). */

/* OR into `scorer` of fz_query_topk / fz_query_topk_keys: every candidate that passes min_score
 * counts as a tie, so the results are the FIRST k passing candidates in list order (their score is
 * reported as min_score).  Stands in for the first-hit scan of the reference's weighted FuzzySearch,
 * `for ...: if fuzz.ratio(q, item) > 98: return item` (reference app.py:1009-1011). */
#define FZ_ORDER_BY_INDEX 0x100

/* limits */
#define FZ_MAX_K 1024u            /* top-k size */
#define FZ_MAX_PATTERN 1024u      /* query pattern length in code points incl. token separators */
#define FZ_MAX_CAND_LEN 16000u    /* canonical candidate length in code points */
#define FZ_MAX_IN_FLIGHT 2u       /* fz_query_topk_submit tickets of one corpus not yet fetched */

/* fz_query_topk_noalloc: size of the library's static device scratch array (held by every GPU the library
 * runs on until the process ends) and of the pinned host staging each corpus sets aside at creation */
#define FZ_NOALLOC_SCRATCH_BYTES (32u << 20)
#define FZ_NOALLOC_BLOB_BYTES (1u << 20)    /* query tables of one chunk */
#define FZ_NOALLOC_KEYS_BYTES (256u << 10)  /* nq * k * 8 bytes of result keys */

typedef struct fz_corpus fz_corpus;

typedef struct {
    uint64_t n;              /* candidates in this shard */
    uint64_t base_index;     /* global index of candidate 0 */
    uint64_t vocab_size;     /* distinct tokens */
    uint32_t alphabet_size;  /* distinct code points (+2 reserved symbols) */
    uint32_t n_tiles;        /* 64-candidate tiles */
    uint32_t max_len;        /* longest canonical candidate (symbols, incl. trailing separator) */
    uint32_t max_tokens;     /* most tokens in one candidate */
    uint64_t device_bytes;   /* HBM held by the corpus (without the query workspace) */
    uint64_t total_symbols;  /* sum of canonical lengths without the trailing separator */
    uint64_t total_tokens;   /* sum of token counts */
    int32_t device;
    /* how the tiles of the token_set view are dealt to the kernels: pairs of short tiles sharing one pattern word
     * (fz_score_stream_pk), tiles with one / two pattern words of their own (fz_score_stream<1> / <2>), and tiles
     * only the general kernel takes (long or many-token candidates, wide alphabets) */
    uint32_t n_packed_pairs, n_stream1_tiles, n_stream2_tiles, n_general_tiles;
} fz_info;

typedef struct {
    float score_ms;     /* hipEvent time of the scoring kernels of the last query call */
    float select_ms;    /* hipEvent time of the top-k selection kernels of the last call */
    float total_ms;     /* hipEvent time from first upload to last download of the last call */
    uint32_t score_launches;
    uint32_t select_launches;
    uint64_t pairs;     /* query x candidate pairs scored by the last call */
    double host_prep_ms;/* wall time spent building query tables on the host */
    uint32_t fused;     /* 1: the last top-k call ran fused (sample + append lists, no score matrix over the shard) */
    uint32_t fused_fallback_chunks; /* chunks of that call redone through the score matrix (list overflow) */
    float exchange_ms;  /* fz_query_topk_sharded: hipEvent time of the all-gather + device merge (0 otherwise) */
    uint32_t fused_fallback_rows;   /* rows of that call whose appended list overflowed (redone through the score matrix) */
} fz_timing;

/* library version (FZ_VERSION of the build) */
int fz_version(void);

/* message of the last error raised on the calling thread ("" when none) */
const char *fz_last_error(void);

/* number of visible HIP devices (0 and FZ_ERR_NO_DEVICE when none) */
int fz_device_count(int *out_n);

/*
 * fz_corpus_create -- stands in for FuzzySearch.__init__ (reference src/fuzzy_search.py:10-14),
 * which only stores the candidate list.  Here the list is tokenised, each candidate is reduced to
 * its canonical form (sorted unique tokens, what token_set_ratio compares), mapped to 8-bit symbols,
 * length-sorted into 64-candidate tiles (code-point-major, so loads coalesce) and uploaded to HBM
 * on `device`.  The candidates are immutable afterwards (reference: the lists are never mutated).
 *   cps/offs : n preprocessed strings, offs has n+1 entries
 *   base_index : global index of candidate 0 (contiguous index-range sharding)
 */
int fz_corpus_create(const uint32_t *cps, const uint64_t *offs, uint64_t n, int device,
                     uint64_t base_index, fz_corpus **out);

/* frees device and host memory of a corpus (reference: garbage collection of the lists) */
void fz_corpus_destroy(fz_corpus *c);

int fz_corpus_info(const fz_corpus *c, fz_info *out);

/*
 * fz_corpus_save / fz_corpus_load -- a packed corpus file holding the built device layout (symbols,
 * token ids, tiles, permutations, vocabulary), so that a process start is read + upload instead of
 * tokenise + sort + transpose.  The reference re-fetches its lists from the database on every start
 * (src/synapse_connection.py:103-121, called from src/synapse_agent.py:68-69) and has no such file;
 * this is the natural unit to shard a 10^8-candidate corpus by.  `base_index` = ~0 keeps the base
 * index the file was saved with.  Every offset, symbol and token end of the file is range-checked
 * on load before it reaches the device.
 */
int fz_corpus_save(fz_corpus *c, const char *path);
int fz_corpus_load(const char *path, int device, uint64_t base_index, fz_corpus **out);

/*
 * run later kernels on the caller's hipStream_t (NULL = the corpus' own stream).
 * ONE HIP RUNTIME PER PROCESS: the stream must come from the very libamdhip64 this library is bound to
 * (fz_runtime_paths names it).  PyTorch wheels ship their own copy of libamdhip64 / libhsa-runtime64 / librccl;
 * a process that loads this library first and torch afterwards maps BOTH runtimes, the second one finds no
 * device, and a stream or device pointer of one is meaningless to the other.  Rule: import torch BEFORE this
 * library is loaded (the dynamic loader then binds this library to torch's runtime by its SONAME), or keep
 * torch out of the process.  tts_amd/_lib.py checks /proc/self/maps at load time and wherever a foreign
 * stream or pointer comes in, and raises with both paths named.
 */
int fz_set_stream(fz_corpus *c, void *hip_stream);

/* path of the HIP runtime this library is bound to, and of the librccl it opened ("" before the first
 * fz_comm_* call); NUL-terminated, truncated to the capacities given.  No counterpart in the reference. */
int fz_runtime_paths(char *hip_out, size_t hip_cap, char *rccl_out, size_t rccl_cap);

/* bound on the device workspace the score matrix may take (bytes; default 4 GiB) */
int fz_set_workspace_limit(fz_corpus *c, uint64_t bytes);

/*
 * fz_query_topk -- the loop, threshold, stable sort and slice of find_best_matches
 * (reference src/fuzzy_search.py:22-38) for nq queries at once.
 *   min_score : smallest integer score that passes `score/100.0 >= threshold` (host computes it
 *               with Python floats: src/fuzzy_search.py:29-31)
 *   k         : TOP_N_MATCHES_INT (src/fuzzy_search.py:38), <= FZ_MAX_K
 *   out_idx   : nq*k global candidate indices, row q sorted by (score desc, index asc) -- the order
 *               Python's stable sort(reverse=True) yields (src/fuzzy_search.py:35)
 *   out_score : nq*k integer scores 0..100 (the reference's similarity_score before /100.0)
 *   out_cnt   : nq result counts (<= k)
 * Empty query, empty corpus, k == 0: zero results, not an error.
 */
int fz_query_topk(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                  int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx, int32_t *out_score,
                  uint32_t *out_cnt);

/*
 * fz_query_topk_submit / fz_query_topk_wait -- fz_query_topk in two phases, for callers that have the next batch ready
 * while the GPU still works on this one (a serving loop over a queue of batches).  The reference's call is synchronous
 * (src/fuzzy_search.py:16-38) and so is fz_query_topk; these two add nothing to its semantics.  submit builds the query
 * tables, enqueues the upload, every kernel and the copy of the result on the corpus' stream and returns a ticket without
 * waiting for the GPU; wait blocks until that call's result has arrived and decodes it into the caller's arrays (same
 * meaning and order as fz_query_topk's, bit for bit).  Between them the host is free: the table building of call i+1
 * (0.13 ms per 1000 queries) and the decoding of call i-1 pass under the kernels of call i.  A submit that finds the
 * GPU busy with the call before it builds its tables and stops there; the wait that sees that earlier call finish
 * enqueues it before decoding (FUZZYHIP_SUBMIT_EAGER=1: enqueue at once, behind the running kernels -- no host turn
 * between calls, but the HIP runtime of the PyTorch wheels stalls in such launches).
 *   - at most FZ_MAX_IN_FLIGHT tickets per corpus (a further submit fails with FZ_ERR_LIMIT); every ticket must be
 *     waited for exactly once (by one thread), in any order; fz_corpus_destroy waits for the GPU but fetches nothing
 *   - q_cps / q_offs must stay valid and unchanged until the wait: rows whose fused lists overflow are redone from them
 *   - calls of one corpus run on its one stream in the order they were submitted; other entry points may be called
 *     in between (they queue up behind); fz_set_stream must not be called while a ticket is outstanding
 *   - fz_last_timing after a wait describes the call most recently SUBMITTED, once it has finished
 */
int fz_query_topk_submit(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                         int scorer, int32_t min_score, uint32_t k, int *ticket);
int fz_query_topk_wait(fz_corpus *c, int ticket, uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt);

/*
 * fz_query_topk_noalloc -- fz_query_topk for callers that cannot allocate device memory while they run:
 * tight serving loops, harnesses that forbid hipMalloc, processes whose device memory is spoken for.  (Not for a
 * stream under graph capture: like fz_query_topk the call synchronises its stream, copies results to the host and
 * decodes them there; it makes no device or pinned allocation, that is all it promises.)  It has NO
 * counterpart in the reference (whose loop allocates Python objects per candidate, src/fuzzy_search.py:22-32).
 * Same arguments, same results bit for bit; it calls no hipMalloc / hipFree / hipHostMalloc / hipEventCreate
 * and no other allocating HIP routine.  Its device scratch -- work-queue counters, query tables, the score
 * matrix of a chunk, result keys and counts -- is carved out of ONE static __device__ array of
 * FZ_NOALLOC_SCRATCH_BYTES (32 MiB) per GPU; pinned host staging, events and the scorer's device view were
 * set up when the corpus was created (a scorer other than FZ_TOKEN_SET needs one earlier fz_query_topk call
 * with that scorer, which builds its view).  Limits, each refused with FZ_ERR_LIMIT and a message naming
 * it: query tables of one chunk <= FZ_NOALLOC_BLOB_BYTES, nq * k * 8 <= FZ_NOALLOC_KEYS_BYTES, nq <= 32768,
 * one score row (the padded candidate count, bytes) <= the ~30 MiB the other parts leave; a batch whose
 * score matrix is larger goes through the array in chunks of rows.
 * Every call on a GPU shares that GPU's array: the library serialises the calls of ONE process per GPU,
 * calls from different processes on one GPU use different copies (one per process), and nothing read from
 * the array survives the call.  What must start at zero (the work-queue counters) is cleared by the call.
 */
int fz_query_topk_noalloc(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                          int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx, int32_t *out_score,
                          uint32_t *out_cnt);

/*
 * fz_query_topk_keys -- same work, results as packed sort keys for the multi-GPU exchange
 * (SURVEY.md 8e): key = (score << 40) | ((2^40-1) - global_index); larger key = better match;
 * rows padded with 0.  `out_keys_device` is a DEVICE pointer to nq*k u64 (the buffer the RCCL
 * all-gather sends); `out_cnt_device` a DEVICE pointer to nq u32 or NULL.  Asynchronous on the
 * corpus' stream when `sync` is 0.  No counterpart in the reference (it has no multi-GPU path).
 */
int fz_query_topk_keys(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                       int scorer, int32_t min_score, uint32_t k, uint64_t *out_keys_device,
                       uint32_t *out_cnt_device, int sync);

/*
 * fz_query_topk_pair -- the scoring loop of the reference's WEIGHTED FuzzySearch (app.py:995-1002,
 * 1014-1032): score = partial_ratio/100*0.5 + token_set_ratio/100*0.5, capped at 0.99, kept if
 * >= threshold, stable sort descending, slice top_n.  The two scorers see different forms of the
 * same candidate (partial_ratio the normalised string, token_set_ratio its full_process-ed form),
 * hence two corpora `ca`, `cb` over the same list (same n and base index, same device) and two
 * packed query lists.  The fp64 expression is evaluated by the HOST LANGUAGE for all 101 x 101
 * integer score pairs and passed as ranks: rank_tab[a*101 + b] = rank of the value among the
 * distinct values (larger = better), min_rank = smallest rank whose value passes the threshold.
 *   out_idx  : nq*k global indices, row q ordered by (rank desc, index asc)
 *   out_rank : nq*k ranks (the caller maps a rank back to its fp64 value)
 */
int fz_query_topk_pair(fz_corpus *ca, int scorer_a, const uint32_t *qa_cps, const uint64_t *qa_offs,
                       fz_corpus *cb, int scorer_b, const uint32_t *qb_cps, const uint64_t *qb_offs,
                       uint32_t nq, const uint16_t *rank_tab, uint32_t min_rank, uint32_t k,
                       uint64_t *out_idx, int32_t *out_rank, uint32_t *out_cnt);

/*
 * fz_merge_topk -- deterministic merge of n_lists per-shard key lists (each nq*k keys as written
 * by fz_query_topk_keys, gathered in any order) into the global top-k: descending key, i.e.
 * (score desc, global index asc).  Host code; it is the tail of find_best_matches' sort + slice
 * (src/fuzzy_search.py:35-38) for a sharded corpus.
 */
int fz_merge_topk(const uint64_t *keys, uint32_t n_lists, uint32_t nq, uint32_t k,
                  uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt);

/*
 * The exchange inside the library ("one ncclAllGather per query batch"): one process per
 * GPU, each with its shard in an fz_corpus (contiguous index ranges, `base_index`), and ONE collective per
 * batch: the all-gather of every rank's nq*k packed keys over RCCL (xGMI), issued on the corpus' stream
 * right behind the top-k kernels, followed by a device merge of the n_ranks lists of every query
 * (descending key = score desc, global index asc: identical on every rank and for every n_ranks).  No
 * counterpart in the reference (single process).  librccl is opened with dlopen at the first fz_comm_*
 * call, so single-GPU users never load it: FUZZYHIP_RCCL_LIB if set, else the librccl that sits NEXT TO the
 * libamdhip64 this library is bound to (so that RCCL and the library call the same HIP runtime), else the
 * loader's default search.
 * The local top-k is only enqueued: no host synchronisation stands between it and the all-gather.  Each rank
 * sends one status word behind its keys; a rank whose local step fails still enters the collective (zeroed
 * keys, status 1) and every rank then returns an error instead of waiting for ever; a fused list that
 * overflowed (status 2) makes every rank go round once more with that rank's keys redone.  Exchange buffers for
 * 131 072 keys per rank are reserved by fz_comm_create; larger batches grow them on first use.
 *
 *   fz_comm_unique_id  rank 0 only: FZ_COMM_ID_BYTES bytes to hand to every rank by whatever launched the
 *                      processes (torch.distributed broadcast, a file, MPI, ...)
 *   fz_comm_create     every rank, collectively: joins the communicator on `device`
 *   fz_query_topk_sharded  every rank, collectively, with the SAME queries: fz_query_topk over the whole
 *                      sharded list; every rank receives the global result.  n_ranks * k <= FZ_COMM_MAX_MERGE
 */
typedef struct fz_comm fz_comm;
#define FZ_COMM_ID_BYTES 128
#define FZ_COMM_MAX_MERGE 8192u
int fz_comm_unique_id(void *id_out);
int fz_comm_create(const void *id, int n_ranks, int rank, int device, fz_comm **out);
int fz_comm_destroy(fz_comm *comm);
int fz_query_topk_sharded(fz_corpus *c, fz_comm *comm, const uint32_t *q_cps, const uint64_t *q_offs,
                          uint32_t nq, int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                          int32_t *out_score, uint32_t *out_cnt);

/*
 * fz_score_all -- every integer score of nq queries against every candidate, in ORIGINAL
 * candidate order (out_scores[q*n + i], host memory).  This is `similarity_score` of
 * src/fuzzy_search.py:26 for all i; used by the parity tests.
 */
int fz_score_all(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                 int scorer, uint8_t *out_scores);

/*
 * fz_distance_all -- raw distances of nq queries against every candidate AS GIVEN (no token
 * canonicalisation), ORIGINAL candidate order: out_dist[q*n + i], host memory, u16.
 *   kind 0: InDel distance  
 *   kind 1: uniform-cost Levenshtein by the Myers/Hyyro Pv/Mv bit-vector recurrence; this is
 *           AdvancedFuzzyMatcher.levenshtein_distance of reference app.py:296-317 once the host has
 *           applied its normalize_text (app.py:271-273) to both sides
 *   kind 2: number of the candidate's distinct whitespace-separated tokens that the query also has.
 *           With each string replaced by its distinct character n-grams this is the |A & B| of
 *           AdvancedFuzzyMatcher.jaccard_similarity (app.py:282-293)
 *   kind 3: 1 if the query occurs in the candidate as a contiguous substring, else 0 (bit-parallel
 *           Shift-And): `query_norm in target_norm` of substring_match_score (app.py:337-345) and,
 *           with a sentinel in front of both sides, the startswith tests of prefix_match_score
 *           (app.py:320-334)
 *   kind 4: the number M of matching characters difflib.SequenceMatcher(None, query, candidate) finds
 *           (Ratcliff/Obershelp: longest common block, earliest in the query, then earliest in the candidate, then the
 *           same on both sides of it), so that SequenceMatcher.ratio() = 2.0 * M / (|query| + |candidate|)
 *           (reference app.py:421-424).  Queries of at most 64 code points (FZ_ERR_TOO_LONG beyond); a candidate
 *           longer than 64 code points gets 0xFFFF ("ask difflib": its junk heuristic starts at 200)
 */
int fz_distance_all(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                    int kind, uint16_t *out_dist);

/*
 * fz_distance_batch -- raw distances of n_pairs explicit (a_i, b_i) string pairs, computed by the
 * same bit-parallel device routines the scorers use.
 *   kind 0: InDel distance |a|+|b|-2*LCS (what thefuzz's ratios normalise)
 *   kind 1: uniform-cost Levenshtein (reference app.py:296-317)
 */
int fz_distance_batch(int device, int kind, const uint32_t *a_cps, const uint64_t *a_offs,
                      const uint32_t *b_cps, const uint64_t *b_offs, uint64_t n_pairs,
                      uint32_t *out_dist);

/* kernel timings of the last query call on this corpus (hipEvents on its stream) */
int fz_last_timing(const fz_corpus *c, fz_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* FUZZYHIP_H */
