// fz_host.cpp -- host side of libfuzzyhip.so: corpus ingest, query preparation, launches, C ABI.
//
// Reference path: FuzzySearch.__init__ / find_best_matches (reference src/fuzzy_search.py:10-38).
// Nothing here scores on the CPU: without a gfx950 device every scoring entry point fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fuzzyhip.h"
#include "fz_common.h"

// from fz_kernels.hip
int fz_pick_width(uint32_t m);
size_t fz_score_lds_bytes(int W, const FzScoreParams &p);
hipError_t fz_launch_score(int W, int mode, const FzScoreParams &p, uint32_t nq, hipStream_t st);
hipError_t fz_launch_select(const FzSelectParams &p, uint32_t rows, hipStream_t st);
size_t fz_stream_lds_bytes(int W, int stride);
hipError_t fz_launch_stream(int W, int mode, const FzStreamParams &p, hipStream_t st);
hipError_t fz_launch_stream_pk(int mode, const FzStreamParams &p, hipStream_t st);
hipError_t fz_launch_combine(const FzCombineParams &p, uint32_t rows, hipStream_t st);
hipError_t fz_launch_select16(const FzSelect16Params &p, uint32_t rows, hipStream_t st);
hipError_t fz_launch_merge(const FzMergeParams &p, uint32_t rows, hipStream_t st);
hipError_t fz_launch_merge_gathered(const uint64_t *all, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t *out, hipStream_t st);
hipError_t fz_launch_unpermute(const void *in, uint64_t in_stride, void *out, uint64_t out_stride, const uint32_t *phys_of_orig,
                               uint32_t n, uint32_t rows, uint32_t elem_bytes, hipStream_t st);
hipError_t fz_scratch_acquire(int device, void **ptr, size_t *bytes);

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(FZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// No C++ exception may cross the C ABI (a std::bad_alloc from a corrupt size field or plain memory
// exhaustion would abort the host process): every extern "C" body runs inside guarded().
template <typename F>
static int guarded(F &&f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(FZ_ERR_LIMIT, "out of host memory");
    } catch (const std::length_error &e) {
        return fail(FZ_ERR_LIMIT, "size beyond what the host can allocate: %s", e.what());
    } catch (const std::exception &e) {
        return fail(FZ_ERR_INVALID, "unexpected failure: %s", e.what());
    } catch (...) {
        return fail(FZ_ERR_INVALID, "unexpected failure");
    }
}

// ---------------------------------------------------------------------------------------------
// tokens
// ---------------------------------------------------------------------------------------------
// the characters Python's str.split() (and rapidfuzz's tokeniser) split on
static inline bool is_space(uint32_t c)
{
    if (c == 0x20u) return true;
    if (c < 0x85u) return (c >= 0x09u && c <= 0x0Du) || (c >= 0x1Cu && c <= 0x1Fu);
    return c == 0x85u || c == 0xA0u || c == 0x1680u || (c >= 0x2000u && c <= 0x200Au) ||
           c == 0x2028u || c == 0x2029u || c == 0x202Fu || c == 0x205Fu || c == 0x3000u;
}

using Tok = std::u32string_view;

// whitespace split, sort by code point, optionally drop repeats: sorted(set(s.split()))
static void split_sorted(const uint32_t *s, uint64_t n, bool dedupe, std::vector<Tok> &out)
{
    out.clear();
    const char32_t *p = reinterpret_cast<const char32_t *>(s);
    uint64_t i = 0;
    while (i < n) {
        while (i < n && is_space(s[i])) ++i;
        if (i >= n) break;
        uint64_t st = i;
        while (i < n && !is_space(s[i])) ++i;
        out.emplace_back(p + st, (size_t)(i - st));
    }
    std::sort(out.begin(), out.end());          // u32string_view compares by code point
    if (dedupe) out.erase(std::unique(out.begin(), out.end()), out.end());
}

struct TokHash {
    size_t operator()(Tok t) const noexcept
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)t.size();
        for (char32_t ch : t) {
            h ^= (uint64_t)ch;
            h *= 0x100000001B3ull;
            h ^= h >> 29;
        }
        return (size_t)h;
    }
};

// ---------------------------------------------------------------------------------------------
// corpus = host dictionaries + up to three device "views" of the same candidate list
// ---------------------------------------------------------------------------------------------
enum { VIEW_SET = 0,    // sorted unique tokens, each followed by a separator  (token_set_ratio)
       VIEW_SORT = 1,   // sorted tokens joined by single spaces, repeats kept   (token_sort_ratio)
       VIEW_RAW = 2,    // the string as given                                   (ratio, distances)
       N_VIEWS = 3 };

struct View {
    bool built = false;
    bool wide = false;                                    // > 254 distinct code points: 16-bit symbols
    uint32_t npad = 0, ntiles = 0, max_ndw = 0, max_len = 0, max_tok = 0, alphabet = 2;
    uint64_t total_symbols = 0, total_tokens = 0, device_bytes = 0;
    uint64_t n_sym_words = 0, n_tok_slots = 0;            // lengths of d_sym and d_tokid / d_tokend
    std::vector<uint16_t> sym_of_cp;                      // code point -> symbol (0 = not in corpus)
    uint32_t *d_sym = nullptr, *d_tokid = nullptr, *d_orig_of_phys = nullptr, *d_phys_of_orig = nullptr;
    uint32_t *d_sample_walk = nullptr;                    // original indices of the sample candidates, ascending
    uint16_t *d_tokend = nullptr;
    FzTile *d_tiles = nullptr;
    // tile indices by kernel: stream kernel with 1 / 2 pattern words, general kernel for the rest
    // (s0: candidates of at most 15 symbols, taken two tiles at a time by the packed stream kernel)
    uint32_t *d_tiles_s0 = nullptr, *d_tiles_s1 = nullptr, *d_tiles_s2 = nullptr, *d_tiles_slow = nullptr;
    uint32_t n_s0 = 0, n_s1 = 0, n_s2 = 0, n_slow = 0;
    // Sample tiles (every FZ sample_every-th tile of the length-sorted order) hold the physical positions
    // [0, n_sample_tiles * 64); each tile list holds its sample entries first (ns_*).  n_sample_tiles ==
    // ntiles: no split (small shards), every top-k call goes through the score matrix.
    uint32_t n_sample_tiles = 0, ns_s0 = 0, ns_s1 = 0, ns_s2 = 0, ns_slow = 0;
    std::vector<uint32_t> h_orig_of_phys;
};

// Device memory a query call works in beyond its inputs and outputs.  The ordinary entry points own
// one per corpus and grow it with hipMalloc as batches get larger; fz_query_topk_noalloc carves one out
// of the library's static __device__ scratch array (fz_kernels.hip) and never grows it.
struct Workspace {
    uint8_t *d_scores = nullptr; size_t scores_cap = 0;   // u8 / u16 score matrix [rows][row_stride]
    uint8_t *d_qbuf = nullptr; size_t qbuf_cap = 0;       // query blob: Peq tables, token triples, stream records
    uint64_t *d_keys = nullptr; size_t keys_cap = 0;      // top-k keys [nq][k]
    uint32_t *d_cnt = nullptr; size_t cnt_cap = 0;        // result counts [nq]
    uint32_t *d_counters = nullptr;                       // stream kernel work-queue counters (FZ_COUNTER_SLOTS launches x FZ_STREAM_QUEUES)
    bool fixed = false;                                   // true: carved from the static scratch, a larger need is an error
};
#define FZ_COUNTER_SLOTS 64u
#define FZ_COUNTERS_AHEAD 8u      // slots zeroed by the one fill at a call's first stream launch (a fused call of one chunk uses six)

struct Prepared;

struct fz_corpus {
    int device = 0;
    uint64_t base = 0, n = 0;
    std::vector<uint32_t> h_cps;                          // the input, kept for views built on demand
    std::vector<uint64_t> h_offs;
    std::deque<std::u32string> vocab_store;               // stable storage of the token strings
    std::unordered_map<Tok, uint32_t, TokHash> vocab;     // token -> id (VIEW_SET)
    View views[N_VIEWS];

    // query-time state
    hipStream_t stream = nullptr, own_stream = nullptr;
    hipStream_t aux_stream[2] = {nullptr, nullptr};        // the stream kernels of one call run side by side (FUZZYHIP_NO_OVERLAP=1: one after the other)
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool overlap = true;
    uint64_t ws_limit = 4ull << 30;
    Workspace own_ws;                                     // grown on demand by the ordinary entry points
    Workspace *ws = &own_ws;                              // fz_query_topk_noalloc points it at the static scratch for one call
    uint8_t *d_score_tab[2] = {nullptr, nullptr};         // stream kernel: T[L][d] per rounding form
    uint32_t counter_next = 0, counters_clean = 0;        // work-queue counter slots of the current call: next to hand out, zeroed ahead
    uint64_t call_seq = 0, counters_call = ~0ull;          // calls run so far; the call the counters were last zeroed for
    // pinned host staging: the query blob of a chunk (two buffers, so the host fills one while the copy
    // engine still reads the other) and the keys read back at the end of a call
    uint8_t *h_blob[2] = {nullptr, nullptr}; size_t h_blob_cap[2] = {0, 0};
    hipEvent_t ev_blob[2] = {nullptr, nullptr};
    uint32_t blob_next = 0;
    uint64_t *h_keys = nullptr; size_t h_keys_cap = 0;
    uint32_t n_cus = 256;
    uint16_t *d_ranks = nullptr; size_t ranks_cap = 0;    // fz_query_topk_pair: combined ranks, rank table
    uint16_t *d_rank_tab = nullptr; size_t rank_tab_cap = 0;
    bool use_stream = true;                               // FUZZYHIP_NO_STREAM=1 forces the general kernel
    bool use_fused = true;                                // FUZZYHIP_NO_FUSED=1: every top-k call through the score matrix
    uint32_t prep_threads = 4;                            // host threads that build the query tables of a large chunk (FUZZYHIP_PREP_THREADS)
    uint64_t fused_min_bytes = 64ull << 20;               // FUZZYHIP_FUSED_MIN_MB: smaller score matrices are simply written (fewer launches)
    uint64_t slab_bytes = 24ull << 20;                    // FUZZYHIP_SLAB_KB: corpus bytes one stream kernel keeps cache resident
    uint32_t stream_oversub = 1;                          // FUZZYHIP_STREAM_OVERSUB: persistent blocks per slot
    uint32_t pipe_chunks = 1;                             // FUZZYHIP_CHUNKS: pieces of a large top-k batch
    bool submit_eager = false;                            // FUZZYHIP_SUBMIT_EAGER=1: fz_query_topk_submit enqueues behind a running call
    uint32_t stream_waves = 0, stream_blocks = 192;       // tuning knobs (FUZZYHIP_STREAM_WAVES: waves per block, 0 = by kernel; _BLOCKS)
    // sliced select (FzSelectParams::n_slices): a few rows over a large shard are selected by several blocks per row
    uint64_t *d_slice_keys = nullptr; uint32_t *d_slice_done = nullptr;
    uint32_t select_slice = 65536, select_slice_rows = 8;  // FUZZYHIP_SELECT_SLICE (positions per block, 0: off), rows up to which rows are sliced
    uint32_t stream_blocks_sample = 48;                   // ... of the sample phase of a fused call (FUZZYHIP_STREAM_BLOCKS_SAMPLE)
    uint32_t stream_min_queries = 8;                      // FUZZYHIP_STREAM_MIN_QUERIES: smaller batches take the general kernel for every tile
    uint32_t stream_max_group = 1024;                     // most queries one work item takes (FUZZYHIP_STREAM_MAX_GROUP); its waves share them out
    std::vector<hipEvent_t> events;
    fz_timing timing{};
    // events of an asynchronous call whose kernel times are read once its caller has synchronised (fz_query_topk_sharded)
    struct Pending {
        bool valid = false;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> score, select;
        hipEvent_t first = nullptr, last = nullptr;
    } pending;
    // an asynchronous fused call (fz_query_topk_sharded) leaves its overflow records in the workspace: which rows must be
    // redone is only known once the caller has synchronised; redo_async_overflow() then repairs those rows
    struct AsyncFused {
        bool valid = false;
        uint32_t *d_ovf = nullptr, *d_ovf_rows = nullptr;
        uint32_t nchunks = 0; uint64_t rows_per_chunk = 0;
        int view_kind = 0, kmode = 0; FzTies ties{0, 0};
        uint64_t *d_keys = nullptr; uint32_t *d_cnt = nullptr; int32_t min_score = 0; uint32_t k = 0, flat = 0;
    } async_fused;
    // query tables built ahead by a fz_query_topk_submit that found the GPU busy (prepare_rows takes them when the call is launched)
    struct Prebuilt {
        bool valid = false;
        const uint32_t *q_cps = nullptr; const uint64_t *q_offs = nullptr; uint64_t q0 = 0; uint32_t rows = 0; int view_kind = 0, kmode = 0;
        std::shared_ptr<Prepared> P;
    } prebuilt;
    // fz_query_topk_submit / fz_query_topk_wait: calls submitted whose result has not been fetched.  `launched` is false
    // while a call only has its tables built (the GPU was busy with the call before it)
    struct Flight {
        std::atomic<bool> busy{false};                      // set by submit, cleared by wait (also on its error paths, outside the lock)
        bool empty = false, fused = false, launched = false;
        int scorer = 0;
        uint64_t seq = 0;
        const uint32_t *q_cps = nullptr; const uint64_t *q_offs = nullptr; uint32_t nq = 0, k = 0, flat = 0;
        int view_kind = 0, kmode = 0; FzTies ties{0, 0}; int32_t min_score = 0;
        uint32_t nchunks = 0; uint64_t rows_per_chunk = 0;
        uint64_t *d_keys = nullptr; size_t d_words = 0;     // nq*k keys + the status word
        uint64_t *h_keys = nullptr; size_t h_words = 0;     // pinned copy of the same
        uint32_t *h_ovf = nullptr;                          // pinned [FZ_FUSED_MAX_CHUNKS | 1 + FZ_FUSED_REDO_ROWS]: the call's overflow records
        hipEvent_t done = nullptr;
    } flight[FZ_MAX_IN_FLIGHT];
    uint64_t flight_seq = 0;
    std::mutex mu;
};

static FzTies g_ties_norm_distance, g_ties_norm_sim;
static std::once_flag g_ties_once;

// The fp64 value of thefuzz's score is only ever ambiguous at exact ties q + 0.5 (fz_kernels.hip
// fz_round_ratio).  A tie means 200*d/L == 199 - 2q, so evaluate each formula at d = 199 - 2q,
// L = 200 (same quotient, and the fp64 result depends on the quotient only) and record whether
// int(round()) -- round-half-even on the computed double -- lands on q + 1.
static void make_tie_tables()
{
    for (int form = 0; form < 2; ++form) {
        FzTies t{0, 0};
        for (int q = 0; q < 100; ++q) {
            volatile double d = 199.0 - 2.0 * q, L = 200.0, x;
            if (form == 0) {          // rapidfuzz norm_distance<100>: 100 - 100*d/L   (row a6)
                volatile double num = 100.0 * d;
                volatile double quo = num / L;
                x = 100.0 - quo;
            } else {                  // rapidfuzz ratio: (1 - d/L) * 100              (row a8)
                volatile double nd = d / L;
                volatile double ns = 1.0 - nd;
                x = ns * 100.0;
            }
            long r = std::lrint((double)x);   // FE_TONEAREST: ties to even, as Python round()
            if (r == q + 1) (q < 64 ? t.lo : t.hi) |= 1ull << (q & 63);
        }
        (form == 0 ? g_ties_norm_distance : g_ties_norm_sim) = t;
    }
}

// exact integer form of the rounded ratio (the device twin is fz_round_ratio in fz_kernels.hip)
static uint32_t round_ratio_host(uint32_t d, uint32_t L, const FzTies &t)
{
    const uint32_t N = 100u * (L - d);
    uint32_t q = N / L;
    const uint32_t twice = 2u * (N % L);
    if (twice > L) q += 1;
    else if (twice == L) q += (uint32_t)(((q < 64 ? t.lo >> q : t.hi >> (q - 64))) & 1ull);
    return q;
}

// stream kernel score table: T[L * 64 + (d >> 1)] = rounded ratio of distance d at length sum L
// (d <= L < 128, d of L's parity); T[0][0] = 100 is thefuzz's score of two empty strings under ratio().
static std::vector<uint8_t> make_score_table(const FzTies &t)
{
    std::vector<uint8_t> tab((size_t)FZ_STREAM_TAB_BYTES, 0);
    tab[0] = 100;
    for (uint32_t L = 1; L < FZ_STREAM_TAB; ++L)
        for (uint32_t d = L & 1u; d <= L; d += 2u)          // an InDel distance has the parity of the length sum
            tab[(size_t)L * (FZ_STREAM_TAB / 2u) + (d >> 1)] = (uint8_t)round_ratio_host(d, L, t);
    return tab;
}

template <typename T>
static int upload(T **dptr, const std::vector<T> &h, uint64_t *bytes)
{
    size_t sz = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), sz));
    if (!h.empty()) HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += sz;
    return FZ_OK;
}

extern "C" int fz_version(void) { return FZ_VERSION; }

extern "C" const char *fz_last_error(void) { return g_err.c_str(); }

extern "C" int fz_device_count(int *out_n)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (out_n) *out_n = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n == 0) return fail(FZ_ERR_NO_DEVICE, "no HIP device visible");
    return FZ_OK;
}

static void free_view(View &v)
{
    for (void *p : {(void *)v.d_sym, (void *)v.d_tokid, (void *)v.d_tokend, (void *)v.d_tiles,
                    (void *)v.d_orig_of_phys, (void *)v.d_phys_of_orig, (void *)v.d_sample_walk, (void *)v.d_tiles_s0, (void *)v.d_tiles_s1,
                    (void *)v.d_tiles_s2, (void *)v.d_tiles_slow})
        if (p) (void)hipFree(p);
    v = View();
}

extern "C" void fz_corpus_destroy(fz_corpus *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (View &v : c->views) free_view(v);
    Workspace &w = c->own_ws;
    for (void *p : {(void *)w.d_scores, (void *)w.d_qbuf, (void *)w.d_keys, (void *)w.d_cnt,
                    (void *)c->d_score_tab[0], (void *)c->d_score_tab[1], (void *)c->d_ranks, (void *)c->d_rank_tab, (void *)w.d_counters,
                    (void *)c->d_slice_keys, (void *)c->d_slice_done})
        if (p) (void)hipFree(p);
    for (fz_corpus::Flight &f : c->flight) {
        if (f.d_keys) (void)hipFree(f.d_keys);
        if (f.h_keys) (void)hipHostFree(f.h_keys);
        if (f.h_ovf) (void)hipHostFree(f.h_ovf);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->h_blob[i]) (void)hipHostFree(c->h_blob[i]);
        if (c->ev_blob[i]) (void)hipEventDestroy(c->ev_blob[i]);
    }
    if (c->h_keys) (void)hipHostFree(c->h_keys);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (hipStream_t a : c->aux_stream) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
    for (hipEvent_t e : {c->ev_fork, c->ev_join[0], c->ev_join[1]}) if (e) (void)hipEventDestroy(e);
    delete c;
}

// tile indices by kernel: stream kernel with one / two pattern words, general kernel for the rest
static int upload_tile_lists(View &v, const std::vector<FzTile> &tiles, const std::vector<uint32_t> &tile_maxlen)
{
    std::vector<uint32_t> ts0, ts1, ts2, tslow;
    const bool pack = getenv("FUZZYHIP_NO_PACK") == nullptr;      // experiments: keep short tiles in the plain kernel
    if (v.n_sample_tiles == 0 || v.n_sample_tiles > v.ntiles) v.n_sample_tiles = v.ntiles;
    // the sample tiles [0, n_sample_tiles) and the others are listed (and paired) separately, sample first
    for (int part = 0; part < 2; ++part) {
        const uint32_t t_lo = part == 0 ? 0u : v.n_sample_tiles, t_hi = part == 0 ? v.n_sample_tiles : v.ntiles;
        // Packed pairs: two tiles whose longest candidates have 31 symbols or fewer TOGETHER share one
        // pattern word per lane (tile A below the split, tile B above it).  Tiles are sorted by their longest
        // candidate and the shortest is paired with the longest that still fits; a tile that finds no
        // partner stays with the one-tile kernel.
        std::vector<uint8_t> paired(v.ntiles, 0);
        if (pack && !v.wide && v.ntiles < (1u << FZ_PK_SPLIT_SHIFT)) {
            std::vector<uint32_t> el;
            for (uint32_t t = t_lo; t < t_hi; ++t)
                if (tiles[t].ntok <= FZ_PK_MAX_TOK && tile_maxlen[t] <= 29u) el.push_back(t);
            std::stable_sort(el.begin(), el.end(), [&](uint32_t a, uint32_t b) { return tile_maxlen[a] < tile_maxlen[b]; });
            size_t lo = 0, hi = el.size();
            while (lo + 1 < hi) {
                const uint32_t a = el[lo], b = el[hi - 1];
                if (tile_maxlen[a] + tile_maxlen[b] <= 31u) {
                    ts0.push_back(a | (tile_maxlen[a] << FZ_PK_SPLIT_SHIFT));      // split = field width of tile A
                    ts0.push_back(b);
                    paired[a] = paired[b] = 1;
                    ++lo; --hi;
                } else {
                    --hi;                                                         // too long for any partner left
                }
            }
        }
        for (uint32_t t = t_lo; t < t_hi; ++t) {
            // the stream kernel keeps one Peq row per symbol byte: 8-bit symbols only
            if (paired[t]) continue;
            if (!v.wide && tiles[t].ntok <= FZ_STREAM_MAX_TOK && tiles[t].ndw <= 8u) ts1.push_back(t);
            else if (!v.wide && tiles[t].ntok <= FZ_STREAM_MAX_TOK && tiles[t].ndw <= 16u) ts2.push_back(t);
            else tslow.push_back(t);
        }
        if (part == 0) {
            v.ns_s0 = (uint32_t)ts0.size(); v.ns_s1 = (uint32_t)ts1.size();
            v.ns_s2 = (uint32_t)ts2.size(); v.ns_slow = (uint32_t)tslow.size();
        }
    }
    v.n_s0 = (uint32_t)ts0.size();
    v.n_s1 = (uint32_t)ts1.size(); v.n_s2 = (uint32_t)ts2.size(); v.n_slow = (uint32_t)tslow.size();
    if (getenv("FUZZYHIP_TRACE"))
        fprintf(stderr, "fuzzyhip: %u tiles (%u sample) -> packed %u, stream1 %u, stream2 %u, general %u\n",
                v.ntiles, v.n_sample_tiles, v.n_s0, v.n_s1, v.n_s2, v.n_slow);
    int rc;
    if (v.n_sample_tiles < v.ntiles) {
        std::vector<uint32_t> walk(v.h_orig_of_phys.begin(), v.h_orig_of_phys.begin() + (size_t)v.n_sample_tiles * FZ_TILE);
        std::sort(walk.begin(), walk.end());
        if ((rc = upload(&v.d_sample_walk, walk, &v.device_bytes)) != FZ_OK) return rc;
    }
    if ((rc = upload(&v.d_tiles_s0, ts0, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_tiles_s1, ts1, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_tiles_s2, ts2, &v.device_bytes)) != FZ_OK) return rc;
    return upload(&v.d_tiles_slow, tslow, &v.device_bytes);
}

// Builds one device view: reduce every candidate to the text its scorer compares, map code points
// to 8-bit symbols, sort by (length, tokens, original index), cut into 64-candidate tiles,
// transpose to symbol-major order and upload.
static int build_view(fz_corpus *c, int kind)
{
    View &v = c->views[kind];
    if (v.built) return FZ_OK;
    const uint64_t n = c->n;
    const uint32_t *cps = c->h_cps.data();
    const uint64_t *offs = c->h_offs.data();

    // ---- alphabet ---------------------------------------------------------------------------
    {
        std::vector<uint8_t> seen(0x110000, 0);
        const uint64_t total = n ? offs[n] : 0;
        for (uint64_t i = n ? offs[0] : 0; i < total; ++i) seen[cps[i]] = 1;
        v.sym_of_cp.assign(0x110000, 0);
        uint32_t next = 2;
        if (kind == VIEW_SORT) v.sym_of_cp[0x20] = (uint16_t)FZ_SYM_SPACE;
        for (uint32_t cp = 0; cp < 0x110000u; ++cp) {
            if (!seen[cp]) continue;
            if (kind != VIEW_RAW && is_space(cp)) continue;       // separators never reach the text
            if (next > 0xFFFFu) return fail(FZ_ERR_ALPHABET, "corpus has more than 65534 distinct code points; the 16-bit symbol table is full");
            v.sym_of_cp[cp] = (uint16_t)next++;
        }
        v.alphabet = next;
        v.wide = next > 256u;                                     // 8-bit symbols no longer suffice
    }
    const uint32_t spd = v.wide ? 2u : 4u;                        // symbols per dword

    // ---- text of every candidate ------------------------------------------------------------
    std::vector<uint16_t> canon;         // symbols, all candidates back to back
    std::vector<uint64_t> canon_off(n + 1, 0);
    std::vector<uint32_t> tok_id;        // VIEW_SET: token ids, all candidates back to back
    std::vector<uint16_t> tok_end;       // VIEW_SET: exclusive end (incl. separator) of each token
    std::vector<uint64_t> tok_off(n + 1, 0);
    {
        std::vector<Tok> toks;
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t *s = cps + offs[i];
            const uint64_t sl = offs[i + 1] - offs[i];
            uint64_t len = 0;
            if (kind == VIEW_RAW) {
                for (uint64_t j = 0; j < sl; ++j) canon.push_back(v.sym_of_cp[s[j]]);
                len = sl;
                v.total_symbols += len;
            } else {
                split_sorted(s, sl, kind == VIEW_SET, toks);
                for (size_t ti = 0; ti < toks.size(); ++ti) {
                    Tok t = toks[ti];
                    if (kind == VIEW_SORT && ti) { canon.push_back((uint16_t)FZ_SYM_SPACE); ++len; }
                    for (char32_t ch : t) canon.push_back(v.sym_of_cp[(uint32_t)ch]);
                    len += t.size();
                    if (kind == VIEW_SET) {
                        auto it = c->vocab.find(t);
                        uint32_t id;
                        if (it == c->vocab.end()) {
                            if (c->vocab.size() >= 0xFFFFFFF0ull) return fail(FZ_ERR_LIMIT, "vocabulary exceeds 2^32 tokens");
                            c->vocab_store.emplace_back(t);
                            id = (uint32_t)c->vocab.size();
                            c->vocab.emplace(Tok(c->vocab_store.back()), id);
                        } else {
                            id = it->second;
                        }
                        canon.push_back((uint16_t)FZ_SYM_SPACE);
                        len += 1;
                        tok_id.push_back(id);
                        tok_end.push_back((uint16_t)std::min<uint64_t>(len, 0xFFFF));
                    }
                }
                v.total_symbols += (kind == VIEW_SET && len) ? len - 1 : len;
                v.total_tokens += toks.size();
                v.max_tok = std::max<uint32_t>(v.max_tok, (uint32_t)toks.size());
            }
            if (len > FZ_MAX_CAND_LEN) return fail(FZ_ERR_TOO_LONG, "candidate %llu is longer than %u code points", (unsigned long long)i, FZ_MAX_CAND_LEN);
            if (kind != VIEW_SET) { tok_id.push_back(FZ_TOK_NONE); tok_end.push_back((uint16_t)len); }   // one slot: the length
            canon_off[i + 1] = canon.size();
            tok_off[i + 1] = tok_id.size();
            v.max_len = std::max<uint32_t>(v.max_len, (uint32_t)len);
        }
    }

    // ---- physical order and tiles -----------------------------------------------------------
    v.ntiles = (uint32_t)((n + FZ_TILE - 1) / FZ_TILE);
    v.npad = v.ntiles * FZ_TILE;
    std::vector<uint64_t> keys(n);
    for (uint64_t i = 0; i < n; ++i) {
        // by exact length (so that a tile's longest candidate is close to its shortest: the packed
        // kernel pairs tiles by that length), then token slots, then original index
        uint64_t len = canon_off[i + 1] - canon_off[i], nt = tok_off[i + 1] - tok_off[i];
        keys[i] = (len << 48) | (std::min<uint64_t>(nt, 0xFFFF) << 32) | i;
    }
    std::sort(keys.begin(), keys.end());
    // Large shards keep a SAMPLE in front of the physical order: every `every`-th tile of the sorted order
    // (so the sample has the length distribution of the shard).  A fused top-k call scores it first; its
    // k-th best key is the limit the other tiles' candidates must reach (FzEmit, fz_common.h).  The last
    // tile may be partly padding and stays last.
    v.n_sample_tiles = v.ntiles;
    {
        uint32_t every = 16, min_tiles = 256;
        if (const char *e = getenv("FUZZYHIP_SAMPLE_EVERY")) every = (uint32_t)std::max(0, atoi(e));
        if (const char *e = getenv("FUZZYHIP_SAMPLE_MIN_TILES")) min_tiles = (uint32_t)std::max(2, atoi(e));
        if (every >= 2 && v.ntiles >= min_tiles) {
            std::vector<uint64_t> front, back;
            front.reserve(n / every + FZ_TILE); back.reserve(n);
            uint32_t ns = 0;
            for (uint32_t t = 0; t < v.ntiles; ++t) {
                const uint64_t lo = (uint64_t)t * FZ_TILE, hi = std::min<uint64_t>(n, lo + FZ_TILE);
                const bool sample = t % every == every / 2 && t + 1 != v.ntiles;
                (sample ? front : back).insert((sample ? front : back).end(), keys.begin() + lo, keys.begin() + hi);
                ns += sample ? 1u : 0u;
            }
            if (ns != 0) {
                front.insert(front.end(), back.begin(), back.end());
                keys.swap(front);
                v.n_sample_tiles = ns;
            }
        }
    }
    v.h_orig_of_phys.assign(v.npad, 0xFFFFFFFFu);
    std::vector<uint32_t> phys_of_orig(n);
    for (uint64_t ph = 0; ph < n; ++ph) {
        uint32_t o = (uint32_t)(keys[ph] & 0xFFFFFFFFull);
        v.h_orig_of_phys[ph] = o;
        phys_of_orig[o] = (uint32_t)ph;
    }
    std::vector<FzTile> tiles(v.ntiles);
    std::vector<uint32_t> tile_maxlen(v.ntiles, 0u);
    uint64_t sym_words = 0, tok_slots = 0;
    for (uint32_t t = 0; t < v.ntiles; ++t) {
        uint32_t ndw = 0, ntok = 0;
        for (uint32_t l = 0; l < FZ_TILE; ++l) {
            uint64_t ph = (uint64_t)t * FZ_TILE + l;
            if (ph >= n) break;
            uint32_t o = v.h_orig_of_phys[ph];
            tile_maxlen[t] = std::max<uint32_t>(tile_maxlen[t], (uint32_t)(canon_off[o + 1] - canon_off[o]));
            ndw = std::max<uint32_t>(ndw, (uint32_t)((canon_off[o + 1] - canon_off[o] + spd - 1) / spd));
            ntok = std::max<uint32_t>(ntok, (uint32_t)(tok_off[o + 1] - tok_off[o]));
        }
        if (ntok > 0xFFFFu) return fail(FZ_ERR_LIMIT, "a candidate has more than 65535 tokens");
        if (sym_words + (uint64_t)ndw * 64 > 0xFFFFFFFFull || tok_slots + (uint64_t)ntok * 64 > 0xFFFFFFFFull)
            return fail(FZ_ERR_LIMIT, "shard too large for 32-bit tile offsets; use more shards");
        tiles[t].coff = (uint32_t)sym_words; tiles[t].toff = (uint32_t)tok_slots;
        tiles[t].ndw = (uint16_t)ndw; tiles[t].ntok = (uint16_t)ntok;
        sym_words += (uint64_t)ndw * 64; tok_slots += (uint64_t)ntok * 64;
        v.max_ndw = std::max(v.max_ndw, ndw);
    }
    v.n_sym_words = sym_words; v.n_tok_slots = tok_slots;
    std::vector<uint32_t> h_sym(sym_words, 0u);
    std::vector<uint32_t> h_tokid(tok_slots, FZ_TOK_NONE);
    std::vector<uint16_t> h_tokend(tok_slots, 0);
    for (uint32_t t = 0; t < v.ntiles; ++t) {
        const FzTile &td = tiles[t];
        for (uint32_t l = 0; l < FZ_TILE; ++l) {
            uint64_t ph = (uint64_t)t * FZ_TILE + l;
            uint32_t len = 0, nt = 0;
            if (ph < n) {
                uint32_t o = v.h_orig_of_phys[ph];
                len = (uint32_t)(canon_off[o + 1] - canon_off[o]);
                nt = (uint32_t)(tok_off[o + 1] - tok_off[o]);
                const uint16_t *src = canon.data() + canon_off[o];
                for (uint32_t j = 0; j < len; ++j)
                    h_sym[td.coff + (uint64_t)(j / spd) * 64 + l] |= (uint32_t)src[j] << ((32u / spd) * (j % spd));
                for (uint32_t i = 0; i < nt; ++i) {
                    h_tokid[td.toff + (uint64_t)i * 64 + l] = tok_id[tok_off[o] + i];
                    h_tokend[td.toff + (uint64_t)i * 64 + l] = tok_end[tok_off[o] + i];
                }
            }
            for (uint32_t i = nt; i < td.ntok; ++i) h_tokend[td.toff + (uint64_t)i * 64 + l] = (uint16_t)len;
        }
    }

    int rc;
    if ((rc = upload(&v.d_sym, h_sym, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_tokid, h_tokid, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_tokend, h_tokend, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_tiles, tiles, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_orig_of_phys, v.h_orig_of_phys, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload(&v.d_phys_of_orig, phys_of_orig, &v.device_bytes)) != FZ_OK) return rc;
    if ((rc = upload_tile_lists(v, tiles, tile_maxlen)) != FZ_OK) return rc;
    v.built = true;
    return FZ_OK;
}

static int finish_create(fz_corpus *c);
static hipEvent_t get_event(fz_corpus *c, size_t i);

static int corpus_create(const uint32_t *cps, const uint64_t *offs, uint64_t n, int device,
                         uint64_t base_index, int eager_view, fz_corpus **out)
{
    if (!out) return fail(FZ_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n && (!cps || !offs)) return fail(FZ_ERR_INVALID, "cps/offs null with n > 0");
    if (n >= 0xFFFFFFFFull - 64) return fail(FZ_ERR_LIMIT, "shard holds %llu candidates; limit is 2^32-65 per shard", (unsigned long long)n);
    if (base_index + n > FZ_IDX_MASK) return fail(FZ_ERR_LIMIT, "global index beyond 2^40");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FZ_ERR_NO_DEVICE, "no HIP device visible: libfuzzyhip has no CPU path");
    if (device < 0 || device >= ndev) return fail(FZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));
    std::call_once(g_ties_once, make_tie_tables);
    for (uint64_t i = 0; i < n; ++i)
        if (offs[i + 1] < offs[i]) return fail(FZ_ERR_INVALID, "offsets not monotone at %llu", (unsigned long long)i);
    const uint64_t lo = n ? offs[0] : 0, hi = n ? offs[n] : 0;
    for (uint64_t i = lo; i < hi; ++i)
        if (cps[i] >= 0x110000u) return fail(FZ_ERR_INVALID, "code point 0x%x out of range at %llu", cps[i], (unsigned long long)i);

    struct CorpusDeleter { void operator()(fz_corpus *p) const { fz_corpus_destroy(p); } };
    std::unique_ptr<fz_corpus, CorpusDeleter> holder(new fz_corpus());     // released on every way out
    fz_corpus *c = holder.get();
    c->device = device; c->base = base_index; c->n = n;
    c->h_cps.assign(cps + lo, cps + hi);
    c->h_offs.resize(n + 1);
    for (uint64_t i = 0; i <= n; ++i) c->h_offs[i] = n ? offs[i] - lo : 0;
    int rc = build_view(c, eager_view);
    if (rc == FZ_OK) rc = finish_create(c);
    if (rc != FZ_OK) return rc;
    *out = holder.release();
    return FZ_OK;
}

// score tables, tuning knobs, the corpus' own stream: common tail of create and load
static int finish_create(fz_corpus *c)
{
    int rc = FZ_OK;
    for (int form = 0; form < 2 && rc == FZ_OK; ++form) {
        uint64_t unused = 0;
        rc = upload(&c->d_score_tab[form], make_score_table(form == 0 ? g_ties_norm_distance : g_ties_norm_sim), &unused);
    }
    if (rc != FZ_OK) return rc;
    if (const char *e = getenv("FUZZYHIP_NO_STREAM")) c->use_stream = !(e[0] == '1');
    if (const char *e = getenv("FUZZYHIP_NO_FUSED")) c->use_fused = !(e[0] == '1');
    if (const char *e = getenv("FUZZYHIP_PREP_THREADS")) c->prep_threads = (uint32_t)std::min(8, std::max(1, atoi(e)));
    if (const char *e = getenv("FUZZYHIP_FUSED_MIN_MB")) c->fused_min_bytes = (uint64_t)std::max(0, atoi(e)) << 20;
    if (const char *e = getenv("FUZZYHIP_SLAB_KB")) c->slab_bytes = (uint64_t)std::max(1, atoi(e)) << 10;
    if (const char *e = getenv("FUZZYHIP_STREAM_WAVES")) c->stream_waves = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("FUZZYHIP_STREAM_BLOCKS")) c->stream_blocks = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("FUZZYHIP_STREAM_BLOCKS_SAMPLE")) c->stream_blocks_sample = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("FUZZYHIP_STREAM_MAX_GROUP")) c->stream_max_group = (uint32_t)std::max(32, atoi(e));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = (uint32_t)prop.multiProcessorCount;
        HIP_TRY(hipMalloc((void **)&c->own_ws.d_counters, FZ_COUNTER_SLOTS * FZ_STREAM_QUEUES * FZ_STREAM_QPAD * sizeof(uint32_t)));
        HIP_TRY(hipMemset(c->own_ws.d_counters, 0, FZ_COUNTER_SLOTS * FZ_STREAM_QUEUES * FZ_STREAM_QPAD * sizeof(uint32_t)));
    }
    // Everything a call of fz_query_topk_noalloc will need is set up here, once: pinned staging of a fixed
    // size, the events its timings use, and this device's address of the static scratch array.
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc((void **)&c->h_blob[i], FZ_NOALLOC_BLOB_BYTES, hipHostMallocDefault));
        c->h_blob_cap[i] = FZ_NOALLOC_BLOB_BYTES;
        HIP_TRY(hipEventCreateWithFlags(&c->ev_blob[i], hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc((void **)&c->h_keys, FZ_NOALLOC_KEYS_BYTES, hipHostMallocDefault));
    c->h_keys_cap = FZ_NOALLOC_KEYS_BYTES;
    if (!get_event(c, 7)) return fail(FZ_ERR_HIP, "hipEventCreate failed");
    {
        void *scratch = nullptr; size_t scratch_bytes = 0;
        HIP_TRY(fz_scratch_acquire(c->device, &scratch, &scratch_bytes));
    }
    if (const char *e = getenv("FUZZYHIP_SELECT_SLICE")) c->select_slice = (uint32_t)std::max(0, atoi(e)) & ~15u;
    if (const char *e = getenv("FUZZYHIP_STREAM_MIN_QUERIES")) c->stream_min_queries = (uint32_t)std::max(0, atoi(e));
    HIP_TRY(hipMalloc((void **)&c->d_slice_keys, (size_t)FZ_MAX_K * 8 * c->select_slice_rows));
    HIP_TRY(hipMalloc((void **)&c->d_slice_done, 256));
    HIP_TRY(hipMemset(c->d_slice_done, 0, 256));
    if (const char *e = getenv("FUZZYHIP_STREAM_OVERSUB")) c->stream_oversub = (uint32_t)std::min(16, std::max(1, atoi(e)));
    if (const char *e = getenv("FUZZYHIP_SUBMIT_EAGER")) c->submit_eager = atoi(e) != 0;
    if (const char *e = getenv("FUZZYHIP_CHUNKS")) c->pipe_chunks = (uint32_t)std::min(64, std::max(1, atoi(e)));
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(FZ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    c->stream = c->own_stream;
    if (const char *ev = getenv("FUZZYHIP_NO_OVERLAP")) c->overlap = !(ev[0] == '1');
    for (int i = 0; i < 2 && c->overlap; ++i)
        if (hipStreamCreateWithFlags(&c->aux_stream[i], hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev_join[i]) != hipSuccess) c->overlap = false;
    if (c->overlap && hipEventCreate(&c->ev_fork) != hipSuccess) c->overlap = false;
    return FZ_OK;
}

static int fz_corpus_create_impl(const uint32_t *cps, const uint64_t *offs, uint64_t n, int device,
                                uint64_t base_index, fz_corpus **out)
{
    return corpus_create(cps, offs, n, device, base_index, VIEW_SET, out);
}

// ---------------------------------------------------------------------------------------------
// packed corpus file: the built device views, so that a process start is read + upload instead of
// tokenise + sort + transpose (candidate provenance: src/synapse_connection.py:103-121)
// ---------------------------------------------------------------------------------------------
// layout: "FZCORP01" | u64 n | u64 base | u64 vocab count | vocab tokens {u32 len, u32 cps[len]} in id
// order | input {u64 ncps, u32 cps[], u64 offs[n+1]} | per view {u32 built; if built: header of u64s,
// sparse alphabet {u32 cp, u32 sym}, then the arrays exactly as they sit in HBM}.
static const char kFileMagic[8] = {'F', 'Z', 'C', 'O', 'R', 'P', '0', '1'};

template <typename T>
static bool put(FILE *f, const T *p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }
template <typename T>
static bool get(FILE *f, T *p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }

template <typename T>
static int dump_device(FILE *f, const T *d, size_t n)
{
    std::vector<T> h(n);
    if (n) HIP_TRY(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    if (!put(f, h.data(), n)) return fail(FZ_ERR_INVALID, "write failed");
    return FZ_OK;
}

template <typename T>
static int load_device(FILE *f, T **d, size_t n, uint64_t *bytes)
{
    std::vector<T> h(n);
    if (!get(f, h.data(), n)) return fail(FZ_ERR_INVALID, "corpus file is truncated");
    return upload(d, h, bytes);
}

static int fz_corpus_save_impl(fz_corpus *c, const char *path)
{
    if (!c || !path) return fail(FZ_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(FZ_ERR_INVALID, "cannot open %s for writing", path);
    int rc = FZ_OK;
    bool ok = put(f, kFileMagic, 8);
    const uint64_t hdr[3] = {c->n, c->base, (uint64_t)c->vocab_store.size()};
    ok = ok && put(f, hdr, 3);
    for (const std::u32string &t : c->vocab_store) {         // id order: ids are handed out in push order
        const uint32_t len = (uint32_t)t.size();
        ok = ok && put(f, &len, 1) && put(f, reinterpret_cast<const uint32_t *>(t.data()), len);
    }
    const uint64_t ncps = c->h_cps.size();
    ok = ok && put(f, &ncps, 1) && put(f, c->h_cps.data(), ncps) && put(f, c->h_offs.data(), c->h_offs.size());
    for (int k = 0; k < N_VIEWS && ok && rc == FZ_OK; ++k) {
        const View &v = c->views[k];
        const uint32_t built = v.built ? 1u : 0u;
        ok = ok && put(f, &built, 1);
        if (!built) continue;
        std::vector<uint32_t> alpha;
        for (uint32_t cp = 0; cp < 0x110000u; ++cp)
            if (v.sym_of_cp[cp]) { alpha.push_back(cp); alpha.push_back(v.sym_of_cp[cp]); }
        const uint64_t vh[16] = {v.wide, v.npad, v.ntiles, v.max_ndw, v.max_len, v.max_tok, v.alphabet,
                                 v.total_symbols, v.total_tokens, v.n_sym_words, v.n_tok_slots,
                                 v.n_s1, v.n_s2, v.n_slow, (uint64_t)alpha.size(), v.n_sample_tiles};
        ok = ok && put(f, vh, 16) && put(f, alpha.data(), alpha.size());
        if (!ok) break;
        if ((rc = dump_device(f, v.d_sym, v.n_sym_words)) != FZ_OK) break;
        if ((rc = dump_device(f, v.d_tokid, v.n_tok_slots)) != FZ_OK) break;
        if ((rc = dump_device(f, v.d_tokend, v.n_tok_slots)) != FZ_OK) break;
        if ((rc = dump_device(f, v.d_tiles, v.ntiles)) != FZ_OK) break;
        if ((rc = dump_device(f, v.d_orig_of_phys, v.npad)) != FZ_OK) break;
        if ((rc = dump_device(f, v.d_phys_of_orig, (size_t)c->n)) != FZ_OK) break;
    }
    if (fclose(f) != 0) ok = false;
    if (rc != FZ_OK) return rc;
    if (!ok) return fail(FZ_ERR_INVALID, "write to %s failed", path);
    return FZ_OK;
}

static int finish_create(fz_corpus *c);

static int fz_corpus_load_impl(const char *path, int device, uint64_t base_index, fz_corpus **out)
{
    if (!out || !path) return fail(FZ_ERR_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FZ_ERR_NO_DEVICE, "no HIP device visible: libfuzzyhip has no CPU path");
    if (device < 0 || device >= ndev) return fail(FZ_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));
    std::call_once(g_ties_once, make_tie_tables);
    // the file and the half-built corpus are released on every way out, an exception included
    struct FileCloser { void operator()(FILE *fp) const { if (fp) fclose(fp); } };
    struct CorpusDeleter { void operator()(fz_corpus *p) const { fz_corpus_destroy(p); } };
    std::unique_ptr<FILE, FileCloser> file(fopen(path, "rb"));
    FILE *f = file.get();
    if (!f) return fail(FZ_ERR_INVALID, "cannot open %s", path);
    if (fseek(f, 0, SEEK_END) != 0) return fail(FZ_ERR_INVALID, "%s: cannot seek", path);
    const long long fsize_ll = ftell(f);
    if (fsize_ll < 0 || fseek(f, 0, SEEK_SET) != 0) return fail(FZ_ERR_INVALID, "%s: cannot seek", path);
    const uint64_t fsize = (uint64_t)fsize_ll;
    // a declared element count is believed only if that many elements can still follow in the file: a
    // corrupt size field is refused here instead of reaching a vector constructor
    auto fits = [&](uint64_t count, uint64_t elem_bytes) {
        const long long pos = ftell(f);
        return pos >= 0 && count <= (fsize - (uint64_t)pos) / elem_bytes;
    };
    std::unique_ptr<fz_corpus, CorpusDeleter> holder(new fz_corpus());
    fz_corpus *c = holder.get();
    auto bad = [&](const char *why) { return fail(FZ_ERR_INVALID, "%s: %s", path, why); };
    char magic[8]; uint64_t hdr[3];
    if (!get(f, magic, 8) || memcmp(magic, kFileMagic, 8) != 0) return bad("not a packed corpus file (magic FZCORP01)");
    if (!get(f, hdr, 3)) return bad("truncated header");
    c->device = device; c->n = hdr[0];
    c->base = base_index == ~0ull ? hdr[1] : base_index;      // ~0: keep the base index the file was saved with
    if (c->n >= 0xFFFFFFFFull - 64 || c->base + c->n > FZ_IDX_MASK) return bad("candidate count or base index out of range");
    if (!fits(hdr[2], 4)) return bad("vocabulary count exceeds the file size");
    for (uint64_t i = 0; i < hdr[2]; ++i) {
        uint32_t len;
        if (!get(f, &len, 1) || len > FZ_MAX_CAND_LEN) return bad("bad vocabulary entry");
        std::u32string t(len, U'\0');
        if (!get(f, reinterpret_cast<uint32_t *>(t.data()), len)) return bad("truncated vocabulary");
        c->vocab_store.push_back(std::move(t));
        c->vocab.emplace(Tok(c->vocab_store.back()), (uint32_t)i);
    }
    uint64_t ncps;
    if (!get(f, &ncps, 1) || ncps > (1ull << 40) || !fits(ncps, 4)) return bad("bad input section");
    c->h_cps.resize(ncps);
    if (!fits(ncps + 2 * (c->n + 1), 4)) return bad("truncated input section");
    c->h_offs.resize(c->n + 1);
    if (!get(f, c->h_cps.data(), ncps) || !get(f, c->h_offs.data(), c->n + 1)) return bad("truncated input section");
    for (uint64_t i = 0; i < c->n; ++i)
        if (c->h_offs[i + 1] < c->h_offs[i] || c->h_offs[i + 1] > ncps) return bad("offsets out of range");
    int rc = FZ_OK;
    for (int k = 0; k < N_VIEWS; ++k) {
        View &v = c->views[k];
        uint32_t built;
        if (!get(f, &built, 1)) return bad("truncated view table");
        if (!built) continue;
        uint64_t vh[16];
        if (!get(f, vh, 16)) return bad("truncated view header");
        v.wide = vh[0] != 0; v.npad = (uint32_t)vh[1]; v.ntiles = (uint32_t)vh[2]; v.max_ndw = (uint32_t)vh[3];
        v.max_len = (uint32_t)vh[4]; v.max_tok = (uint32_t)vh[5]; v.alphabet = (uint32_t)vh[6];
        v.total_symbols = vh[7]; v.total_tokens = vh[8]; v.n_sym_words = vh[9]; v.n_tok_slots = vh[10];
        v.n_s1 = (uint32_t)vh[11]; v.n_s2 = (uint32_t)vh[12]; v.n_slow = (uint32_t)vh[13];
        v.n_sample_tiles = (uint32_t)vh[15];             // 0 (files of the first format revision): no sample in front
        if (vh[15] > vh[2]) return bad("inconsistent view header");
        if (v.npad != (uint32_t)((c->n + FZ_TILE - 1) / FZ_TILE) * FZ_TILE || v.ntiles != v.npad / FZ_TILE ||
            v.alphabet > 0x10000u || v.n_sym_words > 0xFFFFFFFFull || v.n_tok_slots > 0xFFFFFFFFull ||
            vh[14] > 2ull * 0x110000ull || (vh[14] & 1ull))
            return bad("inconsistent view header");
        // what the header promises must still be in the file: alphabet pairs, symbols, token ids and ends,
        // tiles, both permutations
        if (!fits(vh[14] + v.n_sym_words + v.n_tok_slots + (v.n_tok_slots + 1) / 2 + 3ull * v.ntiles + v.npad + c->n, 4))
            return bad("view arrays exceed the file size");
        std::vector<uint32_t> alpha(vh[14]);
        if (!get(f, alpha.data(), alpha.size())) return bad("truncated alphabet");
        v.sym_of_cp.assign(0x110000, 0);
        for (size_t i = 0; i < alpha.size(); i += 2) {
            if (alpha[i] >= 0x110000u || alpha[i + 1] >= v.alphabet) return bad("bad alphabet entry");
            v.sym_of_cp[alpha[i]] = (uint16_t)alpha[i + 1];
        }
        std::vector<FzTile> tiles(v.ntiles);
        std::vector<uint32_t> tile_maxlen(v.ntiles, 0u);
        {
            std::vector<uint32_t> h_sym(v.n_sym_words), h_tokid(v.n_tok_slots);
            std::vector<uint16_t> h_tokend(v.n_tok_slots);
            if (!get(f, h_sym.data(), h_sym.size()) || !get(f, h_tokid.data(), h_tokid.size()) ||
                !get(f, h_tokend.data(), h_tokend.size()) || !get(f, tiles.data(), tiles.size()))
                return bad("truncated view arrays");
            // the kernels trust these values (LDS rows are indexed by symbol, keep-masks by token end):
            // check every one of them before anything reaches the device
            const uint32_t spd = v.wide ? 2u : 4u, sbits = 32u / spd, smask = v.wide ? 0xFFFFu : 0xFFu;
            for (const FzTile &t : tiles) {
                if ((uint64_t)t.coff + (uint64_t)t.ndw * 64 > v.n_sym_words || (uint64_t)t.toff + (uint64_t)t.ntok * 64 > v.n_tok_slots ||
                    t.ndw > v.max_ndw)
                    return bad("tile table points outside the symbol or token arrays");
                for (uint64_t i = t.coff; i < (uint64_t)t.coff + (uint64_t)t.ndw * 64; ++i)
                    for (uint32_t k = 0; k < spd; ++k)
                        if (((h_sym[i] >> (sbits * k)) & smask) >= v.alphabet) return bad("symbol outside the alphabet");
                for (uint32_t l = 0; l < FZ_TILE; ++l) {
                    uint32_t prev = 0;
                    for (uint32_t i = 0; i < t.ntok; ++i) {
                        const uint32_t e = h_tokend[t.toff + (uint64_t)i * 64 + l];
                        if (e < prev || e > (uint32_t)t.ndw * spd) return bad("token end outside its candidate");
                        prev = e;
                    }
                    uint32_t &ml = tile_maxlen[&t - tiles.data()];     // the last slot's end is the length
                    ml = std::max(ml, prev);
                }
            }
            if ((rc = upload(&v.d_sym, h_sym, &v.device_bytes)) != FZ_OK) break;
            if ((rc = upload(&v.d_tokid, h_tokid, &v.device_bytes)) != FZ_OK) break;
            if ((rc = upload(&v.d_tokend, h_tokend, &v.device_bytes)) != FZ_OK) break;
            if ((rc = upload(&v.d_tiles, tiles, &v.device_bytes)) != FZ_OK) break;
        }
        v.h_orig_of_phys.resize(v.npad);
        if (!get(f, v.h_orig_of_phys.data(), v.npad)) return bad("truncated permutation");
        for (uint32_t ph = 0; ph < v.npad; ++ph)
            if (ph < c->n ? v.h_orig_of_phys[ph] >= c->n : v.h_orig_of_phys[ph] != 0xFFFFFFFFu) return bad("bad permutation");
        if ((rc = upload(&v.d_orig_of_phys, v.h_orig_of_phys, &v.device_bytes)) != FZ_OK) break;
        {
            std::vector<uint32_t> pofo(c->n);
            if (!get(f, pofo.data(), pofo.size())) return bad("truncated permutation");
            for (uint64_t o = 0; o < c->n; ++o)
                if (pofo[o] >= c->n || v.h_orig_of_phys[pofo[o]] != o) return bad("permutations do not invert each other");
            if ((rc = upload(&v.d_phys_of_orig, pofo, &v.device_bytes)) != FZ_OK) break;
        }
        rc = upload_tile_lists(v, tiles, tile_maxlen);
        if (rc != FZ_OK) break;
        v.built = true;
    }
    file.reset();
    if (rc == FZ_OK && !c->views[VIEW_SET].built) rc = build_view(c, VIEW_SET);
    if (rc == FZ_OK) rc = finish_create(c);
    if (rc != FZ_OK) return rc;
    *out = holder.release();
    return FZ_OK;
}

extern "C" int fz_corpus_info(const fz_corpus *c, fz_info *o)
{
    if (!c || !o) return fail(FZ_ERR_INVALID, "null argument");
    const View &v = c->views[VIEW_SET].built ? c->views[VIEW_SET] : c->views[VIEW_RAW];
    o->n = c->n; o->base_index = c->base; o->vocab_size = c->vocab.size();
    o->alphabet_size = v.alphabet; o->n_tiles = v.ntiles; o->max_len = v.max_len;
    o->max_tokens = v.max_tok; o->device_bytes = 0;
    for (const View &w : c->views) o->device_bytes += w.device_bytes;
    o->total_symbols = v.total_symbols; o->total_tokens = v.total_tokens; o->device = c->device;
    o->n_packed_pairs = v.n_s0 / 2u; o->n_stream1_tiles = v.n_s1; o->n_stream2_tiles = v.n_s2; o->n_general_tiles = v.n_slow;
    return FZ_OK;
}

extern "C" int fz_set_stream(fz_corpus *c, void *hip_stream)
{
    if (!c) return fail(FZ_ERR_INVALID, "null corpus");
    std::lock_guard<std::mutex> lk(c->mu);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return FZ_OK;
}

extern "C" int fz_set_workspace_limit(fz_corpus *c, uint64_t bytes)
{
    if (!c) return fail(FZ_ERR_INVALID, "null corpus");
    std::lock_guard<std::mutex> lk(c->mu);
    c->ws_limit = std::max<uint64_t>(bytes, 1);
    return FZ_OK;
}

extern "C" int fz_last_timing(const fz_corpus *c, fz_timing *out)
{
    if (!c || !out) return fail(FZ_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(const_cast<fz_corpus *>(c)->mu);   // a query call on another thread rewrites it
    *out = c->timing;
    return FZ_OK;
}

// ---------------------------------------------------------------------------------------------
// query preparation
// ---------------------------------------------------------------------------------------------
struct QueryBatch {
    // per width class W: the prepared queries that use fz_score_general<W> over every tile ...
    std::map<int, std::vector<FzQuery>> classes;
    // ... and those the stream kernel covers; the general kernel then only sees the slow tiles
    std::map<int, std::vector<FzQuery>> slow_classes;
    std::vector<uint32_t> peq, tok_id, tok_pos, tok_len1;
    // stream kernel input (fz_common.h FzStreamParams)
    bool allow_stream = false;
    uint32_t stream_max_m = 0;           // longest pattern the score table covers
    std::vector<FzRec> recs;
    std::vector<uint32_t> qrec;
    uint8_t row_of_sym[256];
    uint32_t nrows = 1;                  // row 0 is never written: the all-zero row that pads a token record
    bool frozen = false;                 // rows were handed out beforehand (parallel build): a symbol without one makes the query ineligible
    struct Trip { uint32_t id, pos, len1; };
    std::u32string s_pat;                // scratch of prepare_query, reused from query to query
    std::vector<Trip> s_trips;
    std::vector<uint8_t> s_text;
    std::vector<uint32_t> s_tok_first, s_ids;
    QueryBatch() { memset(row_of_sym, 0xFF, sizeof row_of_sym); }
};

// Appends the stream records of one query; false (and nothing appended) when the query is not
// eligible: too long for the score table, or its symbols would overflow the Peq rows of the launch.
//   text: the symbols of the query text in order; for VIEW_SET, `tok_first[t]`..`tok_first[t+1]` are
//   the symbols of token t (code points no candidate uses already dropped) and ids[t] its vocabulary
//   id or FZ_QTOK_UNKNOWN; for the other views there is one pseudo token.
static bool stream_append(QueryBatch &b, bool token_set, uint32_t out_row, uint32_t m,
                          const std::vector<uint8_t> &text, const std::vector<uint32_t> &tok_first,
                          const std::vector<uint32_t> &ids)
{
    if (!b.allow_stream || m > b.stream_max_m) return false;
    // rows for symbols this launch has not seen yet
    uint8_t fresh[256]; uint32_t nfresh = 0;
    for (uint8_t sy : text) {
        if (b.row_of_sym[sy] != 0xFF) continue;
        bool dup = false;
        for (uint32_t i = 0; i < nfresh; ++i) dup |= fresh[i] == sy;
        if (!dup) fresh[nfresh++] = sy;
    }
    if (b.nrows + nfresh > FZ_STREAM_MAX_ROWS + 1u) return false;
    if (b.frozen && nfresh != 0u) return false;

    // header {out_row, m | n_ids << 8 | token records << 16, id0, id1}, then the other known ids four to a record
    uint32_t known[16], n_ids = 0;
    if (token_set)
        for (uint32_t id : ids) {
            if (id == FZ_QTOK_UNKNOWN) continue;
            if (n_ids == FZ_STREAM_MAX_IDS) return false;             // two bits per token in a 32-bit lane string
            known[n_ids++] = id;
        }
    for (uint32_t i = 0; i < nfresh; ++i) b.row_of_sym[fresh[i]] = (uint8_t)(4u * b.nrows++);
    const size_t hdr = b.recs.size();
    b.qrec.push_back((uint32_t)hdr);
    b.recs.push_back(FzRec{out_row, 0u, n_ids > 0 ? known[0] : FZ_QTOK_PAD, n_ids > 1 ? known[1] : FZ_QTOK_PAD});
    for (uint32_t j = 2; j < n_ids; j += 4) {
        FzRec r{FZ_QTOK_PAD, FZ_QTOK_PAD, FZ_QTOK_PAD, FZ_QTOK_PAD};
        for (uint32_t k = 0; k < 4 && j + k < n_ids; ++k) (&r.x)[k] = known[j + k];
        b.recs.push_back(r);
    }
    // token records {symbols 0-3, 4-7, 8-11 as row bytes (0 = the all-zero row: padding), class | last << 3 |
    // first << 4 | 2 * tpos << 8}: the kernel runs 4, 8 or 12 steps by class, so that no branch depends on the
    // length.  tpos is where the lanes keep "the candidate has this token": the kernels shift one bit per known
    // id into a per-lane string (two in the packed kernel), so the k-th known token of n sits at bit n - 1 - k;
    // a token without a vocabulary id points at bit 15, which no string of <= 15 ids reaches (always zero).
    uint32_t nrec = 0, k_known = 0;
    const size_t ntok = tok_first.size() - 1;
    for (size_t t = 0; t < ntok; ++t) {
        const uint32_t a = tok_first[t], e = tok_first[t + 1];
        uint32_t pos = a;
        bool first = true;
        uint32_t tpos = 15u;
        if (token_set && ids[t] != FZ_QTOK_UNKNOWN) tpos = n_ids - 1u - k_known++;
        do {                                             // an empty token still owns its separator step
            const uint32_t n = std::min<uint32_t>(12u, e - pos);
            FzRec r{0u, 0u, 0u, n <= 4u ? 0u : (n <= 8u ? 1u : 2u)};
            for (uint32_t k = 0; k < n; ++k)
                (&r.x)[k >> 2] |= (uint32_t)b.row_of_sym[text[pos + k]] << (8u * (k & 3u));
            pos += n;
            if (first) r.w |= 16u;
            r.w |= (2u * tpos) << 8;                     // a continuation record repeats its token's position
            if (pos >= e) r.w |= 8u;                     // last record of the token
            b.recs.push_back(r);
            ++nrec;
            first = false;
        } while (pos < e);
    }
    if (nrec > 0xFFFFu) { b.recs.resize(hdr); b.qrec.pop_back(); return false; }
    b.recs[hdr].y = m | (n_ids << 8) | (nrec << 16);
    return true;
}

// VIEW_SET: pattern = sorted unique tokens, each followed by one separator (m code points); tokens
// known to the corpus vocabulary are listed as (id, first bit, len+1), ascending id, for the
// kernel's binary search.  VIEW_SORT: pattern = sorted tokens joined by single spaces.  VIEW_RAW:
// the query as given.  Peq[s][w] has bit (i % 32) of word i / 32 set when pattern position i holds
// symbol s; code points the corpus never uses get no bit anywhere (they can match nothing).
static int prepare_query(const fz_corpus *c, int kind, const uint32_t *s, uint64_t n, uint32_t out_row,
                         QueryBatch &b, std::vector<Tok> &toks)
{
    const View &v = c->views[kind];
    // per-query scratch lives in the batch: no heap traffic per query
    std::u32string &pat = b.s_pat;
    std::vector<QueryBatch::Trip> &trips = b.s_trips;
    std::vector<uint8_t> &text = b.s_text;     // stream kernel: text symbols, separators left out
    std::vector<uint32_t> &tok_first = b.s_tok_first, &ids = b.s_ids;
    pat.clear(); trips.clear(); text.clear(); ids.clear();
    tok_first.assign(1, 0u);
    using Trip = QueryBatch::Trip;
    uint32_t ntok = 0;
    auto sym_of = [&](uint32_t cp) -> uint32_t { return cp < 0x110000u ? v.sym_of_cp[cp] : 0u; };
    if (kind == VIEW_RAW) {
        pat.assign(reinterpret_cast<const char32_t *>(s), (size_t)n);
    } else {
        split_sorted(s, n, kind == VIEW_SET, toks);
        ntok = (uint32_t)toks.size();
        for (size_t ti = 0; ti < toks.size(); ++ti) {
            Tok t = toks[ti];
            if (kind == VIEW_SORT && ti) pat.push_back(U' ');
            if (kind == VIEW_SET) {
                auto it = c->vocab.find(t);
                if (it != c->vocab.end()) trips.push_back({it->second, (uint32_t)pat.size(), (uint32_t)t.size() + 1u});
                ids.push_back(it != c->vocab.end() ? it->second : FZ_QTOK_UNKNOWN);
                if (b.allow_stream)
                    for (char32_t ch : t)
                        if (uint32_t sy = sym_of((uint32_t)ch)) text.push_back((uint8_t)sy);
                tok_first.push_back((uint32_t)text.size());
            }
            pat.append(t);
            if (kind == VIEW_SET) pat.push_back(U' ');
        }
    }
    const uint64_t m = pat.size();
    if (m > FZ_MAX_PATTERN) return fail(FZ_ERR_TOO_LONG, "query pattern of %llu code points exceeds FZ_MAX_PATTERN (%u)", (unsigned long long)m, FZ_MAX_PATTERN);
    if (kind != VIEW_SET) {                    // one pseudo token: the whole text, no separator steps
        if (b.allow_stream)
            for (char32_t ch : pat) {
                const uint32_t sy = (kind == VIEW_SORT && ch == U' ') ? FZ_SYM_SPACE : sym_of((uint32_t)ch);
                if (sy) text.push_back((uint8_t)sy);
            }
        tok_first.push_back((uint32_t)text.size());
        ids.push_back(FZ_QTOK_UNKNOWN);
    }
    const bool streamed = (kind != VIEW_SET || ntok > 0) &&
                          stream_append(b, kind == VIEW_SET, out_row, (uint32_t)m, text, tok_first, ids);
    if (streamed && v.n_slow == 0) return FZ_OK;          // nothing left for the general kernel

    const int W = fz_pick_width((uint32_t)m);
    FzQuery q{};
    q.tok_off = (uint32_t)b.tok_id.size();
    q.n_tok = ntok;
    q.m = (uint32_t)m;
    q.out_row = out_row;
    auto pat_sym = [&](uint32_t bit) -> uint32_t {
        const uint32_t cp = (uint32_t)pat[bit];
        if (kind == VIEW_SET && cp == 0x20u) return FZ_SYM_SPACE;       // the separators just added
        return sym_of(cp);
    };
    if (!v.wide) {
        q.peq_off = (uint32_t)b.peq.size();
        q.peq_rows = v.alphabet;
        b.peq.resize(b.peq.size() + (size_t)v.alphabet * W, 0u);
        uint32_t *peq = b.peq.data() + q.peq_off;
        for (uint32_t bit = 0; bit < m; ++bit)
            if (const uint32_t sym = pat_sym(bit)) peq[sym * W + (bit >> 5)] |= 1u << (bit & 31u);
    } else {
        // wide alphabet: a u16 map symbol -> row, and a table with one row per distinct symbol of THIS
        // pattern; row 0 stays all zero and stands for every other symbol (FZ_SYM_NULL included)
        q.map_off = (uint32_t)b.peq.size();
        b.peq.resize(b.peq.size() + ((size_t)v.alphabet + 1) / 2, 0u);
        uint32_t nrows = 1;
        std::vector<uint32_t> row_of_bit(m, 0u);
        {
            uint16_t *map = reinterpret_cast<uint16_t *>(b.peq.data() + q.map_off);
            for (uint32_t bit = 0; bit < m; ++bit) {
                const uint32_t sym = pat_sym(bit);
                if (!sym) continue;
                if (!map[sym]) map[sym] = (uint16_t)nrows++;
                row_of_bit[bit] = map[sym];
            }
        }
        q.peq_off = (uint32_t)b.peq.size();
        q.peq_rows = nrows;
        b.peq.resize(b.peq.size() + (size_t)nrows * W, 0u);
        uint32_t *peq = b.peq.data() + q.peq_off;
        for (uint32_t bit = 0; bit < m; ++bit)
            if (row_of_bit[bit]) peq[row_of_bit[bit] * W + (bit >> 5)] |= 1u << (bit & 31u);
    }
    std::sort(trips.begin(), trips.end(), [](const Trip &a, const Trip &b2) { return a.id < b2.id; });
    q.n_ids = (uint32_t)trips.size();
    for (const Trip &t : trips) { b.tok_id.push_back(t.id); b.tok_pos.push_back(t.pos); b.tok_len1.push_back(t.len1); }
    (streamed ? b.slow_classes : b.classes)[W].push_back(q);
    return FZ_OK;
}

// ---------------------------------------------------------------------------------------------
// query tables of a large chunk on several host threads
// ---------------------------------------------------------------------------------------------
// A few worker threads, started at the first large chunk and parked on a condition variable.  One parallel
// build runs at a time (a second caller builds its tables itself); a child process made by fork() has no
// workers and builds serially.  The eight-thread limit matches fz_corpus::prep_threads.
class PrepPool {
public:
    static PrepPool &get() { static PrepPool *pool = new PrepPool(); return *pool; }   // never destroyed: the workers end with the process
    // f(0) .. f(n - 1), f(0) on the calling thread; false when the pool is busy or unusable (nothing ran)
    bool run(uint32_t n, const std::function<void(uint32_t)> &f)
    {
        if (n <= 1u || dead_.load()) return false;
        std::unique_lock<std::mutex> owner(owner_mu_, std::try_to_lock);
        if (!owner.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(mu_);
            try {
                while (threads_.size() + 1u < n && threads_.size() < kMaxWorkers) threads_.emplace_back([this] { worker(); });
            } catch (...) {}                                   // no thread to be had: what exists decides below
            if (threads_.size() + 1u < n) return false;        // every part needs its thread: the caller builds serially instead
            job_ = &f; next_ = 1u; end_ = n; pending_ = n - 1u; ++gen_;
        }
        cv_.notify_all();
        // the workers write into the caller's frame: whatever f(0) does, nobody leaves before they are done
        std::exception_ptr err;
        try { f(0); } catch (...) { err = std::current_exception(); }
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [this] { return pending_ == 0u; });
            job_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
        return true;
    }
    static constexpr uint32_t kMaxWorkers = 7;
private:
    PrepPool() { pthread_atfork(nullptr, nullptr, [] { PrepPool::get().dead_.store(true); }); }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            uint32_t idx;
            const std::function<void(uint32_t)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen && next_ < end_; });
                idx = next_++;
                if (next_ >= end_) seen = gen_;
                job = job_;
            }
            try { (*job)(idx); } catch (...) { failed_.store(true); }       // reported by run()'s caller through its own checks
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0u) done_cv_.notify_all();
            }
        }
    }
    std::mutex owner_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    const std::function<void(uint32_t)> *job_ = nullptr;
    uint32_t next_ = 0, end_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    std::atomic<bool> dead_{false};
public:
    std::atomic<bool> failed_{false};       // a worker's part threw (out of memory): set until the caller clears it
};

// The tables of rows [q0, q0 + rows) into `b`, built by `parts` threads over contiguous runs of queries.
// Stream rows (symbol -> Peq row of the launch) are handed out first, in query order, so that every thread
// writes the same row bytes; each thread then fills a batch of its own, and the batches are appended in
// order with their offsets moved.  Scores do not depend on the split: a row number is only a name.
static int prepare_tables_parallel(const fz_corpus *c, int view_kind, const uint32_t *q_cps, const uint64_t *q_offs,
                                   uint64_t q0, uint32_t rows, uint32_t parts, QueryBatch &b, bool *ran)
{
    const View &v = c->views[view_kind];
    *ran = false;
    if (b.allow_stream) {
        for (uint32_t r = 0; r < rows; ++r) {
            const uint64_t qi = q0 + r;
            if (q_offs[qi + 1] < q_offs[qi]) return fail(FZ_ERR_INVALID, "query offsets not monotone at %llu", (unsigned long long)qi);
            uint8_t fresh[256]; uint32_t nfresh = 0;
            bool in_token = false; uint32_t ntok = 0;
            auto want = [&](uint32_t sy) {
                if (sy == 0u || sy > 255u || b.row_of_sym[sy] != 0xFF) return;
                for (uint32_t i = 0; i < nfresh; ++i) if (fresh[i] == sy) return;
                fresh[nfresh++] = (uint8_t)sy;
            };
            for (uint64_t i = q_offs[qi]; i < q_offs[qi + 1]; ++i) {
                const uint32_t cp = q_cps[i];
                if (view_kind != VIEW_RAW && is_space(cp)) { in_token = false; continue; }
                if (!in_token) { in_token = true; ++ntok; }
                want(cp < 0x110000u ? v.sym_of_cp[cp] : 0u);
            }
            if (view_kind == VIEW_SORT && ntok > 1u) want(FZ_SYM_SPACE);
            if (b.nrows + nfresh > FZ_STREAM_MAX_ROWS + 1u) continue;          // this query will not be streamed
            for (uint32_t i = 0; i < nfresh; ++i) b.row_of_sym[fresh[i]] = (uint8_t)(4u * b.nrows++);
        }
    }
    struct Part { QueryBatch b; int rc = FZ_OK; std::string err; };
    std::vector<Part> part(parts);
    for (Part &pt : part) {
        pt.b.allow_stream = b.allow_stream; pt.b.stream_max_m = b.stream_max_m;
        memcpy(pt.b.row_of_sym, b.row_of_sym, sizeof b.row_of_sym);
        pt.b.nrows = b.nrows; pt.b.frozen = true;
    }
    // the calling thread starts at once, the workers have to wake up first: it takes two shares, they one each
    const uint32_t per = (rows + parts) / (parts + 1u);
    const std::function<void(uint32_t)> work = [&](uint32_t t) {
        Part &pt = part[t];
        std::vector<Tok> toks;
        const uint32_t ra = std::min(rows, t == 0u ? 0u : (t + 1u) * per), rb = t + 1u == parts ? rows : std::min(rows, (t + 2u) * per);
        if (pt.b.allow_stream) { pt.b.recs.reserve((size_t)(rb - ra) * 6 + 8); pt.b.qrec.reserve(rb - ra); }
        for (uint32_t r = ra; r < rb; ++r) {
            const uint64_t qi = q0 + r;
            if (q_offs[qi + 1] < q_offs[qi]) { pt.rc = FZ_ERR_INVALID; pt.err = "query offsets not monotone"; return; }
            pt.rc = prepare_query(c, view_kind, q_cps + q_offs[qi], q_offs[qi + 1] - q_offs[qi], r, pt.b, toks);
            if (pt.rc != FZ_OK) { pt.err = g_err; return; }             // g_err is this thread's own
        }
    };
    PrepPool::get().failed_.store(false);
    if (!PrepPool::get().run(parts, work)) return FZ_OK;                 // pool busy or short of threads: the caller builds serially
    if (PrepPool::get().failed_.exchange(false)) return fail(FZ_ERR_LIMIT, "out of host memory while building query tables");
    for (const Part &pt : part)
        if (pt.rc != FZ_OK) return fail(pt.rc, "%s", pt.err.c_str());
    for (Part &pt : part) {
        const uint32_t peq_base = (uint32_t)b.peq.size(), tok_base = (uint32_t)b.tok_id.size(), rec_base = (uint32_t)b.recs.size();
        for (int pass = 0; pass < 2; ++pass)
            for (auto &kv : (pass == 0 ? pt.b.classes : pt.b.slow_classes)) {
                std::vector<FzQuery> &dst = (pass == 0 ? b.classes : b.slow_classes)[kv.first];
                for (FzQuery q : kv.second) {
                    q.peq_off += peq_base; q.map_off += peq_base; q.tok_off += tok_base;
                    dst.push_back(q);
                }
            }
        b.peq.insert(b.peq.end(), pt.b.peq.begin(), pt.b.peq.end());
        b.tok_id.insert(b.tok_id.end(), pt.b.tok_id.begin(), pt.b.tok_id.end());
        b.tok_pos.insert(b.tok_pos.end(), pt.b.tok_pos.begin(), pt.b.tok_pos.end());
        b.tok_len1.insert(b.tok_len1.end(), pt.b.tok_len1.begin(), pt.b.tok_len1.end());
        b.recs.insert(b.recs.end(), pt.b.recs.begin(), pt.b.recs.end());
        for (uint32_t x : pt.b.qrec) b.qrec.push_back(x + rec_base);
    }
    *ran = true;
    return FZ_OK;
}

template <typename T>
static int ensure(const Workspace &w, T **p, size_t *cap, size_t need_bytes, const char *what)
{
    if (need_bytes <= *cap) return FZ_OK;
    if (w.fixed)
        return fail(FZ_ERR_LIMIT, "fz_query_topk_noalloc: the %s of this call needs %zu bytes, its share of the static scratch "
                                  "buffer (%u MiB in all) holds %zu; use fz_query_topk for larger batches", what, need_bytes,
                    (unsigned)(FZ_NOALLOC_SCRATCH_BYTES >> 20), *cap);
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
#ifdef FZ_EXACT_ALLOC          // sanitizer build (tests/emul): no slack, so every overrun is seen
    size_t sz = need_bytes;
#else
    size_t sz = need_bytes + need_bytes / 4 + 256;
#endif
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), sz));
    *cap = sz;
    return FZ_OK;
}

// pinned host buffer of at least `need` bytes (never grown for a fixed workspace: no allocation there)
template <typename T>
static int ensure_pinned(const Workspace &w, T **p, size_t *cap, size_t need_bytes, const char *what)
{
    if (need_bytes <= *cap) return FZ_OK;
    if (w.fixed)
        return fail(FZ_ERR_LIMIT, "fz_query_topk_noalloc: the %s of this call needs %zu bytes of pinned staging, %zu are set "
                                  "aside; use fz_query_topk for larger batches", what, need_bytes, *cap);
    if (*p) { HIP_TRY(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    const size_t sz = need_bytes + need_bytes / 2 + 4096;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(p), sz, hipHostMallocDefault));
    *cap = sz;
    return FZ_OK;
}

static hipEvent_t get_event(fz_corpus *c, size_t i)
{
    while (c->events.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->events.push_back(e);
    }
    return c->events[i];
}

enum RunMode { RUN_TOPK_HOST, RUN_TOPK_DEVICE, RUN_SCORE_ALL, RUN_DIST_ALL };

struct RunOut {
    uint64_t *idx = nullptr; int32_t *score = nullptr; uint32_t *cnt = nullptr;     // TOPK_HOST
    uint64_t *d_keys = nullptr; uint32_t *d_cnt = nullptr; int sync = 1;           // TOPK_DEVICE
    // TOPK_DEVICE without sync: a device word that is 0 after the call's kernels unless a fused list overflowed (then
    // 2: the keys are not final and the caller redoes the call with force_matrix or sync).  Null: an asynchronous
    // call never takes the fused path, because nobody could be told about an overflow.
    uint64_t *d_status = nullptr;
    int force_matrix = 0;
    uint32_t flat = 0;                         // FZ_ORDER_BY_INDEX: every passing candidate ties
    uint8_t *scores = nullptr;                                                      // SCORE_ALL
    uint16_t *dists = nullptr;                                                      // DIST_ALL
};

static void decode_keys(const uint64_t *keys, uint32_t nq, uint32_t k, uint64_t *idx, int32_t *score, uint32_t *cnt)
{
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t c = 0;
        for (uint32_t i = 0; i < k; ++i) {
            uint64_t key = keys[(uint64_t)q * k + i];
            if (key == 0ull) {   // padding: a real key is never 0 (global indices stay below 2^40-1)
                idx[(uint64_t)q * k + i] = 0; score[(uint64_t)q * k + i] = 0;
                continue;
            }
            idx[(uint64_t)q * k + i] = FZ_IDX_MASK - (key & FZ_IDX_MASK);
            score[(uint64_t)q * k + i] = (int32_t)(key >> FZ_IDX_BITS);
            ++c;
        }
        if (cnt) cnt[q] = c;
    }
}

// scorer -> (view, kernel mode, tie table)
static int scorer_plan(int scorer, int *view, int *mode, FzTies *ties)
{
    switch (scorer) {
    case FZ_TOKEN_SET: *view = VIEW_SET; *mode = FZ_MODE_TOKEN_SET; *ties = g_ties_norm_distance; return FZ_OK;
    case FZ_RATIO: *view = VIEW_RAW; *mode = FZ_MODE_RATIO; *ties = g_ties_norm_sim; return FZ_OK;
    case FZ_TOKEN_SORT: *view = VIEW_SORT; *mode = FZ_MODE_RATIO; *ties = g_ties_norm_sim; return FZ_OK;
    case FZ_PARTIAL: *view = VIEW_RAW; *mode = FZ_MODE_PARTIAL; *ties = g_ties_norm_sim; return FZ_OK;
    default:
        return fail(FZ_ERR_INVALID, "scorer %d is not implemented by this build", scorer);
    }
}

// ---------------------------------------------------------------------------------------------
// scoring a chunk of queries: prepare (host tables + upload) and launch (per phase)
// ---------------------------------------------------------------------------------------------
// PH_ALL scores every tile into the score matrix.  A fused top-k call (FzEmit, fz_common.h) scores the
// SAMPLE tiles into a small matrix first (PH_SAMPLE) and the other tiles without one (PH_REST).
enum Phase { PH_ALL = 0, PH_SAMPLE = 1, PH_REST = 2 };

struct Prepared {
    QueryBatch b;
    struct Launch { int W; uint32_t first, count; bool slow_only; };
    std::vector<FzQuery> qlist;
    std::vector<Launch> launches;
    size_t o_q = 0, o_peq = 0, o_id = 0, o_pos = 0, o_len = 0, o_rec = 0, o_qrec = 0, o_rows = 0;
    uint32_t blob = 0; size_t total = 0; double host_ms = 0.0;          // the pinned blob the tables were assembled in
};

// Builds the query tables of `rows` queries (q0 .. q0+rows; query r of the chunk writes score row r) on
// the host, in one of the two pinned blobs (no HIP call but the wait for the blob's last upload).
static int build_tables(fz_corpus *c, int view_kind, int kmode, const uint32_t *q_cps, const uint64_t *q_offs,
                        uint64_t q0, uint32_t rows, Prepared &P)
{
    const View &v = c->views[view_kind];
    int rc;
    std::vector<Tok> toks;
    auto t0 = std::chrono::steady_clock::now();
    QueryBatch &b = P.b;
    const bool stream_mode = kmode == FZ_MODE_TOKEN_SET || kmode == FZ_MODE_RATIO;
    // a handful of queries: the general kernel over every tile (query as the pattern, nothing to set up per tile) beats
    // the stream kernels, whose per-tile set-up pays off from about 8 queries on (1 query x 10^6 candidates: 0.022 ms
    // against 0.11 ms)
    b.allow_stream = c->use_stream && stream_mode && (v.n_s0 + v.n_s1 + v.n_s2) > 0 && rows >= c->stream_min_queries;
    // the score table covers L < 128; a stream tile holds at most 64 symbols.  token_set looks up
    // L = m + clen - 2, ratio L = m + clen.
    b.stream_max_m = kmode == FZ_MODE_TOKEN_SET ? (FZ_STREAM_TAB + 1u - 64u) : (FZ_STREAM_TAB - 1u - 64u);
    if (b.allow_stream) { b.recs.reserve((size_t)rows * 6 + 8); b.qrec.reserve(rows); }
    // large chunks: the tables are built by a few threads (0.19 -> 0.07 ms per 1000 queries); a failed or busy
    // pool leaves `b` with nothing but stream rows handed out, which the serial loop below reuses
    bool built = false;
    const uint32_t parts = std::min<uint32_t>(c->prep_threads, rows / 128u);
    if (parts > 1u) {
        if ((rc = prepare_tables_parallel(c, view_kind, q_cps, q_offs, q0, rows, parts, b, &built)) != FZ_OK) return rc;
    }
    for (uint32_t r = 0; r < rows && !built; ++r) {
        const uint64_t qi = q0 + r;
        if (q_offs[qi + 1] < q_offs[qi]) return fail(FZ_ERR_INVALID, "query offsets not monotone at %llu", (unsigned long long)qi);
        rc = prepare_query(c, view_kind, q_cps + q_offs[qi], q_offs[qi + 1] - q_offs[qi], r, b, toks);
        if (rc != FZ_OK) return rc;
    }
    const auto t_prep = std::chrono::steady_clock::now();
    for (int i = 0; i < 8; ++i) b.recs.push_back(FzRec{0, 0, 0, 0});      // the kernel prefetches records ahead
    if (b.recs.size() >= (1ull << 28))                                    // the kernels address records by 32-bit byte offsets
        return fail(FZ_ERR_LIMIT, "%zu stream records in one chunk of queries (limit 2^28)", b.recs.size());
    // one device blob: [FzQuery list][peq][tok_id][tok_pos][tok_len1][stream records][qrec][rows]
    for (int pass = 0; pass < 2; ++pass)
        for (auto &kv : (pass == 0 ? b.classes : b.slow_classes)) {
            P.launches.push_back({kv.first, (uint32_t)P.qlist.size(), (uint32_t)kv.second.size(), pass == 1});
            P.qlist.insert(P.qlist.end(), kv.second.begin(), kv.second.end());
        }
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    P.o_q = 0; P.o_peq = align16(P.o_q + P.qlist.size() * sizeof(FzQuery));
    P.o_id = P.o_peq + b.peq.size() * 4; P.o_pos = P.o_id + b.tok_id.size() * 4;
    P.o_len = P.o_pos + b.tok_pos.size() * 4;
    P.o_rec = align16(P.o_len + b.tok_len1.size() * 4);
    P.o_qrec = P.o_rec + b.recs.size() * sizeof(FzRec);
    P.o_rows = align16(P.o_qrec + b.qrec.size() * 4);
    const size_t total = P.o_rows + 256;
    // The blob is assembled in pinned host memory and copied without a stream synchronisation: two
    // buffers take turns, and one is rewritten only after the copy that last read it has completed (its
    // event), so the host builds the tables of the next chunk while the GPU works on this one.
    Workspace &w = *c->ws;
    const uint32_t bi = c->blob_next++ & 1u;
    if (c->prebuilt.valid && c->prebuilt.P->blob == bi) c->prebuilt.valid = false;     // tables built ahead lived in this blob
    HIP_TRY(hipEventSynchronize(c->ev_blob[bi]));
    if ((rc = ensure_pinned(w, &c->h_blob[bi], &c->h_blob_cap[bi], total, "query tables")) != FZ_OK) return rc;
    P.blob = bi; P.total = total;
    uint8_t *blob = c->h_blob[bi];
    if (!P.qlist.empty()) memcpy(blob + P.o_q, P.qlist.data(), P.qlist.size() * sizeof(FzQuery));
    if (!b.peq.empty()) memcpy(blob + P.o_peq, b.peq.data(), b.peq.size() * 4);
    if (!b.tok_id.empty()) {
        memcpy(blob + P.o_id, b.tok_id.data(), b.tok_id.size() * 4);
        memcpy(blob + P.o_pos, b.tok_pos.data(), b.tok_pos.size() * 4);
        memcpy(blob + P.o_len, b.tok_len1.data(), b.tok_len1.size() * 4);
    }
    memcpy(blob + P.o_rec, b.recs.data(), b.recs.size() * sizeof(FzRec));
    if (!b.qrec.empty()) memcpy(blob + P.o_qrec, b.qrec.data(), b.qrec.size() * 4);
    memcpy(blob + P.o_rows, b.row_of_sym, 256);
    P.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("FUZZYHIP_PREP_DEBUG"))
        fprintf(stderr, "[prep dbg] %u queries: tables %.3f ms, tables + blob %.3f ms (%zu bytes)\n", rows,
                std::chrono::duration<double, std::milli>(t_prep - t0).count(), P.host_ms, total);
    return FZ_OK;
}

// prepare = tables (built now, or built ahead by a submit that found the GPU busy: c->prebuilt) + upload on `st`
static int prepare_rows(fz_corpus *c, int view_kind, int kmode, const uint32_t *q_cps, const uint64_t *q_offs,
                        uint64_t q0, uint32_t rows, hipStream_t st, Prepared &P)
{
    int rc;
    fz_corpus::Prebuilt &pb = c->prebuilt;
    if (pb.valid && pb.q_cps == q_cps && pb.q_offs == q_offs && pb.q0 == q0 && pb.rows == rows && pb.view_kind == view_kind && pb.kmode == kmode) {
        P = std::move(*pb.P);
        pb.valid = false;
    } else if ((rc = build_tables(c, view_kind, kmode, q_cps, q_offs, q0, rows, P)) != FZ_OK) return rc;
    c->timing.host_prep_ms += P.host_ms;
    Workspace &w = *c->ws;
    if ((rc = ensure(w, &w.d_qbuf, &w.qbuf_cap, P.total, "query tables")) != FZ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(w.d_qbuf, c->h_blob[P.blob], P.total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(c->ev_blob[P.blob], st));
    return FZ_OK;
}

// Launches the scoring kernels of a prepared chunk over the tiles of `ph`.  Scores go to
// d_scores[row][row_stride] (physical position = tile * 64 + lane), or, with `emit`, keys that reach the
// row's limit are appended to its list.
static int launch_rows(fz_corpus *c, int view_kind, int kmode, FzTies ties, const Prepared &P, Phase ph,
                       uint8_t *d_scores, uint64_t row_stride, const FzEmit *emit, hipStream_t st)
{
    const View &v = c->views[view_kind];
    const Workspace &w = *c->ws;
    const QueryBatch &b = P.b;
    auto range = [&](const uint32_t *list, uint32_t n_sample, uint32_t n_total, const uint32_t **out_list, uint32_t *out_n) {
        if (ph == PH_ALL) { *out_list = list; *out_n = n_total; }
        else if (ph == PH_SAMPLE) { *out_list = list; *out_n = n_sample; }
        else { *out_list = list + n_sample; *out_n = n_total - n_sample; }
    };
    for (auto &l : P.launches) {
        const int W = l.W;
        const uint32_t first = l.first, count = l.count;
        FzScoreParams p{};
        p.sym = v.d_sym; p.tokid = v.d_tokid; p.tokend = v.d_tokend; p.tiles = v.d_tiles;
        if (l.slow_only) {
            range(v.d_tiles_slow, v.ns_slow, v.n_slow, &p.tile_list, &p.ntiles);
        } else {                                            // every tile of the phase: a contiguous physical range
            p.tile_list = nullptr;
            p.tile_base = ph == PH_REST ? v.n_sample_tiles : 0u;
            p.ntiles = ph == PH_ALL ? v.ntiles : (ph == PH_SAMPLE ? v.n_sample_tiles : v.ntiles - v.n_sample_tiles);
        }
        if (p.ntiles == 0) continue;
        p.alphabet = v.alphabet;
        p.wide = v.wide ? 1u : 0u;
        uint32_t pmax = 1, peq_rows = 1;
        for (uint32_t i = 0; i < count; ++i) {
            pmax = std::max(pmax, P.qlist[first + i].n_ids);
            peq_rows = std::max(peq_rows, P.qlist[first + i].peq_rows);
        }
        p.pmax = pmax;
        p.peq_rows = peq_rows;
        const uint32_t tkd = v.wide ? 16u : 8u;             // symbol dwords per text keep-mask word
        p.ntkw = std::max<uint32_t>(1, (v.max_ndw + tkd - 1) / tkd);
        p.nwaves = 4;
        p.ro = kmode == FZ_MODE_RO ? 1u : 0u;
        while (p.nwaves > 1 && fz_score_lds_bytes(W, p) > 64 * 1024) p.nwaves >>= 1;
        if (fz_score_lds_bytes(W, p) > 160 * 1024)
            return fail(FZ_ERR_TOO_LONG, "pattern width %d words x candidate length %u does not fit in LDS", W, v.max_len);
        uint64_t work = (uint64_t)p.ntiles * count;
        uint32_t tpb = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(p.nwaves, work / 8192));
        tpb = (tpb + p.nwaves - 1) / p.nwaves * p.nwaves;
        p.tiles_per_block = tpb;
        p.queries = reinterpret_cast<const FzQuery *>(w.d_qbuf + P.o_q) + first;
        p.qblob = reinterpret_cast<const uint32_t *>(w.d_qbuf + P.o_peq);
        p.qtok_id = reinterpret_cast<const uint32_t *>(w.d_qbuf + P.o_id);
        p.qtok_pos = reinterpret_cast<const uint32_t *>(w.d_qbuf + P.o_pos);
        p.qtok_len1 = reinterpret_cast<const uint32_t *>(w.d_qbuf + P.o_len);
        p.scores = d_scores; p.row_stride = row_stride;
        p.ties = ties;
        if (emit) p.emit = *emit;
        HIP_TRY(fz_launch_score(W, kmode, p, count, st));
        c->timing.score_launches++;
    }
    if (!b.qrec.empty()) {
        // odd strides keep the 64 lane columns on distinct banks; the smallest one that holds the rows of
        // this launch leaves room for the most waves (37: the zero row, a-z and 0-9)
        const uint32_t stride = b.nrows <= 37u ? 37u : (b.nrows <= 41u ? 41u : 65u);
        const uint32_t *lists[3]; uint32_t counts[3];
        range(v.d_tiles_s0, v.ns_s0, v.n_s0, &lists[0], &counts[0]);
        range(v.d_tiles_s1, v.ns_s1, v.n_s1, &lists[1], &counts[1]);
        range(v.d_tiles_s2, v.ns_s2, v.n_s2, &lists[2], &counts[2]);
        // The (up to) three stream kernels write disjoint columns of the score matrix and each ends in a
        // tail where CUs that have drained their queue idle: they go to three streams, forked after the
        // upload and joined before anything reads the scores, so that one kernel's blocks start on the CUs
        // another one's tail frees.
        const bool side_by_side = c->overlap && ((counts[0] != 0) + (counts[1] != 0) + (counts[2] != 0)) > 1;
        // the work-queue counters of a call's first FZ_COUNTERS_AHEAD launches are zeroed by ONE fill on the caller's
        // stream, in front of the fork (a fill per launch cost six small kernels per fused call, four of them on the
        // side streams behind the first kernel's persistent blocks); every kernel of earlier calls has finished by then
        if (c->counters_call != c->call_seq) {
            HIP_TRY(hipMemsetAsync(w.d_counters, 0, (size_t)FZ_COUNTERS_AHEAD * FZ_STREAM_QUEUES * FZ_STREAM_QPAD * sizeof(uint32_t), st));
            c->counters_call = c->call_seq; c->counter_next = 0; c->counters_clean = FZ_COUNTERS_AHEAD;
        }
        if (side_by_side) {
            HIP_TRY(hipEventRecord(c->ev_fork, st));
            for (hipStream_t a : c->aux_stream) HIP_TRY(hipStreamWaitEvent(a, c->ev_fork, 0));
        }
        // bytes of symbols, token ids and token ends one tile brings into the caches
        const uint64_t tile_bytes = std::max<uint64_t>(1, (v.n_sym_words * 4ull + v.n_tok_slots * 6ull) / std::max<uint32_t>(1, v.ntiles));
        int n_aux = 0, n_launched = 0;
        bool used_aux[2] = {false, false};
        for (int cls = 0; cls <= 2; ++cls) {           // 0: packed pairs of short tiles, 1 / 2: pattern words
            const int W = cls == 2 ? 2 : 1;
            const uint32_t nt = counts[cls];
            if (nt == 0) continue;
            hipStream_t lst = st;                      // the first kernel stays on the caller's stream
            if (side_by_side && n_launched > 0) { lst = c->aux_stream[n_aux]; used_aux[n_aux] = true; ++n_aux; }
            const uint32_t units = cls == 0 ? nt / 2u : nt;             // tiles (or tile pairs) to hand out
            FzStreamParams p{};
            p.sym = v.d_sym; p.tokid = v.d_tokid; p.tokend = v.d_tokend; p.tiles = v.d_tiles;
            p.tile_list = lists[cls];
            p.ntiles = nt; p.stride = stride;
            // G = 4 waves share one unit's LDS column and split the item's queries (fz_kernels.hip, "work queues").  One
            // pattern word: 64 registers, 8 blocks = 32 waves per CU; two words: 96 registers and a 19 KB column, 5 blocks
            p.nwaves = std::min<uint32_t>(8u, std::max<uint32_t>(1u, c->stream_waves ? c->stream_waves : 4u));
            const size_t lds_b = fz_stream_lds_bytes(W, (int)stride);
            const uint32_t waves_per_cu = W == 2 ? 20u : 32u;
            const uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds_b, waves_per_cu / p.nwaves));
            p.nq = (uint32_t)b.qrec.size();
            // enough blocks to balance the chip, enough queries per block to amortise the tile set-up
            const uint64_t bx = (units + 15u) / 16u;
            const uint32_t want_blocks = ph == PH_SAMPLE ? c->stream_blocks_sample : c->stream_blocks;
            uint64_t by = std::max<uint64_t>(1, std::min<uint64_t>((p.nq + 31) / 32, (want_blocks + bx - 1) / bx));
            // ... and no item so long that the launch ends in a tail of it (a large batch over a large shard)
            by = std::max<uint64_t>(by, (p.nq + c->stream_max_group - 1) / c->stream_max_group);
            p.q_per_block = (uint32_t)((p.nq + by - 1) / by);
            const uint64_t groups = (p.nq + p.q_per_block - 1) / p.q_per_block;
            if ((uint64_t)units * groups > 0xFFFFFF00ull) return fail(FZ_ERR_LIMIT, "too many work items in one launch");
            p.nitems = (uint32_t)((uint64_t)units * groups);
            // slabs: every query group passes over one cache-sized run of units before the next run is touched
            p.slab_units = (uint32_t)std::min<uint64_t>(units, std::max<uint64_t>(1, c->slab_bytes / (tile_bytes * (cls == 0 ? 2u : 1u))));
            // as many persistent blocks as the chip holds at this LDS size and 8 waves per SIMD; each drains the work queues
            p.grid_blocks = c->n_cus * blocks_per_cu * c->stream_oversub;
            // small items (few queries per group) are drawn several at a time, but never so many that
            // the chip's blocks cannot all get a share
            p.chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint32_t>(1u, 32u / p.q_per_block), p.nitems / p.grid_blocks));
            const uint32_t slot = c->counter_next++;
            p.work_counter = w.d_counters + (size_t)(slot % FZ_COUNTER_SLOTS) * FZ_STREAM_QUEUES * FZ_STREAM_QPAD;
            if (slot >= c->counters_clean) HIP_TRY(hipMemsetAsync(p.work_counter, 0, FZ_STREAM_QUEUES * FZ_STREAM_QPAD * sizeof(uint32_t), lst));
            p.row_of_sym = w.d_qbuf + P.o_rows;
            p.recs = reinterpret_cast<const FzRec *>(w.d_qbuf + P.o_rec);
            p.qrec = reinterpret_cast<const uint32_t *>(w.d_qbuf + P.o_qrec);
            p.score_tab = c->d_score_tab[kmode == FZ_MODE_TOKEN_SET ? 0 : 1];
            p.scores = d_scores; p.row_stride = (uint32_t)row_stride;
            if (emit) p.emit = *emit;
            if (cls == 0) HIP_TRY(fz_launch_stream_pk(kmode, p, lst));
            else HIP_TRY(fz_launch_stream(W, kmode, p, lst));
            c->timing.score_launches++;
            ++n_launched;
        }
        for (int i = 0; i < 2; ++i)
            if (used_aux[i]) {
                HIP_TRY(hipEventRecord(c->ev_join[i], c->aux_stream[i]));
                HIP_TRY(hipStreamWaitEvent(st, c->ev_join[i], 0));
            }
    }
    return FZ_OK;
}

// prepare + launch over every tile into the workspace score matrix (the two-corpus path)
static int score_rows(fz_corpus *c, int view_kind, int kmode, FzTies ties, const uint32_t *q_cps,
                      const uint64_t *q_offs, uint64_t q0, uint32_t rows, uint64_t row_stride, hipStream_t st)
{
    Prepared P;
    int rc = prepare_rows(c, view_kind, kmode, q_cps, q_offs, q0, rows, st, P);
    if (rc != FZ_OK) return rc;
    return launch_rows(c, view_kind, kmode, ties, P, PH_ALL, c->ws->d_scores, row_stride, nullptr, st);
}

// Events of one call, for the kernel timings.  A fixed workspace creates nothing, events included: it
// cycles through the eight made at corpus creation (only its last chunk's kernel times are then reported).
struct CallEvents {
    fz_corpus *c; bool fixed; size_t next = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> score, select;
    hipEvent_t take()
    {
        if (fixed && next >= c->events.size()) { next = 1; score.clear(); select.clear(); }
        return get_event(c, next++);
    }
    int mark(hipEvent_t *out, hipStream_t st)
    {
        *out = take();
        if (!*out) return fail(FZ_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(*out, st));
        return FZ_OK;
    }
};

struct TopkTarget { uint64_t *d_keys; uint32_t *d_cnt; int32_t min_score; uint32_t k; uint32_t flat; };

// Rows [qa, qb) through the score matrix: score every tile, then (top-k) fz_select_topk per row, or copy
// the rows back in original candidate order.
static int run_matrix(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint64_t qa, uint64_t qb,
                      int view_kind, int kmode, FzTies ties, RunMode mode, RunOut &o, const TopkTarget &t,
                      CallEvents &ev, hipStream_t st)
{
    const View &v = c->views[view_kind];
    Workspace &w = *c->ws;
    const bool topk = mode == RUN_TOPK_HOST || mode == RUN_TOPK_DEVICE;
    const uint64_t nq = qb - qa;
    int rc;
    const size_t esz = (mode == RUN_DIST_ALL) ? 2 : 1;            // bytes per (query, candidate) result
    const uint64_t row_stride = v.npad;                            // in elements
    // *_ALL modes return rows in ORIGINAL candidate order: a second, n-wide copy of each row is made on the device
    const uint64_t row_bytes = (row_stride + (topk ? 0 : c->n)) * esz;
    uint64_t rows_per_chunk = std::max<uint64_t>(1, c->ws_limit / row_bytes);
    rows_per_chunk = std::min<uint64_t>(rows_per_chunk, nq);
    // Large top-k batches go in `pipe_chunks` pieces: launches are asynchronous, so the host builds the
    // query tables of piece i+1 while the GPU scores piece i.
    if (topk && c->pipe_chunks > 1 && nq >= 256u * c->pipe_chunks)
        rows_per_chunk = std::min<uint64_t>(rows_per_chunk, (nq + c->pipe_chunks - 1) / c->pipe_chunks);
    // a fixed workspace is never grown: the batch goes in as many chunks as its score rows need
    if (w.fixed) rows_per_chunk = std::min<uint64_t>(rows_per_chunk, w.scores_cap / row_bytes);
    if (rows_per_chunk == 0) rows_per_chunk = 1;                    // ensure() below reports the shortfall
    if ((rc = ensure(w, &w.d_scores, &w.scores_cap, (size_t)(rows_per_chunk * row_bytes), "score matrix")) != FZ_OK) return rc;
    uint8_t *d_orig_rows = w.d_scores + rows_per_chunk * row_stride * esz;   // [rows][n] in original order (*_ALL modes)

    for (uint64_t q0 = qa; q0 < qb; q0 += rows_per_chunk) {
        const uint32_t rows = (uint32_t)std::min<uint64_t>(rows_per_chunk, qb - q0);
        Prepared P;
        if ((rc = prepare_rows(c, view_kind, kmode, q_cps, q_offs, q0, rows, st, P)) != FZ_OK) return rc;
        hipEvent_t e0, e1, e2;
        if ((rc = ev.mark(&e0, st)) != FZ_OK) return rc;
        if ((rc = launch_rows(c, view_kind, kmode, ties, P, PH_ALL, w.d_scores, row_stride, nullptr, st)) != FZ_OK) return rc;
        if ((rc = ev.mark(&e1, st)) != FZ_OK) return rc;
        ev.score.push_back({e0, e1});
        c->timing.pairs += (uint64_t)rows * c->n;

        if (topk) {
            FzSelectParams sp{};
            sp.scores = w.d_scores; sp.row_stride = row_stride; sp.n = (uint32_t)c->n; sp.n_orig = (uint32_t)c->n;
            sp.orig_of_phys = v.d_orig_of_phys; sp.phys_of_orig = v.d_phys_of_orig;
            sp.min_score = t.min_score; sp.k = t.k; sp.base_index = c->base;
            sp.out_keys = t.d_keys + q0 * t.k; sp.out_cnt = t.d_cnt ? t.d_cnt + q0 : nullptr;
            sp.flat = t.flat;
            // a few rows over a large shard (single-query latency: reference src/synapse_agent.py:517-518 asks one entity at
            // a time): one block would walk the whole row, so the row is cut into runs, one block each, and the block that
            // finishes last merges the row's keys
            if (!w.fixed && c->select_slice != 0u && rows <= c->select_slice_rows && c->n >= 2ull * c->select_slice && t.k != 0u && 2u * t.k <= FZ_MAX_K) {
                uint32_t S = 1u;                                  // a power of two: blocks per row
                while (2u * S * t.k <= FZ_MAX_K && (uint64_t)2u * S * c->select_slice <= c->n) S *= 2u;
                if (S > 1u) { sp.n_slices = S; sp.slice_keys = c->d_slice_keys; sp.slice_done = c->d_slice_done; }
            }
            struct DevFree { void operator()(uint64_t *p) const { if (p) (void)hipFree(p); } };
            std::unique_ptr<uint64_t, DevFree> dbg_holder;       // freed on the early returns below too
            uint64_t *d_dbg = nullptr;
            if (!w.fixed && getenv("FUZZYHIP_SELECT_DEBUG")) {
                HIP_TRY(hipMalloc((void **)&d_dbg, (size_t)rows * 64));
                dbg_holder.reset(d_dbg);
                HIP_TRY(hipMemsetAsync(d_dbg, 0, (size_t)rows * 64, st));
                sp.dbg = d_dbg;
            }
            HIP_TRY(fz_launch_select(sp, rows, st));
            if (d_dbg) {      // diagnostics only: phase times of the slowest rows (100 MHz wall clock)
                std::vector<uint64_t> h((size_t)rows * 8);
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
                dbg_holder.reset();
                uint32_t worst = 0;
                for (uint32_t r = 0; r < rows; ++r) if (h[r * 8 + 4] - h[r * 8] > h[worst * 8 + 4] - h[worst * 8]) worst = r;
                const uint64_t *x = &h[(size_t)worst * 8];
                fprintf(stderr, "[select dbg] slowest row %u: passA %.1f us, passB %.1f us, ties %.1f us, sort %.1f us; c_eq %llu need %llu sstar %llu\n",
                        worst, (x[1] - x[0]) / 100.0, (x[2] - x[1]) / 100.0, (x[3] - x[2]) / 100.0, (x[4] - x[3]) / 100.0,
                        (unsigned long long)x[5], (unsigned long long)x[6], (unsigned long long)x[7]);
            }
            c->timing.select_launches++;
            if ((rc = ev.mark(&e2, st)) != FZ_OK) return rc;
            ev.select.push_back({e1, e2});
        } else {
            // physical -> original order on the device (a gather through phys_of_orig), then one contiguous
            // copy straight into the caller's rows: no host-side permutation
            HIP_TRY(fz_launch_unpermute(w.d_scores, row_stride, d_orig_rows, c->n, v.d_phys_of_orig, (uint32_t)c->n, rows, (uint32_t)esz, st));
            uint8_t *dst = esz == 1 ? o.scores + (q0 * c->n) : reinterpret_cast<uint8_t *>(o.dists + q0 * c->n);
            HIP_TRY(hipMemcpyAsync(dst, d_orig_rows, (size_t)rows * c->n * esz, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    return FZ_OK;
}

// Lists that overflowed (more than FZ_EMIT_CAP candidates reached the sample's limit: hits clustered in index or length
// where the sample tiles happen not to fall).  Up to FZ_FUSED_REDO_ROWS such rows are redone one by one through the
// score matrix -- a row writes its own keys, the others stay --, more than that chunk by chunk.  Results are the
// same either way.  The records are host copies taken after the call's kernels (read_overflow, or a submit's pinned copy).
static int read_overflow(const uint32_t *d_ovf, const uint32_t *d_ovf_rows, uint32_t nchunks, std::vector<uint32_t> &h_ovf, std::vector<uint32_t> &h_rows)
{
    h_ovf.assign(FZ_FUSED_MAX_CHUNKS, 0u); h_rows.assign(1u + FZ_FUSED_REDO_ROWS, 0u);
    HIP_TRY(hipMemcpy(h_ovf.data(), d_ovf, (size_t)nchunks * 4, hipMemcpyDeviceToHost));
    bool any = false;
    for (uint32_t i = 0; i < nchunks; ++i) any |= h_ovf[i] != 0u;
    if (any) HIP_TRY(hipMemcpy(h_rows.data(), d_ovf_rows, h_rows.size() * 4, hipMemcpyDeviceToHost));
    return FZ_OK;
}

static int redo_overflow(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq, int view_kind, int kmode,
                         FzTies ties, const TopkTarget &t, RunMode mode, RunOut &o, CallEvents &ev, hipStream_t st,
                         const uint32_t *h_ovf_p, const uint32_t *h_rows_p, uint32_t nchunks, uint64_t rows_per_chunk)
{
    int rc;
    std::vector<uint32_t> h_ovf(h_ovf_p, h_ovf_p + nchunks), h_rows(h_rows_p, h_rows_p + 1u + FZ_FUSED_REDO_ROWS);
    bool any = false;
    for (uint32_t x : h_ovf) any |= x != 0u;
    if (!any) return FZ_OK;
    for (uint32_t x : h_ovf) c->timing.fused_fallback_chunks += x != 0u ? 1u : 0u;
    if (h_rows[0] <= FZ_FUSED_REDO_ROWS) {
        std::sort(h_rows.begin() + 1, h_rows.begin() + 1 + h_rows[0]);
        if (getenv("FUZZYHIP_TRACE")) fprintf(stderr, "fuzzyhip: %u rows overflowed their lists; redone one by one through the score matrix\n", h_rows[0]);
        c->timing.fused_fallback_rows = h_rows[0];
        for (uint32_t i = 0; i < h_rows[0]; ++i) {
            const uint64_t r = h_rows[1u + i];
            if (r >= nq) return fail(FZ_ERR_INVALID, "internal: overflowed row %llu of %u", (unsigned long long)r, nq);
            if ((rc = run_matrix(c, q_cps, q_offs, r, r + 1, view_kind, kmode, ties, mode, o, t, ev, st)) != FZ_OK) return rc;
        }
        return FZ_OK;
    }
    uint32_t ci = 0;
    for (uint64_t q0 = 0; q0 < nq; q0 += rows_per_chunk, ++ci) {
        if (h_ovf[ci] == 0u) continue;
        if (getenv("FUZZYHIP_TRACE")) fprintf(stderr, "fuzzyhip: %u rows of chunk %u overflowed their lists; chunk redone through the score matrix\n", h_ovf[ci], ci);
        c->timing.fused_fallback_rows += h_ovf[ci];
        const uint64_t qb = std::min<uint64_t>(nq, q0 + rows_per_chunk);
        if ((rc = run_matrix(c, q_cps, q_offs, q0, qb, view_kind, kmode, ties, mode, o, t, ev, st)) != FZ_OK) return rc;
    }
    return FZ_OK;
}

// Fused threshold + top-k (FzEmit, fz_common.h): sample tiles -> small score matrix -> fz_select_topk fixes
// each row's limit -> the other tiles append the keys that reach it -> fz_merge_lists.  No score matrix
// over the whole shard, so a batch is not cut to fit one.  A chunk whose lists overflowed is redone
// through run_matrix at the end (results identical either way).
hipError_t fz_launch_fold_status(const uint32_t *ovf, uint32_t n, uint64_t *status, hipStream_t st);

static int run_fused(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                     int view_kind, int kmode, FzTies ties, const TopkTarget &t, RunMode mode, RunOut &o,
                     CallEvents &ev, hipStream_t st)
{
    const View &v = c->views[view_kind];
    Workspace &w = *c->ws;
    int rc;
    const uint64_t ns_pad = (uint64_t)v.n_sample_tiles * FZ_TILE;
    const uint32_t cap = FZ_EMIT_CAP;
    auto a256 = [](uint64_t x) { return (x + 255ull) & ~255ull; };
    // arena (the workspace's score buffer): [overflow counters][per chunk: limits | counts | lists | sample scores]
    const uint64_t ovf_bytes = a256((uint64_t)FZ_FUSED_MAX_CHUNKS * 4) + a256((uint64_t)(1u + FZ_FUSED_REDO_ROWS) * 4);   // chunk counters | overflowed rows
    const uint64_t per_row = ns_pad + (uint64_t)cap * 8 + 8 + 4;
    const uint64_t budget = w.fixed ? w.scores_cap : std::max<uint64_t>(c->ws_limit, ovf_bytes + per_row + 1024);
    if (budget < ovf_bytes + per_row + 1024)
        return fail(FZ_ERR_LIMIT, "fz_query_topk_noalloc: one row of this corpus needs %llu bytes of the static scratch buffer, %llu are left",
                    (unsigned long long)(per_row + 1024), (unsigned long long)(budget > ovf_bytes ? budget - ovf_bytes : 0));
    uint64_t rows_per_chunk = std::min<uint64_t>(nq, (budget - ovf_bytes - 1024) / per_row);
    if (c->pipe_chunks > 1 && nq >= 256u * c->pipe_chunks)
        rows_per_chunk = std::min<uint64_t>(rows_per_chunk, (nq + c->pipe_chunks - 1) / c->pipe_chunks);
    rows_per_chunk = std::max<uint64_t>(rows_per_chunk, (nq + FZ_FUSED_MAX_CHUNKS - 1) / FZ_FUSED_MAX_CHUNKS);
    const uint64_t o_cnt = ovf_bytes, o_lim = o_cnt + a256(rows_per_chunk * 4), o_list = o_lim + a256(rows_per_chunk * 8);
    const uint64_t o_sc = o_list + a256(rows_per_chunk * cap * 8);
    const uint64_t arena = o_sc + a256(rows_per_chunk * ns_pad);
    if ((rc = ensure(w, &w.d_scores, &w.scores_cap, (size_t)arena, "fused top-k workspace")) != FZ_OK) return rc;
    uint32_t *d_ovf = reinterpret_cast<uint32_t *>(w.d_scores);
    uint32_t *d_ovf_rows = reinterpret_cast<uint32_t *>(w.d_scores + a256((uint64_t)FZ_FUSED_MAX_CHUNKS * 4));
    uint64_t *d_lim = reinterpret_cast<uint64_t *>(w.d_scores + o_lim);
    uint32_t *d_ecnt = reinterpret_cast<uint32_t *>(w.d_scores + o_cnt);
    uint64_t *d_list = reinterpret_cast<uint64_t *>(w.d_scores + o_list);
    uint8_t *d_sc = w.d_scores + o_sc;
    const uint32_t nchunks = (uint32_t)((nq + rows_per_chunk - 1) / rows_per_chunk);
    // one fill: the overflow records and, right behind them, the first chunk's list counts
    HIP_TRY(hipMemsetAsync(d_ovf, 0, (size_t)(ovf_bytes + std::min<uint64_t>(rows_per_chunk, nq) * 4), st));

    uint32_t ci = 0;
    for (uint64_t q0 = 0; q0 < nq; q0 += rows_per_chunk, ++ci) {
        const uint32_t rows = (uint32_t)std::min<uint64_t>(rows_per_chunk, nq - q0);
        Prepared P;
        if ((rc = prepare_rows(c, view_kind, kmode, q_cps, q_offs, q0, rows, st, P)) != FZ_OK) return rc;
        hipEvent_t e0, e1, e2, e3, e4;
        if ((rc = ev.mark(&e0, st)) != FZ_OK) return rc;
        if ((rc = launch_rows(c, view_kind, kmode, ties, P, PH_SAMPLE, d_sc, ns_pad, nullptr, st)) != FZ_OK) return rc;
        if ((rc = ev.mark(&e1, st)) != FZ_OK) return rc;
        FzSelectParams sp{};
        sp.scores = d_sc; sp.row_stride = ns_pad; sp.n = (uint32_t)ns_pad; sp.n_orig = (uint32_t)c->n;
        sp.orig_of_phys = v.d_orig_of_phys; sp.phys_of_orig = v.d_phys_of_orig;
        sp.min_score = t.min_score; sp.k = t.k; sp.base_index = c->base;
        sp.out_keys = t.d_keys + q0 * t.k; sp.out_cnt = nullptr; sp.flat = 0; sp.out_lim = d_lim; sp.walk = v.d_sample_walk;
        HIP_TRY(fz_launch_select(sp, rows, st));
        if (ci != 0) HIP_TRY(hipMemsetAsync(d_ecnt, 0, (size_t)rows * 4, st));
        if ((rc = ev.mark(&e2, st)) != FZ_OK) return rc;
        FzEmit em{};
        em.lim = d_lim; em.list = d_list; em.cnt = d_ecnt; em.cap = cap;
        em.orig_of_phys = v.d_orig_of_phys; em.base_index = c->base;
        if ((rc = launch_rows(c, view_kind, kmode, ties, P, PH_REST, nullptr, 0, &em, st)) != FZ_OK) return rc;
        if ((rc = ev.mark(&e3, st)) != FZ_OK) return rc;
        FzMergeParams mp{};
        mp.sample_keys = t.d_keys + q0 * t.k; mp.list = d_list; mp.cnt = d_ecnt; mp.cap = cap; mp.k = t.k;
        mp.out_keys = t.d_keys + q0 * t.k; mp.out_cnt = t.d_cnt ? t.d_cnt + q0 : nullptr; mp.overflow = d_ovf + ci;
        mp.ovf_rows = d_ovf_rows; mp.ovf_rows_cap = FZ_FUSED_REDO_ROWS; mp.row_base = (uint32_t)q0;
        HIP_TRY(fz_launch_merge(mp, rows, st));
        if ((rc = ev.mark(&e4, st)) != FZ_OK) return rc;
        ev.score.push_back({e0, e1}); ev.score.push_back({e2, e3});
        ev.select.push_back({e1, e2}); ev.select.push_back({e3, e4});
        c->timing.select_launches += 2;
        c->timing.pairs += (uint64_t)rows * c->n;
    }
    if (mode == RUN_TOPK_DEVICE && !o.sync) {
        // asynchronous caller: no host look at the counters; they are folded into its status word on the stream, and the
        // caller repairs the overflowed rows after its own synchronisation (redo_async_overflow)
        HIP_TRY(fz_launch_fold_status(d_ovf, nchunks, o.d_status, st));
        fz_corpus::AsyncFused &a = c->async_fused;
        a.valid = true; a.d_ovf = d_ovf; a.d_ovf_rows = d_ovf_rows; a.nchunks = nchunks; a.rows_per_chunk = rows_per_chunk;
        a.view_kind = view_kind; a.kmode = kmode; a.ties = ties;
        a.d_keys = t.d_keys; a.d_cnt = t.d_cnt; a.min_score = t.min_score; a.k = t.k; a.flat = t.flat;
        return FZ_OK;
    }
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> h_ovf, h_rows;
    if ((rc = read_overflow(d_ovf, d_ovf_rows, nchunks, h_ovf, h_rows)) != FZ_OK) return rc;
    return redo_overflow(c, q_cps, q_offs, nq, view_kind, kmode, ties, t, mode, o, ev, st, h_ovf.data(), h_rows.data(), nchunks, rows_per_chunk);
}

// c->mu held by the caller
static int run_queries_locked(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                              int view_kind, int kmode, FzTies ties, int32_t min_score, uint32_t k,
                              RunMode mode, RunOut &o, Workspace *fixed_ws = nullptr)
{
    if (nq && (!q_offs || (!q_cps && q_offs[nq] != q_offs[0]))) return fail(FZ_ERR_INVALID, "null query buffers");
    const bool topk = mode == RUN_TOPK_HOST || mode == RUN_TOPK_DEVICE;
    if (topk && k > FZ_MAX_K) return fail(FZ_ERR_LIMIT, "k=%u exceeds FZ_MAX_K=%u", k, FZ_MAX_K);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (fixed_ws && !c->views[view_kind].built)
        return fail(FZ_ERR_INVALID, "fz_query_topk_noalloc: the device view of this scorer is not built yet (building it allocates); "
                                    "one fz_query_topk call with the scorer builds it");
    if ((rc = build_view(c, view_kind)) != FZ_OK) return rc;
    const View &v = c->views[view_kind];
    hipStream_t st = c->stream;
    if (mode == RUN_TOPK_DEVICE && !o.sync && o.d_status) HIP_TRY(hipMemsetAsync(o.d_status, 0, 8, st));
    struct WsSwitch {                                    // the corpus works in `fixed_ws` for this call only
        fz_corpus *c; Workspace *prev;
        WsSwitch(fz_corpus *c_, Workspace *f) : c(c_), prev(c_->ws) { if (f) c->ws = f; }
        ~WsSwitch() { c->ws = prev; }
    } ws_switch(c, fixed_ws);
    Workspace &w = *c->ws;
    c->timing = fz_timing{};
    c->pending.valid = false;
    c->async_fused.valid = false;
    ++c->call_seq;

    const bool nothing = topk && (k == 0 || c->n == 0 || min_score > 100);
    uint64_t *d_keys = nullptr; uint32_t *d_cnt = nullptr;
    if (topk && nq && k) {
        if (mode == RUN_TOPK_DEVICE) { d_keys = o.d_keys; d_cnt = o.d_cnt; }
        else {
            if ((rc = ensure(w, &w.d_keys, &w.keys_cap, (size_t)nq * k * 8, "top-k keys")) != FZ_OK) return rc;
            if ((rc = ensure(w, &w.d_cnt, &w.cnt_cap, (size_t)nq * 4, "result counts")) != FZ_OK) return rc;
            d_keys = w.d_keys; d_cnt = w.d_cnt;
        }
    }
    if (nq == 0) return FZ_OK;
    if (nothing) {
        if (mode == RUN_TOPK_HOST) {
            if (o.cnt) memset(o.cnt, 0, (size_t)nq * 4);
            if (k) { memset(o.idx, 0, (size_t)nq * k * 8); memset(o.score, 0, (size_t)nq * k * 4); }
        } else if (k) {
            HIP_TRY(hipMemsetAsync(d_keys, 0, (size_t)nq * k * 8, st));
            if (d_cnt) HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)nq * 4, st));
            if (o.sync) HIP_TRY(hipStreamSynchronize(st));
        }
        return FZ_OK;
    }
    if (c->n == 0) return FZ_OK;                                  // *_ALL of an empty corpus

    CallEvents ev{c, w.fixed};
    hipEvent_t ev_first, ev_last;
    if ((rc = ev.mark(&ev_first, st)) != FZ_OK) return rc;
    const TopkTarget t{d_keys, d_cnt, min_score, k, o.flat};
    // fused threshold + top-k when the view keeps a sample in front (large shards), the order is by score,
    // and k is small enough for the lists; the score matrix otherwise
    const bool fused = topk && c->use_fused && !o.force_matrix && !o.flat && k <= FZ_FUSED_MAX_K && v.n_sample_tiles < v.ntiles &&
                       (uint64_t)nq * v.npad >= c->fused_min_bytes && (mode == RUN_TOPK_HOST || o.sync || o.d_status);
    if (fused && getenv("FUZZYHIP_TRACE")) fprintf(stderr, "fuzzyhip: fused top-k over %u rows (%u of %u tiles sampled)\n", nq, v.n_sample_tiles, v.ntiles);
    if (fused) rc = run_fused(c, q_cps, q_offs, nq, view_kind, kmode, ties, t, mode, o, ev, st);
    else rc = run_matrix(c, q_cps, q_offs, 0, nq, view_kind, kmode, ties, mode, o, t, ev, st);
    if (rc != FZ_OK) return rc;
    c->timing.fused = fused ? 1u : 0u;

    if (mode == RUN_TOPK_HOST) {
        if ((rc = ensure_pinned(w, &c->h_keys, &c->h_keys_cap, (size_t)nq * k * 8, "top-k keys")) != FZ_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->h_keys, d_keys, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
        if ((rc = ev.mark(&ev_last, st)) != FZ_OK) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        decode_keys(c->h_keys, nq, k, o.idx, o.score, o.cnt);
    } else {
        if (mode == RUN_TOPK_DEVICE && !o.sync) {
            // the kernel times are read by finish_pending_timing() once the caller has synchronised the stream
            if (!w.fixed && (rc = ev.mark(&ev_last, st)) == FZ_OK) {
                c->pending.valid = true; c->pending.score = ev.score; c->pending.select = ev.select;
                c->pending.first = ev_first; c->pending.last = ev_last;
            }
            return FZ_OK;
        }
        if ((rc = ev.mark(&ev_last, st)) != FZ_OK) return rc;
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (auto &s : ev.score) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.score_ms += ms; }
    for (auto &s : ev.select) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.select_ms += ms; }
    { float ms = 0; if (hipEventElapsedTime(&ms, ev_first, ev_last) == hipSuccess) c->timing.total_ms = ms; }
    return FZ_OK;
}

static int run_queries(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                       int view_kind, int kmode, FzTies ties, int32_t min_score, uint32_t k,
                       RunMode mode, RunOut &o, Workspace *fixed_ws = nullptr)
{
    if (!c) return fail(FZ_ERR_INVALID, "null corpus");
    std::lock_guard<std::mutex> lk(c->mu);
    return run_queries_locked(c, q_cps, q_offs, nq, view_kind, kmode, ties, min_score, k, mode, o, fixed_ws);
}

// The rows of the last asynchronous fused call whose lists overflowed, redone (synchronously) into the call's key
// buffer; the stream has been synchronised since the call and nothing else has used the corpus (c->mu held throughout).
static int redo_async_overflow(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq)
{
    fz_corpus::AsyncFused &a = c->async_fused;
    if (!a.valid) return FZ_OK;
    a.valid = false;
    HIP_TRY(hipSetDevice(c->device));
    CallEvents ev{c, false};
    RunOut o; o.d_keys = a.d_keys; o.d_cnt = a.d_cnt; o.sync = 1; o.flat = a.flat;
    const TopkTarget t{a.d_keys, a.d_cnt, a.min_score, a.k, a.flat};
    std::vector<uint32_t> h_ovf, h_rows;
    int rc = read_overflow(a.d_ovf, a.d_ovf_rows, a.nchunks, h_ovf, h_rows);
    if (rc != FZ_OK) return rc;
    rc = redo_overflow(c, q_cps, q_offs, nq, a.view_kind, a.kmode, a.ties, t, RUN_TOPK_DEVICE, o, ev, c->stream,
                       h_ovf.data(), h_rows.data(), a.nchunks, a.rows_per_chunk);
    if (rc != FZ_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &s : ev.score) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.score_ms += ms; }
    for (auto &s : ev.select) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.select_ms += ms; }
    return FZ_OK;
}

// kernel times of the last asynchronous call (its stream must have been synchronised since); c->mu held
static void finish_pending_timing(fz_corpus *c)
{
    if (!c->pending.valid) return;
    for (auto &s : c->pending.score) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.score_ms += ms; }
    for (auto &s : c->pending.select) { float ms = 0; if (hipEventElapsedTime(&ms, s.first, s.second) == hipSuccess) c->timing.select_ms += ms; }
    { float ms = 0; if (hipEventElapsedTime(&ms, c->pending.first, c->pending.last) == hipSuccess) c->timing.total_ms = ms; }
    c->pending.valid = false;
}

static int fz_query_topk_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                             int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                             int32_t *out_score, uint32_t *out_cnt)
{
    if (nq && k && (!out_idx || !out_score)) return fail(FZ_ERR_INVALID, "null output buffers");
    int view, kmode; FzTies ties; int rc;
    if ((rc = scorer_plan(scorer & 0xFF, &view, &kmode, &ties)) != FZ_OK) return rc;
    RunOut o; o.idx = out_idx; o.score = out_score; o.cnt = out_cnt;
    o.flat = (scorer & FZ_ORDER_BY_INDEX) ? 1u : 0u;
    return run_queries(c, q_cps, q_offs, nq, view, kmode, ties, min_score, k, RUN_TOPK_HOST, o);
}

// ---- fz_query_topk in two phases -------------------------------------------------------------------------------
// submit: query tables, upload, every kernel, the copy of keys + status word + overflow records to pinned memory and
// an event, all enqueued on the corpus' stream; nothing waits for the GPU.  wait: the event, then (rarely) the rows whose
// fused lists overflowed redone through the score matrix, then the keys decoded.  One stream carries every call of
// a corpus, so the workspace a later call reuses is only touched after the earlier call's copies have run; what a
// call must keep until its wait lives in its slot (c->flight).  The host work of call i+1 -- 0.13 ms of table
// building per 1000 queries, the Python side, the decoding of call i-1 -- then passes under the kernels of call i.
// enqueue everything of a submitted call: upload, kernels, the copy of keys + status (+ overflow records), the event
static int launch_flight(fz_corpus *c, fz_corpus::Flight &f)
{
    int rc;
    const size_t nkeys = (size_t)f.nq * f.k, words = nkeys + 1;
    hipStream_t st = c->stream;
    f.fused = false;
    if (!f.empty) {
        RunOut o; o.d_keys = f.d_keys; o.d_cnt = nullptr; o.sync = 0; o.d_status = f.d_keys + nkeys; o.flat = f.flat;
        HIP_TRY(hipMemsetAsync(f.d_keys + nkeys, 0, 8, st));          // (a call that takes the matrix path never writes it)
        if ((rc = run_queries_locked(c, f.q_cps, f.q_offs, f.nq, f.view_kind, f.kmode, f.ties, f.min_score, f.k, RUN_TOPK_DEVICE, o)) != FZ_OK) return rc;
        st = c->stream;
        fz_corpus::AsyncFused &a = c->async_fused;
        if (a.valid) {                                               // the overflow records leave the workspace with the call
            f.fused = true; f.nchunks = a.nchunks; f.rows_per_chunk = a.rows_per_chunk;
            HIP_TRY(hipMemcpyAsync(f.h_ovf, a.d_ovf, (size_t)a.nchunks * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(f.h_ovf + FZ_FUSED_MAX_CHUNKS, a.d_ovf_rows, (size_t)(1u + FZ_FUSED_REDO_ROWS) * 4, hipMemcpyDeviceToHost, st));
            a.valid = false;
        }
        HIP_TRY(hipMemcpyAsync(f.h_keys, f.d_keys, words * 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(f.done, st));
    f.seq = ++c->flight_seq;
    f.launched = true;
    return FZ_OK;
}

static int fz_query_topk_submit_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                                     int scorer, int32_t min_score, uint32_t k, int *ticket)
{
    if (!c || !ticket) return fail(FZ_ERR_INVALID, "null corpus or ticket");
    *ticket = -1;
    int view, kmode; FzTies ties; int rc;
    if ((rc = scorer_plan(scorer & 0xFF, &view, &kmode, &ties)) != FZ_OK) return rc;
    if (k > FZ_MAX_K) return fail(FZ_ERR_LIMIT, "k=%u exceeds FZ_MAX_K=%u", k, FZ_MAX_K);
    if (nq && (!q_offs || (!q_cps && q_offs[nq] != q_offs[0]))) return fail(FZ_ERR_INVALID, "null query buffers");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int si = -1;
    bool gpu_busy = false;                                           // a call of this corpus launched and not fetched yet
    for (int i = 0; i < (int)FZ_MAX_IN_FLIGHT; ++i) {
        if (!c->flight[i].busy) { if (si < 0) si = i; }
        else gpu_busy |= c->flight[i].launched;
    }
    if (si < 0) return fail(FZ_ERR_LIMIT, "fz_query_topk_submit: %u calls of this corpus are in flight already; fetch one with fz_query_topk_wait first", FZ_MAX_IN_FLIGHT);
    fz_corpus::Flight &f = c->flight[si];
    if (!f.done) HIP_TRY(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    if (!f.h_ovf) HIP_TRY(hipHostMalloc((void **)&f.h_ovf, (size_t)(FZ_FUSED_MAX_CHUNKS + 1u + FZ_FUSED_REDO_ROWS) * 4, hipHostMallocDefault));
    const size_t nkeys = (size_t)nq * k, words = nkeys + 1;
    if (f.d_words < words) {
        if (f.d_keys) { HIP_TRY(hipFree(f.d_keys)); f.d_keys = nullptr; f.d_words = 0; }
        HIP_TRY(hipMalloc((void **)&f.d_keys, words * 8));
        f.d_words = words;
    }
    if (f.h_words < words) {
        if (f.h_keys) { HIP_TRY(hipHostFree(f.h_keys)); f.h_keys = nullptr; f.h_words = 0; }
        HIP_TRY(hipHostMalloc((void **)&f.h_keys, words * 8, hipHostMallocDefault));
        f.h_words = words;
    }
    f.q_cps = q_cps; f.q_offs = q_offs; f.nq = nq; f.k = k; f.flat = (scorer & FZ_ORDER_BY_INDEX) ? 1u : 0u;
    f.view_kind = view; f.kmode = kmode; f.ties = ties; f.min_score = min_score;
    f.fused = false; f.launched = false; f.empty = nkeys == 0;
    if (gpu_busy && !c->submit_eager) {
        // The GPU works on the call before this one: only the host part happens now -- the query tables, built into a
        // pinned blob --, and the call is enqueued by the fz_query_topk_wait that sees its predecessor finish (the
        // first thing that wait does).  Enqueueing behind running kernels instead lets the GPU go from one call to
        // the next without the host, but the HIP runtime PyTorch wheels bundle (7.0) then stalls inside the launches
        // (3.15 ms per C2 step against 2.65 with ROCm 7.2's, 2.80 one call at a time): FUZZYHIP_SUBMIT_EAGER=1.
        if (!f.empty && c->views[view].built && c->n != 0 && min_score <= 100) {
            fz_corpus::Prebuilt &pb = c->prebuilt;
            pb.valid = false;
            pb.P = std::make_shared<Prepared>();
            if ((rc = build_tables(c, view, kmode, q_cps, q_offs, 0, nq, *pb.P)) != FZ_OK) return rc;
            pb.q_cps = q_cps; pb.q_offs = q_offs; pb.q0 = 0; pb.rows = nq; pb.view_kind = view; pb.kmode = kmode;
            pb.valid = true;
        }
    } else if ((rc = launch_flight(c, f)) != FZ_OK) return rc;
    f.busy = true;
    *ticket = si;
    return FZ_OK;
}

static int fz_query_topk_wait_impl(fz_corpus *c, int ticket, uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt)
{
    if (!c) return fail(FZ_ERR_INVALID, "null corpus");
    if (ticket < 0 || ticket >= (int)FZ_MAX_IN_FLIGHT) return fail(FZ_ERR_INVALID, "fz_query_topk_wait: no such ticket (%d)", ticket);
    fz_corpus::Flight &f = c->flight[ticket];
    struct Release { fz_corpus::Flight &f; bool armed = false; ~Release() { if (armed) f.busy = false; } } release{f};     // on every way out
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!f.busy) return fail(FZ_ERR_INVALID, "fz_query_topk_wait: ticket %d is not in flight", ticket);
        if (f.nq && f.k && (!out_idx || !out_score)) return fail(FZ_ERR_INVALID, "null output buffers");
        release.armed = true;
        if (!f.launched) {                                           // fetched before its predecessor: it goes behind it now
            HIP_TRY(hipSetDevice(c->device));
            int rc = launch_flight(c, f);
            if (rc != FZ_OK) return rc;
        }
    }
    // the slot does not change while it is busy: the wait for the GPU runs outside the lock, so that another thread can
    // submit meanwhile
    hipError_t he = hipEventSynchronize(f.done);
    std::lock_guard<std::mutex> lk(c->mu);
    if (he != hipSuccess) return fail(FZ_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(he));
    HIP_TRY(hipSetDevice(c->device));
    // the GPU is free: the call that was waiting with its tables built goes out before this one's result is decoded
    int deferred_rc = FZ_OK; std::string deferred_err;
    for (fz_corpus::Flight &g : c->flight)
        if (&g != &f && g.busy && !g.launched && (deferred_rc = launch_flight(c, g)) != FZ_OK) { deferred_err = g_err; g.busy = false; }
    if (f.empty) {
        if (out_cnt && f.nq) memset(out_cnt, 0, (size_t)f.nq * 4);
    } else {
        const size_t nkeys = (size_t)f.nq * f.k;
        if (f.seq == c->flight_seq) finish_pending_timing(c);         // the kernel times of the last call launched
        if (f.fused && f.h_keys[nkeys] != 0ull) {
            // rows whose lists overflowed: redone now, behind whatever was launched since (stream order), into the slot's keys
            CallEvents ev{c, false};
            ev.next = 48;                                             // clear of the events a call in flight still needs
            RunOut o; o.d_keys = f.d_keys; o.d_cnt = nullptr; o.sync = 1; o.flat = f.flat;
            const TopkTarget t{f.d_keys, nullptr, f.min_score, f.k, f.flat};
            int rc = redo_overflow(c, f.q_cps, f.q_offs, f.nq, f.view_kind, f.kmode, f.ties, t, RUN_TOPK_DEVICE, o, ev, c->stream,
                                   f.h_ovf, f.h_ovf + FZ_FUSED_MAX_CHUNKS, f.nchunks, f.rows_per_chunk);
            if (rc != FZ_OK) return rc;
            HIP_TRY(hipMemcpyAsync(f.h_keys, f.d_keys, nkeys * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        decode_keys(f.h_keys, f.nq, f.k, out_idx, out_score, out_cnt);
    }
    if (deferred_rc != FZ_OK) return fail(deferred_rc, "the call submitted after this one could not be launched (its ticket is void): %s", deferred_err.c_str());
    return FZ_OK;
}

// fz_query_topk over the static scratch array: nothing is allocated (see include/fuzzyhip.h).
static std::mutex g_scratch_mu[64];

static int fz_query_topk_noalloc_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                                      int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                                      int32_t *out_score, uint32_t *out_cnt)
{
    if (!c) return fail(FZ_ERR_INVALID, "null corpus");
    if (nq && k && (!out_idx || !out_score)) return fail(FZ_ERR_INVALID, "null output buffers");
    int view, kmode; FzTies ties; int rc;
    if ((rc = scorer_plan(scorer & 0xFF, &view, &kmode, &ties)) != FZ_OK) return rc;
    if (nq > 32768u) return fail(FZ_ERR_LIMIT, "fz_query_topk_noalloc: nq=%u exceeds 32768 queries per call", nq);
    if (c->device < 0 || c->device >= 64) return fail(FZ_ERR_INVALID, "device %d out of range", c->device);
    // every call on this GPU works in the same array: one at a time
    std::lock_guard<std::mutex> dl(g_scratch_mu[c->device]);
    void *base = nullptr; size_t bytes = 0;
    HIP_TRY(fz_scratch_acquire(c->device, &base, &bytes));       // cached since the corpus was created
    // carve-up (every part a multiple of 256 bytes): counters | keys | counts | query tables | score rows
    uint8_t *p = static_cast<uint8_t *>(base);
    Workspace w;
    w.fixed = true;
    w.d_counters = reinterpret_cast<uint32_t *>(p);
    p += (size_t)FZ_COUNTER_SLOTS * FZ_STREAM_QUEUES * FZ_STREAM_QPAD * sizeof(uint32_t);
    w.d_keys = reinterpret_cast<uint64_t *>(p); w.keys_cap = FZ_NOALLOC_KEYS_BYTES; p += FZ_NOALLOC_KEYS_BYTES;
    w.d_cnt = reinterpret_cast<uint32_t *>(p); w.cnt_cap = 32768u * 4u; p += 32768u * 4u;
    w.d_qbuf = p; w.qbuf_cap = FZ_NOALLOC_BLOB_BYTES; p += FZ_NOALLOC_BLOB_BYTES;
    w.d_scores = p; w.scores_cap = bytes - (size_t)(p - static_cast<uint8_t *>(base));
    RunOut o; o.idx = out_idx; o.score = out_score; o.cnt = out_cnt;
    o.flat = (scorer & FZ_ORDER_BY_INDEX) ? 1u : 0u;
    return run_queries(c, q_cps, q_offs, nq, view, kmode, ties, min_score, k, RUN_TOPK_HOST, o, &w);
}

static int fz_query_topk_keys_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs,
                                  uint32_t nq, int scorer, int32_t min_score, uint32_t k,
                                  uint64_t *out_keys_device, uint32_t *out_cnt_device, int sync)
{
    if (nq && k && !out_keys_device) return fail(FZ_ERR_INVALID, "null device key buffer");
    int view, kmode; FzTies ties; int rc;
    if ((rc = scorer_plan(scorer & 0xFF, &view, &kmode, &ties)) != FZ_OK) return rc;
    RunOut o; o.d_keys = out_keys_device; o.d_cnt = out_cnt_device; o.sync = sync;
    o.flat = (scorer & FZ_ORDER_BY_INDEX) ? 1u : 0u;
    return run_queries(c, q_cps, q_offs, nq, view, kmode, ties, min_score, k, RUN_TOPK_DEVICE, o);
}

static int fz_score_all_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                            int scorer, uint8_t *out_scores)
{
    if (nq && c && c->n && !out_scores) return fail(FZ_ERR_INVALID, "null output buffer");
    int view, kmode; FzTies ties; int rc;
    if ((rc = scorer_plan(scorer, &view, &kmode, &ties)) != FZ_OK) return rc;
    RunOut o; o.scores = out_scores;
    return run_queries(c, q_cps, q_offs, nq, view, kmode, ties, 0, 0, RUN_SCORE_ALL, o);
}

static int fz_distance_all_impl(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                               int kind, uint16_t *out_dist)
{
    if (kind < 0 || kind > 4) return fail(FZ_ERR_INVALID, "kind must be 0 (InDel), 1 (Levenshtein), 2 (common tokens), 3 (substring) or 4 (SequenceMatcher matches)");
    if (nq && c && c->n && !out_dist) return fail(FZ_ERR_INVALID, "null output buffer");
    if (kind == 4 && nq && q_offs)
        for (uint32_t i = 0; i < nq; ++i)
            if (q_offs[i + 1] - q_offs[i] > 64u) return fail(FZ_ERR_TOO_LONG, "kind 4 (SequenceMatcher matches) takes queries of at most 64 code points");
    RunOut o; o.dists = out_dist;
    static const int kModes[5] = {FZ_MODE_INDEL, FZ_MODE_LEV, FZ_MODE_NHITS, FZ_MODE_FIND, FZ_MODE_RO};
    return run_queries(c, q_cps, q_offs, nq, kind == 2 ? VIEW_SET : VIEW_RAW, kModes[kind],
                       FzTies{0, 0}, 0, 0, RUN_DIST_ALL, o);
}

// Two scorers over two corpora that hold the SAME candidate list in two string forms (e.g. the
// normalised strings for partial_ratio and their full_process-ed forms for token_set_ratio),
// combined per candidate through the caller's 101 x 101 rank table, then top-k on the rank.
static int fz_query_topk_pair_impl(fz_corpus *ca, int scorer_a, const uint32_t *qa_cps, const uint64_t *qa_offs,
                                  fz_corpus *cb, int scorer_b, const uint32_t *qb_cps, const uint64_t *qb_offs,
                                  uint32_t nq, const uint16_t *rank_tab, uint32_t min_rank, uint32_t k,
                                  uint64_t *out_idx, int32_t *out_rank, uint32_t *out_cnt)
{
    if (!ca || !cb || ca == cb) return fail(FZ_ERR_INVALID, "two distinct corpora are required");
    if (ca->n != cb->n || ca->device != cb->device || ca->base != cb->base)
        return fail(FZ_ERR_INVALID, "the two corpora must hold the same candidates (n, base index) on one device");
    if (k > FZ_MAX_K) return fail(FZ_ERR_LIMIT, "k=%u exceeds FZ_MAX_K=%u", k, FZ_MAX_K);
    if (nq && (!qa_offs || !qb_offs || !rank_tab)) return fail(FZ_ERR_INVALID, "null argument");
    if (nq && k && (!out_idx || !out_rank)) return fail(FZ_ERR_INVALID, "null output buffers");
    if (min_rank > 0xFFFFu) min_rank = 0x10000u;             // nothing can pass
    int va, ma, vb, mb, rc; FzTies ta, tb;
    if ((rc = scorer_plan(scorer_a, &va, &ma, &ta)) != FZ_OK) return rc;
    if ((rc = scorer_plan(scorer_b, &vb, &mb, &tb)) != FZ_OK) return rc;
    if (nq == 0) return FZ_OK;
    if (out_cnt) memset(out_cnt, 0, (size_t)nq * 4);
    if (k) { memset(out_idx, 0, (size_t)nq * k * 8); memset(out_rank, 0, (size_t)nq * k * 4); }
    if (k == 0 || ca->n == 0 || min_rank > 0xFFFFu) return FZ_OK;

    // lock both, lowest address first
    fz_corpus *l1 = ca < cb ? ca : cb, *l2 = ca < cb ? cb : ca;
    std::lock_guard<std::mutex> lk1(l1->mu);
    std::lock_guard<std::mutex> lk2(l2->mu);
    HIP_TRY(hipSetDevice(ca->device));
    if ((rc = build_view(ca, va)) != FZ_OK) return rc;
    if ((rc = build_view(cb, vb)) != FZ_OK) return rc;
    const View &A = ca->views[va], &B = cb->views[vb];
    hipStream_t st = ca->stream;
    const uint64_t n = ca->n, stride_r = (n + 7) & ~7ull;
    uint64_t rows_per_chunk = std::max<uint64_t>(1, ca->ws_limit / (std::max<uint64_t>(A.npad, B.npad) + 2 * stride_r));
    rows_per_chunk = std::min<uint64_t>(rows_per_chunk, nq);
    Workspace &wa = ca->own_ws, &wb = cb->own_ws;
    if ((rc = ensure(wa, &wa.d_scores, &wa.scores_cap, (size_t)(rows_per_chunk * A.npad), "score matrix")) != FZ_OK) return rc;
    if ((rc = ensure(wb, &wb.d_scores, &wb.scores_cap, (size_t)(rows_per_chunk * B.npad), "score matrix")) != FZ_OK) return rc;
    if ((rc = ensure(wa, &ca->d_ranks, &ca->ranks_cap, (size_t)(rows_per_chunk * stride_r * 2), "rank matrix")) != FZ_OK) return rc;
    if ((rc = ensure(wa, &ca->d_rank_tab, &ca->rank_tab_cap, (size_t)FZ_RANK_DIM * FZ_RANK_DIM * 2, "rank table")) != FZ_OK) return rc;
    if ((rc = ensure(wa, &wa.d_keys, &wa.keys_cap, (size_t)nq * k * 8, "top-k keys")) != FZ_OK) return rc;
    if ((rc = ensure(wa, &wa.d_cnt, &wa.cnt_cap, (size_t)nq * 4, "result counts")) != FZ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ca->d_rank_tab, rank_tab, (size_t)FZ_RANK_DIM * FZ_RANK_DIM * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                       // rank_tab is the caller's pageable memory
    ca->timing = fz_timing{};
    cb->timing = fz_timing{};
    for (uint64_t q0 = 0; q0 < nq; q0 += rows_per_chunk) {
        const uint32_t rows = (uint32_t)std::min<uint64_t>(rows_per_chunk, nq - q0);
        if ((rc = score_rows(ca, va, ma, ta, qa_cps, qa_offs, q0, rows, A.npad, st)) != FZ_OK) return rc;
        if ((rc = score_rows(cb, vb, mb, tb, qb_cps, qb_offs, q0, rows, B.npad, st)) != FZ_OK) return rc;
        FzCombineParams cp{};
        cp.scores_a = wa.d_scores; cp.scores_b = wb.d_scores; cp.stride_a = A.npad; cp.stride_b = B.npad;
        cp.phys_a = A.d_phys_of_orig; cp.phys_b = B.d_phys_of_orig; cp.rank_tab = ca->d_rank_tab;
        cp.ranks = ca->d_ranks; cp.stride_r = stride_r; cp.n = (uint32_t)n;
        HIP_TRY(fz_launch_combine(cp, rows, st));
        FzSelect16Params sp{};
        sp.ranks = ca->d_ranks; sp.stride_r = stride_r; sp.n = (uint32_t)n; sp.min_rank = min_rank; sp.k = k;
        sp.base_index = ca->base; sp.out_keys = wa.d_keys + q0 * k; sp.out_cnt = wa.d_cnt + q0;
        HIP_TRY(fz_launch_select16(sp, rows, st));
        ca->timing.pairs += 2ull * rows * n;
    }
    std::vector<uint64_t> h_keys((size_t)nq * k);
    HIP_TRY(hipMemcpyAsync(h_keys.data(), wa.d_keys, h_keys.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    decode_keys(h_keys.data(), nq, k, out_idx, out_rank, out_cnt);
    return FZ_OK;
}

// Deterministic merge of per-shard key lists: plain descending sort of the n_lists*k keys of each
// query.  A key orders by (score desc, global index asc), so the result does not depend on how the
// corpus was sharded or on the order the lists were gathered in.
static int fz_merge_topk_impl(const uint64_t *keys, uint32_t n_lists, uint32_t nq, uint32_t k,
                             uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt)
{
    if ((uint64_t)n_lists * nq * k && (!keys || !out_idx || !out_score)) return fail(FZ_ERR_INVALID, "null argument");
    std::vector<uint64_t> merged((size_t)nq * k, 0ull);
    auto merge_range = [&](uint32_t qa, uint32_t qb) {
        std::vector<uint64_t> tmp;
        for (uint32_t q = qa; q < qb; ++q) {
            tmp.clear();
            for (uint32_t l = 0; l < n_lists; ++l)
                for (uint32_t i = 0; i < k; ++i) {
                    uint64_t key = keys[((uint64_t)l * nq + q) * k + i];
                    if (key) tmp.push_back(key);
                }
            const size_t keep = std::min<size_t>(k, tmp.size());          // only the first k are wanted
            std::partial_sort(tmp.begin(), tmp.begin() + keep, tmp.end(), [](uint64_t a, uint64_t b) { return a > b; });
            for (size_t i = 0; i < keep; ++i) merged[(size_t)q * k + i] = tmp[i];
        }
    };
    // a C3-sized exchange (10^4 queries x 8 shards x 10 keys) is ~15 ms on one host thread: queries are
    // independent, so large merges are cut over a few threads
    const uint64_t work = (uint64_t)n_lists * nq * k;
    uint32_t nthreads = work < 200000ull ? 1u : (uint32_t)std::min<uint64_t>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("FUZZYHIP_MERGE_THREADS")) nthreads = (uint32_t)std::min(64, std::max(1, atoi(e)));
    if (nthreads <= 1u) {
        merge_range(0, nq);
    } else {
        std::vector<std::thread> pool;
        const uint32_t per = (nq + nthreads - 1u) / nthreads;
        for (uint32_t t = 0; t < nthreads; ++t) {
            const uint32_t qa = std::min(nq, t * per), qb = std::min(nq, qa + per);
            if (qa < qb) pool.emplace_back(merge_range, qa, qb);
        }
        for (std::thread &t : pool) t.join();
    }
    if (nq && k) decode_keys(merged.data(), nq, k, out_idx, out_score, out_cnt);
    else if (out_cnt) memset(out_cnt, 0, (size_t)nq * 4);
    return FZ_OK;
}

// Explicit pairs: group the pairs by their `a` string; each distinct `a` becomes one query against
// a temporary raw view of its `b` strings, so the distances come out of the very kernels the
// scorers use (pattern = a, text = b).
static int fz_distance_batch_impl(int device, int kind, const uint32_t *a_cps, const uint64_t *a_offs,
                                 const uint32_t *b_cps, const uint64_t *b_offs, uint64_t n_pairs,
                                 uint32_t *out_dist)
{
    if (kind != 0 && kind != 1) return fail(FZ_ERR_INVALID, "kind must be 0 (InDel) or 1 (Levenshtein)");
    if (n_pairs == 0) return FZ_OK;
    if (!a_offs || !b_offs || !out_dist) return fail(FZ_ERR_INVALID, "null argument");
    std::map<std::u32string, std::vector<uint64_t>> groups;
    for (uint64_t i = 0; i < n_pairs; ++i) {
        if (a_offs[i + 1] < a_offs[i] || b_offs[i + 1] < b_offs[i]) return fail(FZ_ERR_INVALID, "offsets not monotone at %llu", (unsigned long long)i);
        groups[std::u32string(reinterpret_cast<const char32_t *>(a_cps) + a_offs[i], (size_t)(a_offs[i + 1] - a_offs[i]))].push_back(i);
    }
    for (auto &g : groups) {
        std::vector<uint32_t> cps; std::vector<uint64_t> offs{0};
        for (uint64_t i : g.second) {
            cps.insert(cps.end(), b_cps + b_offs[i], b_cps + b_offs[i + 1]);
            offs.push_back(cps.size());
        }
        if (cps.empty()) cps.push_back(0);
        fz_corpus *c = nullptr;
        int rc = corpus_create(cps.data(), offs.data(), g.second.size(), device, 0, VIEW_RAW, &c);
        if (rc != FZ_OK) return rc;
        struct CorpusDeleter { void operator()(fz_corpus *p) const { fz_corpus_destroy(p); } };
        std::unique_ptr<fz_corpus, CorpusDeleter> holder(c);
        std::vector<uint32_t> q(g.first.begin(), g.first.end());
        if (q.empty()) q.push_back(0);
        uint64_t qo[2] = {0, g.first.size()};
        std::vector<uint16_t> d(g.second.size());
        rc = fz_distance_all_impl(c, q.data(), qo, 1, kind, d.data());
        holder.reset();
        if (rc != FZ_OK) return rc;
        for (size_t j = 0; j < g.second.size(); ++j) out_dist[g.second[j]] = d[j];
    }
    return FZ_OK;
}

// ---------------------------------------------------------------------------------------------
// the exchange of the sharded path inside the library: RCCL all-gather + device merge (fuzzyhip.h)
// ---------------------------------------------------------------------------------------------
// librccl is opened at run time; the few declarations used are restated here (rccl.h: NCCL_UNIQUE_ID_BYTES
// 128, ncclSuccess 0, ncclUint64 5) so that neither the library nor its CPU sanitizer build needs the header.
namespace rccl {
struct UniqueId { char internal[FZ_COMM_ID_BYTES]; };
typedef struct ncclComm *Comm;
typedef int (*GetUniqueId)(UniqueId *);
typedef int (*CommInitRank)(Comm *, int, UniqueId, int);
typedef int (*CommDestroy)(Comm);
typedef int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t);
typedef const char *(*GetErrorString)(int);
constexpr int kUint64 = 5;
struct Api {
    void *handle = nullptr;
    GetUniqueId get_unique_id = nullptr; CommInitRank comm_init_rank = nullptr; CommDestroy comm_destroy = nullptr;
    AllGather all_gather = nullptr; GetErrorString get_error_string = nullptr;
};
static Api g_api;
static std::mutex g_api_mu;

// directory of the HIP runtime THIS library is bound to (the object that defines hipGetDeviceCount for us).
// A process may map a second copy (PyTorch wheels ship their own libamdhip64 / libhsa-runtime64 / librccl): RCCL
// must be the one built against the runtime we call, so it is looked for next to that runtime first.
static std::string hip_runtime_path()
{
    Dl_info di;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &di) && di.dli_fname) return di.dli_fname;
    return "";
}

static std::string g_rccl_path;

static int load()
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return FZ_OK;
    std::vector<std::string> names;
    if (const char *e = getenv("FUZZYHIP_RCCL_LIB")) if (*e) names.push_back(e);
    const std::string hip = hip_runtime_path();
    const size_t slash = hip.rfind('/');
    if (slash != std::string::npos) {
        names.push_back(hip.substr(0, slash + 1) + "librccl.so.1");
        names.push_back(hip.substr(0, slash + 1) + "librccl.so");
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    void *h = nullptr;
    std::string tried;
    for (const std::string &nm : names) {
        h = dlopen(nm.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (h) { g_rccl_path = nm; break; }
        const char *e = dlerror();
        tried += std::string(tried.empty() ? "" : "; ") + nm + ": " + (e ? e : "?");
    }
    if (!h) return fail(FZ_ERR_COMM, "librccl could not be opened (%s)", tried.c_str());
    Api a; a.handle = h;
    a.get_unique_id = (GetUniqueId)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (CommInitRank)dlsym(h, "ncclCommInitRank");
    a.comm_destroy = (CommDestroy)dlsym(h, "ncclCommDestroy");
    a.all_gather = (AllGather)dlsym(h, "ncclAllGather");
    a.get_error_string = (GetErrorString)dlsym(h, "ncclGetErrorString");
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.get_error_string) {
        dlclose(h);
        return fail(FZ_ERR_COMM, "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString");
    }
    {
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(a.all_gather), &di) && di.dli_fname) g_rccl_path = di.dli_fname;
    }
    g_api = a;
    return FZ_OK;
}
}  // namespace rccl

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        int r_ = (expr);                                                                            \
        if (r_ != 0) return fail(FZ_ERR_COMM, "%s: %s", #expr, rccl::g_api.get_error_string(r_));   \
    } while (0)

struct fz_comm {
    rccl::Comm comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    std::mutex mu;
    // per rank and call: nq*k keys + one status word (0 ok, 1 the rank's local step failed, 2 a fused list overflowed)
    uint64_t *d_local = nullptr, *d_all = nullptr, *d_out = nullptr, *h_out = nullptr;
    size_t local_cap = 0, all_cap = 0, out_cap = 0, h_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~fz_comm()
    {
        if (hipSetDevice(device) != hipSuccess) return;
        if (comm) rccl::g_api.comm_destroy(comm);
        if (d_local) (void)hipFree(d_local);
        if (d_all) (void)hipFree(d_all);
        if (d_out) (void)hipFree(d_out);
        if (h_out) (void)hipHostFree(h_out);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

// exchange buffers for `words` u64 per rank; grown outside the steady state only (fz_comm_create reserves
// FZ_COMM_RESERVE_WORDS, enough for 10^4 queries x top-10)
#define FZ_COMM_RESERVE_WORDS (131072u + 1u)
static int comm_reserve(fz_comm *cm, size_t words)
{
    auto grow = [&](uint64_t **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return FZ_OK;
        if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), need * 8));
        *cap = need;
        return FZ_OK;
    };
    int rc;
    if ((rc = grow(&cm->d_local, &cm->local_cap, words)) != FZ_OK) return rc;
    if ((rc = grow(&cm->d_all, &cm->all_cap, words * (size_t)cm->n_ranks)) != FZ_OK) return rc;
    if ((rc = grow(&cm->d_out, &cm->out_cap, words)) != FZ_OK) return rc;
    if (words + 1 > cm->h_cap) {
        if (cm->h_out) { HIP_TRY(hipHostFree(cm->h_out)); cm->h_out = nullptr; cm->h_cap = 0; }
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&cm->h_out), (words + 1) * 8, hipHostMallocDefault));
        cm->h_cap = words + 1;
    }
    return FZ_OK;
}

static int fz_comm_unique_id_impl(void *id_out)
{
    if (!id_out) return fail(FZ_ERR_INVALID, "null id buffer");
    int rc;
    if ((rc = rccl::load()) != FZ_OK) return rc;
    rccl::UniqueId id;
    RCCL_TRY(rccl::g_api.get_unique_id(&id));
    memcpy(id_out, id.internal, FZ_COMM_ID_BYTES);
    return FZ_OK;
}

static int fz_comm_create_impl(const void *id, int n_ranks, int rank, int device, fz_comm **out)
{
    if (!id || !out) return fail(FZ_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(FZ_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    int rc;
    if ((rc = rccl::load()) != FZ_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FZ_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= ndev) return fail(FZ_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<fz_comm> cm(new fz_comm());
    cm->n_ranks = n_ranks; cm->rank = rank; cm->device = device;
    rccl::UniqueId uid;
    memcpy(uid.internal, id, FZ_COMM_ID_BYTES);
    RCCL_TRY(rccl::g_api.comm_init_rank(&cm->comm, n_ranks, uid, rank));
    HIP_TRY(hipEventCreate(&cm->ev0));
    HIP_TRY(hipEventCreate(&cm->ev1));
    if ((rc = comm_reserve(cm.get(), FZ_COMM_RESERVE_WORDS)) != FZ_OK) return rc;
    *out = cm.release();
    return FZ_OK;
}

// Local top-k keys -> all-gather on the corpus' stream -> device merge of the n_ranks lists per query ->
// one copy of nq*k keys to the host.  Every rank ends up with the same, global, result.
//
// The local step is only ENQUEUED (no host synchronisation in front of the collective), and every rank enters
// the collective whatever happened to its local step: a status word travels behind each rank's keys (0 ok, 1
// the local step failed -- its keys are zeroed --, 2 a fused list overflowed, so the keys are not final).  After
// the one synchronisation at the end every rank knows the worst status: on 1 all ranks return an error (the
// failing rank its own), on 2 all ranks go round once more, the overflowed ones after redoing the rows whose lists
// overflowed through the score matrix (redo_async_overflow), the others with the keys they have.
hipError_t fz_launch_merge_gathered_strided(const uint64_t *all, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                                            uint64_t *out, uint64_t *status_out, hipStream_t st);

static int fz_query_topk_sharded_impl(fz_corpus *c, fz_comm *cm, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                                     int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx, int32_t *out_score,
                                     uint32_t *out_cnt)
{
    if (!c || !cm) return fail(FZ_ERR_INVALID, "null corpus or communicator");
    if (nq && k && (!out_idx || !out_score)) return fail(FZ_ERR_INVALID, "null output buffers");
    if (c->device != cm->device) return fail(FZ_ERR_INVALID, "the corpus lives on device %d, the communicator on device %d", c->device, cm->device);
    if (k > FZ_MAX_K) return fail(FZ_ERR_LIMIT, "k=%u exceeds FZ_MAX_K=%u", k, FZ_MAX_K);
    if ((uint64_t)cm->n_ranks * k > FZ_COMM_MAX_MERGE)
        return fail(FZ_ERR_LIMIT, "n_ranks * k = %llu exceeds FZ_COMM_MAX_MERGE=%u (gather with fz_query_topk_keys and merge with fz_merge_topk instead)",
                    (unsigned long long)cm->n_ranks * k, FZ_COMM_MAX_MERGE);
    if (nq == 0 || k == 0) {
        if (out_cnt && nq) memset(out_cnt, 0, (size_t)nq * 4);
        return FZ_OK;
    }
    int view, kmode; FzTies ties; int rc;
    // argument errors are the same on every rank (same queries, scorer, k): returning here leaves no peer waiting
    if ((rc = scorer_plan(scorer & 0xFF, &view, &kmode, &ties)) != FZ_OK) return rc;
    std::lock_guard<std::mutex> lk(cm->mu);
    std::lock_guard<std::mutex> lc(c->mu);                          // the corpus' stream, workspace and timing record, for the whole call
    HIP_TRY(hipSetDevice(cm->device));
    const size_t nkeys = (size_t)nq * k, words = nkeys + 1;
    if ((rc = comm_reserve(cm, words)) != FZ_OK) return rc;         // grows only beyond what fz_comm_create reserved
    uint64_t *d_status = cm->d_local + nkeys;
    hipStream_t st = c->stream;
    float exchange_ms = 0.f;
    bool redo = false;
    for (int round = 0; round < 2; ++round) {
        int local_rc = FZ_OK;
        std::string local_err;
        if (round == 0) {
            RunOut o; o.d_keys = cm->d_local; o.d_cnt = nullptr; o.sync = 0; o.d_status = d_status;
            o.flat = (scorer & FZ_ORDER_BY_INDEX) ? 1u : 0u;
            local_rc = run_queries_locked(c, q_cps, q_offs, nq, view, kmode, ties, min_score, k, RUN_TOPK_DEVICE, o);
        } else if (redo) {
            local_rc = redo_async_overflow(c, q_cps, q_offs, nq);    // only the rows whose lists overflowed
        }
        if (local_rc != FZ_OK) local_err = g_err;
        st = c->stream;
        if (local_rc != FZ_OK) {
            // still enter the collective, with no keys and status 1: the peers must not wait for this rank for ever
            HIP_TRY(hipMemsetAsync(cm->d_local, 0, words * 8, st));
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_status), 1, 1, st));
        } else if (round == 1) {
            HIP_TRY(hipMemsetAsync(d_status, 0, 8, st));            // keys are final now (rows redone, or never in doubt)
        }
        HIP_TRY(hipEventRecord(cm->ev0, st));
        RCCL_TRY(rccl::g_api.all_gather(cm->d_local, cm->d_all, words, rccl::kUint64, cm->comm, st));
        HIP_TRY(fz_launch_merge_gathered_strided(cm->d_all, words, (uint32_t)cm->n_ranks, nq, k, cm->d_out, cm->d_out + nkeys, st));
        HIP_TRY(hipEventRecord(cm->ev1, st));
        HIP_TRY(hipMemcpyAsync(cm->h_out, cm->d_out, words * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(cm->h_out + words, d_status, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        { float ms = 0; if (hipEventElapsedTime(&ms, cm->ev0, cm->ev1) == hipSuccess) exchange_ms += ms; }
        if (round == 0) finish_pending_timing(c);
        c->timing.exchange_ms = exchange_ms;
        const uint64_t worst = cm->h_out[nkeys], mine = cm->h_out[words];
        if (local_rc != FZ_OK) return fail(local_rc, "%s", local_err.c_str());
        if (worst == 1ull) return fail(FZ_ERR_COMM, "another rank failed in its local top-k step (its own call returns the cause); no result");
        if (worst == 0ull) break;
        if (round == 1) return fail(FZ_ERR_COMM, "a rank reported unfinished keys twice");
        redo = mine == 2ull;                                        // worst == 2: one more round, on every rank
    }
    decode_keys(cm->h_out, nq, k, out_idx, out_score, out_cnt);
    return FZ_OK;
}

// ---------------------------------------------------------------------------------------------
// the C ABI: every entry point that can allocate runs inside guarded() (no exception leaves the library)
// ---------------------------------------------------------------------------------------------
extern "C" int fz_corpus_create(const uint32_t *cps, const uint64_t *offs, uint64_t n, int device,
                                uint64_t base_index, fz_corpus **out)
{
    return guarded([&] { return fz_corpus_create_impl(cps, offs, n, device, base_index, out); });
}

extern "C" int fz_corpus_save(fz_corpus *c, const char *path)
{
    return guarded([&] { return fz_corpus_save_impl(c, path); });
}

extern "C" int fz_corpus_load(const char *path, int device, uint64_t base_index, fz_corpus **out)
{
    return guarded([&] { return fz_corpus_load_impl(path, device, base_index, out); });
}

extern "C" int fz_query_topk(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                             int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                             int32_t *out_score, uint32_t *out_cnt)
{
    return guarded([&] { return fz_query_topk_impl(c, q_cps, q_offs, nq, scorer, min_score, k, out_idx, out_score, out_cnt); });
}

extern "C" int fz_query_topk_submit(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                                    int scorer, int32_t min_score, uint32_t k, int *ticket)
{
    return guarded([&] { return fz_query_topk_submit_impl(c, q_cps, q_offs, nq, scorer, min_score, k, ticket); });
}

extern "C" int fz_query_topk_wait(fz_corpus *c, int ticket, uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt)
{
    return guarded([&] { return fz_query_topk_wait_impl(c, ticket, out_idx, out_score, out_cnt); });
}

extern "C" int fz_query_topk_noalloc(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                                     int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                                     int32_t *out_score, uint32_t *out_cnt)
{
    return guarded([&] { return fz_query_topk_noalloc_impl(c, q_cps, q_offs, nq, scorer, min_score, k, out_idx, out_score, out_cnt); });
}

extern "C" int fz_query_topk_keys(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs,
                                  uint32_t nq, int scorer, int32_t min_score, uint32_t k,
                                  uint64_t *out_keys_device, uint32_t *out_cnt_device, int sync)
{
    return guarded([&] { return fz_query_topk_keys_impl(c, q_cps, q_offs, nq, scorer, min_score, k, out_keys_device, out_cnt_device, sync); });
}

extern "C" int fz_score_all(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                            int scorer, uint8_t *out_scores)
{
    return guarded([&] { return fz_score_all_impl(c, q_cps, q_offs, nq, scorer, out_scores); });
}

extern "C" int fz_distance_all(fz_corpus *c, const uint32_t *q_cps, const uint64_t *q_offs, uint32_t nq,
                               int kind, uint16_t *out_dist)
{
    return guarded([&] { return fz_distance_all_impl(c, q_cps, q_offs, nq, kind, out_dist); });
}

extern "C" int fz_query_topk_pair(fz_corpus *ca, int scorer_a, const uint32_t *qa_cps, const uint64_t *qa_offs,
                                  fz_corpus *cb, int scorer_b, const uint32_t *qb_cps, const uint64_t *qb_offs,
                                  uint32_t nq, const uint16_t *rank_tab, uint32_t min_rank, uint32_t k,
                                  uint64_t *out_idx, int32_t *out_rank, uint32_t *out_cnt)
{
    return guarded([&] {
        return fz_query_topk_pair_impl(ca, scorer_a, qa_cps, qa_offs, cb, scorer_b, qb_cps, qb_offs, nq, rank_tab,
                                       min_rank, k, out_idx, out_rank, out_cnt);
    });
}

extern "C" int fz_merge_topk(const uint64_t *keys, uint32_t n_lists, uint32_t nq, uint32_t k,
                             uint64_t *out_idx, int32_t *out_score, uint32_t *out_cnt)
{
    return guarded([&] { return fz_merge_topk_impl(keys, n_lists, nq, k, out_idx, out_score, out_cnt); });
}

extern "C" int fz_distance_batch(int device, int kind, const uint32_t *a_cps, const uint64_t *a_offs,
                                 const uint32_t *b_cps, const uint64_t *b_offs, uint64_t n_pairs,
                                 uint32_t *out_dist)
{
    return guarded([&] { return fz_distance_batch_impl(device, kind, a_cps, a_offs, b_cps, b_offs, n_pairs, out_dist); });
}

// Which HIP runtime this library is bound to, and which librccl it opened ("" before the first fz_comm_* call).
extern "C" int fz_runtime_paths(char *hip_out, size_t hip_cap, char *rccl_out, size_t rccl_cap)
{
    return guarded([&] {
        const std::string hip = rccl::hip_runtime_path();
        std::string rc;
        { std::lock_guard<std::mutex> lk(rccl::g_api_mu); rc = rccl::g_rccl_path; }
        if (hip_out && hip_cap) { snprintf(hip_out, hip_cap, "%s", hip.c_str()); }
        if (rccl_out && rccl_cap) { snprintf(rccl_out, rccl_cap, "%s", rc.c_str()); }
        return (int)FZ_OK;
    });
}

extern "C" int fz_comm_unique_id(void *id_out)
{
    return guarded([&] { return fz_comm_unique_id_impl(id_out); });
}

extern "C" int fz_comm_create(const void *id, int n_ranks, int rank, int device, fz_comm **out)
{
    return guarded([&] { return fz_comm_create_impl(id, n_ranks, rank, device, out); });
}

extern "C" int fz_comm_destroy(fz_comm *comm)
{
    return guarded([&] { delete comm; return (int)FZ_OK; });
}

extern "C" int fz_query_topk_sharded(fz_corpus *c, fz_comm *comm, const uint32_t *q_cps, const uint64_t *q_offs,
                                     uint32_t nq, int scorer, int32_t min_score, uint32_t k, uint64_t *out_idx,
                                     int32_t *out_score, uint32_t *out_cnt)
{
    return guarded([&] { return fz_query_topk_sharded_impl(c, comm, q_cps, q_offs, nq, scorer, min_score, k, out_idx, out_score, out_cnt); });
}
