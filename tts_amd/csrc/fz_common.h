// fz_common.h -- structures shared by the host side (fz_host.cpp) and the gfx950 kernels
// (fz_kernels.hip) of libfuzzyhip.so.  See DESIGN.md for the HBM layout these describe.
#pragma once
#include <stdint.h>

#define FZ_TILE 64u          // candidates per tile = lanes per wavefront
#define FZ_SYM_NULL 0u       // padding symbol: its Peq row is all zero, so it never changes the state
#define FZ_SYM_SPACE 1u      // token separator
#define FZ_TOK_NONE 0xFFFFFFFFu
#define FZ_MODE_TOKEN_SET 0
#define FZ_MODE_RATIO 1
#define FZ_MODE_INDEL 2
#define FZ_MODE_LEV 3
#define FZ_MODE_PARTIAL 4
#define FZ_MODE_NHITS 5
#define FZ_MODE_FIND 6
#define FZ_MODE_RO 7      // matching characters of difflib.SequenceMatcher (Ratcliff/Obershelp), u16; pattern and candidate <= 64 symbols
#define FZ_IDX_BITS 40
#define FZ_EMIT_CAP 2048u          // keys a row can append in the fused top-k path (about 10x what a 1-in-16 sample leaves at k = 64)
#define FZ_FUSED_MAX_K 64u         // larger k goes through the score matrix
#define FZ_FUSED_MAX_CHUNKS 1024u  // overflow counters of one fused call
#define FZ_GRID_Y_MAX 65535u   // rows of one launch that indexes them by blockIdx.y
#define FZ_IDX_MASK ((1ull << FZ_IDX_BITS) - 1ull)

// One 64-candidate tile.  Symbols: dword d of lane l at sym[coff + d*64 + l] (4 8-bit symbols per
// dword, little endian; 2 16-bit symbols when the view is wide).  Tokens: slot i of lane l at tokid[toff + i*64 + l] / tokend[toff + i*64 + l].
struct FzTile {
    uint32_t coff;   // dword offset into the symbol array
    uint32_t toff;   // element offset into the token arrays
    uint16_t ndw;    // symbol dwords per lane in this tile (max over its candidates)
    uint16_t ntok;   // token slots per lane in this tile (max over its candidates)
};

// One prepared query (token_set / token_sort / ratio all use it; see fz_host.cpp prepare_query).
struct FzQuery {
    uint32_t peq_off;    // dword offset of this query's Peq table [alphabet][W] in the query blob
    uint32_t tok_off;    // element offset of its (id, pos, len1) token triples, sorted by id
    uint32_t n_ids;      // tokens that exist in the corpus vocabulary (only those can intersect)
    uint32_t n_tok;      // all tokens of the query (0 => every score is 0)
    uint32_t m;          // pattern length: sum(len+1) over tokens (each token + one separator)
    uint32_t out_row;    // row of the score matrix this query writes
    uint32_t map_off;    // wide alphabets: dword offset of the u16 symbol -> Peq row map [alphabet]
    uint32_t peq_rows;   // rows of this query's Peq table (alphabet, or 1 + its distinct symbols when wide)
};

// Round-to-nearest tie tables (see fz_host.cpp make_tie_table): bit q says whether the exact value
// q + 0.5 rounds up under the scorer's own fp64 formula.
struct FzTies {
    uint64_t lo, hi;
};

// ---- fused threshold + top-k (phase 2 of a top-k call) ---------------------------------------------
// A top-k call first scores the SAMPLE tiles of the view (every 16th tile of the length-sorted order, kept
// in front of the physical order) into a small score matrix and selects its top-k; that fixes, per query,
// a lower limit `lim` no result can fall below: the k-th best sample key + 1, or min_score << 40 while the
// sample holds fewer than k passing candidates.  The other tiles are then scored WITHOUT a score matrix: a
// lane whose key (score << 40 | 2^40-1 - global index) reaches the limit appends it to the query's list,
// and a last kernel merges list and sample keys.  A list that overflows its capacity is reported and the
// call falls back to the score matrix for that chunk, so the result is exact in every case.
struct FzEmit {
    const uint64_t *lim;          // [rows] smallest key that can still be a result; null: store scores instead
    uint64_t *list;               // [rows][cap] appended keys
    uint32_t *cnt;                // [rows] appended so far (may run past cap: overflow)
    uint32_t cap;
    const uint32_t *orig_of_phys; // physical position -> original index (0xFFFFFFFF: padding lane)
    uint64_t base_index;
};

struct FzScoreParams {
    const uint32_t *sym;
    const uint32_t *tokid;
    const uint16_t *tokend;
    const FzTile *tiles;
    uint32_t ntiles;
    uint32_t tiles_per_block;
    uint32_t alphabet;        // symbols of the corpus view (rows of a Peq table unless wide)
    uint32_t peq_rows;        // LDS rows reserved for a Peq table: max over the queries of the launch
    uint32_t wide;            // 1: 16-bit symbols, two per dword, Peq rows looked up through a map
    uint32_t pmax;            // capacity of the per-query token arrays in LDS
    uint32_t ntkw;            // text keep-mask words per lane (ceil(max ndw / 8))
    uint32_t nwaves;          // waves per block
    uint32_t ro;              // 1: the launch is fz_ro_matches (its own LDS layout: fz_score_lds_bytes)
    const uint32_t *tile_list;// optional: indices of the tiles to cover (null = tile_base .. tile_base+ntiles-1)
    uint32_t tile_base;
    const FzQuery *queries;   // queries of this launch (blockIdx.y indexes it)
    const uint32_t *qblob;    // Peq tables
    const uint32_t *qtok_id;
    const uint32_t *qtok_pos;
    const uint32_t *qtok_len1;
    uint8_t *scores;          // [rows][row_stride]
    uint64_t row_stride;
    FzTies ties;
    FzEmit emit;
};

// ---- stream kernel (fz_score_stream): short candidates as per-lane patterns, queries as text -----
// Eligible tiles: <= 8 token slots and <= 32*W symbols (W = 1, 2).  Eligible queries: see
// fz_host.cpp stream_append.  The query side is a stream of 16-byte records, the same for every lane:
//   header  {out_row, m | n_ids << 8 | n_token_records << 16, id0, id1}
//   ids     {id2..id5}, {id6..id9}, ... (pad FZ_QTOK_PAD): the query tokens that exist in the corpus
//           vocabulary, the only ones that can intersect, in token order
//   tokens  {4 row bytes, 4 row bytes, 4 row bytes, meta}: up to 12 text symbols of one token, each byte =
//           4 * (Peq row of the symbol), padded with 0 (row 0 is all zero in every column, so a step on it
//           changes nothing) up to the 4, 8 or 12 steps of the record's class; meta = class | last << 3 |
//           first << 4 | (2 * tpos) << 8; a longer token continues in the next record.  tpos is the bit of the
//           lane's "candidate has this token" string that belongs to the token (the kernels shift one bit per
//           id into the string, two in the packed kernel: bits 2*tpos + 1 and 2*tpos); a token without a
//           vocabulary id points at bit 15, beyond any string.  `last` says the separator step follows
//           (token_set only).  Symbols no candidate uses are dropped on the host: they cannot change the LCS
//           state.
#define FZ_STREAM_MAX_TOK 8u      // token slots of an eligible tile
#define FZ_PK_MAX_TOK 4u          // ... of a tile the packed kernel takes (two tiles per wave, <= 31 symbols together)
#define FZ_PK_SPLIT_SHIFT 27u     // packed tile list: entry 2i = tile A | split << 27, entry 2i + 1 = tile B
#define FZ_STREAM_MAX_ROWS 63u    // Peq rows (the zero row + the distinct text symbols of one launch), row byte = 4*row < 256
#define FZ_STREAM_MAX_IDS 15u     // known tokens of a streamed query
#define FZ_STREAM_QUEUES 32u      // work queues of one launch (each a contiguous range of items)
#define FZ_STREAM_QPAD 64u        // u32 words between two queue counters (256 B: one counter per cache line)
#define FZ_STREAM_TAB 128u        // score table covers length sums L < 128 ...
#define FZ_STREAM_TAB_BYTES (FZ_STREAM_TAB * FZ_STREAM_TAB / 2u)   // ... as T[L][d >> 1]: d and L have the same parity
#define FZ_QTOK_UNKNOWN 0xFFFFFFFEu   // token that is not in the corpus vocabulary
#define FZ_QTOK_CONT 0xFFFFFFFDu      // continuation record of the previous token
#define FZ_QTOK_PAD 0xFFFFFFFCu       // padding of an id record

struct FzRec { uint32_t x, y, z, w; };

struct FzStreamParams {
    const uint32_t *sym;
    const uint32_t *tokid;
    const uint16_t *tokend;
    const FzTile *tiles;
    const uint32_t *tile_list; // indices of the eligible tiles
    uint32_t ntiles;           // entries of tile_list
    uint32_t nwaves;           // waves per block (1..8): they share the block's ONE unit and split the item's queries
    uint32_t stride;           // dwords per lane of one Peq table word plane (odd, > max row)
    uint32_t nq;               // queries of this launch
    uint32_t q_per_block;      // queries per group
    uint32_t nitems;           // work items = units (tiles or tile pairs) * query groups, numbered slab-major:
                               // all query groups over the first slab_units units, then over the next slab, ...
    uint32_t slab_units;       // units per slab: a slab's symbols, ids and ends stay cache resident while every
                               // query group passes over it (a shard larger than the Infinity Cache is otherwise
                               // re-streamed from HBM once per group)
    uint32_t grid_blocks;      // persistent blocks to launch (the chip's CU count x blocks a CU holds at 8 waves per SIMD)
    uint32_t chunk;            // items a wave draws at a time (1..64)
    uint32_t *work_counter;    // FZ_STREAM_QUEUES device counters FZ_STREAM_QPAD words apart, zero before the launch
    const uint8_t *row_of_sym; // [256] corpus symbol -> 4 * row, 0xFF = no query of this launch uses it
    const FzRec *recs;         // record stream
    const uint32_t *qrec;      // [nq] index of each query's header record
    const uint8_t *score_tab;  // [128*64] rounded ratio T[L*64 + (d >> 1)], d of L's parity
    uint8_t *scores;
    uint32_t row_stride;       // bytes per score row (< 2^32)
    FzEmit emit;
};

struct FzSelectParams {
    const uint8_t *scores;
    uint64_t row_stride;
    uint32_t n;               // physical positions of a row that hold candidates (positions >= n are padding or absent)
    uint32_t n_orig;          // original indices of the view (the walk in original order skips those whose position is >= n)
    const uint32_t *orig_of_phys;
    const uint32_t *phys_of_orig;
    const uint32_t *walk;     // optional [n]: the original indices of the row's candidates, ascending (null: 0..n_orig-1)
    int32_t min_score;
    uint32_t k;
    uint64_t base_index;
    uint64_t *out_keys;       // [rows][k]
    uint32_t *out_cnt;        // [rows] or null
    uint32_t flat;            // 1: every passing score counts as min_score, so the order is by index only
    uint64_t *out_lim;        // [rows] or null: the limit the row's result sets for the tiles not in it (FzEmit)
    uint64_t *dbg;            // diagnostics (FUZZYHIP_SELECT_DEBUG=1): [rows][8] phase timestamps, or null
    // A few rows over a large shard: the 64-position tiles of every row are dealt round-robin to n_slices blocks
    // (blockIdx.y; a power of two), each block selects the top-k of its tiles, and the block that finishes last
    // merges the row's n_slices * k keys (n_slices * k <= FZ_MAX_K).  n_slices <= 1: one block per row.
    uint32_t n_slices;
    uint64_t *slice_keys;     // [rows][n_slices][k]
    uint32_t *slice_done;     // [rows] blocks of the row that have written their keys; zero before the launch
};

// ---- two scorers combined through a rank table (fz_query_topk_pair) ------------------------------
#define FZ_RANK_DIM 101u          // scores 0..100 on both axes

struct FzCombineParams {
    const uint8_t *scores_a;      // [rows][stride_a], physical order of view a
    const uint8_t *scores_b;
    uint64_t stride_a, stride_b;
    const uint32_t *phys_a;       // [n] original index -> physical position in view a
    const uint32_t *phys_b;
    const uint16_t *rank_tab;     // [101*101] rank of the pair (a, b); larger = better
    uint16_t *ranks;              // out [rows][stride_r], ORIGINAL candidate order
    uint64_t stride_r;
    uint32_t n;
};

struct FzSelect16Params {
    const uint16_t *ranks;        // [rows][stride_r], original order
    uint64_t stride_r;
    uint32_t n;
    uint32_t min_rank;            // keep rank >= min_rank
    uint32_t k;
    uint64_t base_index;
    uint64_t *out_keys;           // [rows][k]: (rank << 40) | (2^40-1 - global index), 0 = padding
    uint32_t *out_cnt;
};

// merge of the sample's top-k keys with the appended lists (fz_merge_lists)
struct FzMergeParams {
    const uint64_t *sample_keys;  // [rows][k], 0-padded
    const uint64_t *list;         // [rows][cap]
    const uint32_t *cnt;          // [rows]
    uint32_t cap, k;
    uint64_t *out_keys;           // [rows][k]; may alias sample_keys
    uint32_t *out_cnt;            // [rows] or null
    uint32_t *overflow;           // one counter: rows (of this launch) whose list ran past cap
    uint32_t *ovf_rows;           // [0]: rows that overflowed so far in the whole call, [1 + i]: their row numbers (first ovf_rows_cap of them)
    uint32_t ovf_rows_cap;
    uint32_t row_base;            // row number of this launch's row 0 within the call
};
#define FZ_FUSED_REDO_ROWS 256u    // up to this many overflowed rows of a call are redone one by one; more: whole chunks
