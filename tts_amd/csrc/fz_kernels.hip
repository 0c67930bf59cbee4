// fz_kernels.hip -- gfx950 (MI355X, wave64) kernels of libfuzzyhip.so.
//
//  fz_score_general<W>  one candidate per lane, one query per blockIdx.y.  Bit-parallel LCS
//                       (Hyyro) of the query pattern (W 32-bit words, Peq table in LDS) against the
//                       candidate's canonical symbols, with the token-set intersection applied as
//                       bit masks on both sides, then the exact integer form of thefuzz's rounded
//                       ratio.  Writes one u8 score per (query, candidate).
//  fz_select_topk       one block per query row: threshold, (score desc, original index asc)
//                       top-k with exact tie handling.
//
// Reference semantics: src/fuzzy_search.py:22-38 (loop, threshold, stable sort, slice) and the
// third-party scorer it calls.  DESIGN.md derives the masks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "fuzzyhip.h"
#include "fz_common.h"

// ---------------------------------------------------------------------------------------------
// exact rounded ratio: round(100 * (L - d) / L), ties resolved by the host-evaluated table
// ---------------------------------------------------------------------------------------------
// thefuzz returns int(round(x)) of x = 100 - 100*d/L (token_set) or (1 - d/L)*100 (ratio), both in
// fp64.  x differs from the exact rational by ~1e-13 while a non-tie rational is >= 1/(2L) away
// from any k+0.5, so the fp64 result equals exact rounding except on exact ties, where the fp64
// evaluation (a function of d/L only) decides; the host evaluates those 100 cases once per formula
// (fz_host.cpp make_tie_table) and hands them over as a 100-bit table.  Integer-only on the device.
__device__ __forceinline__ uint32_t fz_round_ratio(uint32_t d, uint32_t L, FzTies ties)
{
    uint32_t N = 100u * (L - d);                    // < 2^24 for every supported length
    float r = __builtin_amdgcn_rcpf((float)L);
    uint32_t q = (uint32_t)((float)N * r);          // floor(N/L) +- 1
    int32_t rem = (int32_t)(N - q * L);
    if (rem < 0) { q -= 1; rem += (int32_t)L; }
    else if ((uint32_t)rem >= L) { q += 1; rem -= (int32_t)L; }
    uint32_t twice = 2u * (uint32_t)rem;
    if (twice > L) {
        q += 1;
    } else if (twice == L) {
        uint64_t t = q < 64 ? (ties.lo >> q) : (ties.hi >> (q - 64));
        q += (uint32_t)(t & 1ull);
    }
    return q;
}

// token_set_ratio from the integer summary of one (query, candidate) pair.
//   n_qtok : tokens in the query, m: pattern length = sum(len+1) over query tokens
//   clen   : canonical candidate length = sum(len+1) over its tokens (0 = no tokens)
//   nhits  : tokens in the intersection, sect1 = sum(len+1) over them
//   lcs1   : LCS of (kept pattern, kept text), each with one separator after EVERY kept token; as
//            both then end in a separator this is LCS(diff_ab_joined, diff_ba_joined) + 1.
// Follows rapidfuzz fuzz::token_set_ratio: r0 on the joined differences,
// r1/r2 from the length differences, max, then thefuzz's int(round()).  max(r1, r2) is the one with
// the shorter difference (both are decreasing in it), so two roundings suffice; rounding is monotone,
// hence max-then-round == round-then-max.
__device__ __forceinline__ uint32_t fz_token_set_score(uint32_t n_qtok, uint32_t m, uint32_t clen,
                                                       uint32_t nhits, uint32_t sect1,
                                                       uint32_t lcs1, FzTies ties)
{
    if (n_qtok == 0u || clen == 0u) return 0u;        // either token list empty -> 0
    uint32_t PA = m - sect1, TB = clen - sect1;
    if (nhits != 0u && (PA == 0u || TB == 0u)) return 100u;   // one token set contains the other
    uint32_t ab = PA - 1u, ba = TB - 1u, lcs = lcs1 - 1u;
    uint32_t d0 = ab + ba - 2u * lcs;
    if (nhits == 0u) return fz_round_ratio(d0, ab + ba, ties);
    uint32_t sl = sect1 - 1u;
    uint32_t r0 = fz_round_ratio(d0, 2u * sl + 2u + ab + ba, ties);
    uint32_t mn = ab < ba ? ab : ba;
    uint32_t r1 = fz_round_ratio(1u + mn, 2u * sl + 1u + mn, ties);
    return r0 > r1 ? r0 : r1;
}

// Fused threshold + top-k (FzEmit, fz_common.h): every lane of the wave calls this with its score; lanes whose
// key reaches the row's limit append it to the row's list (one counter add per wave, keys behind it in lane
// order).  `orig` = 0xFFFFFFFF marks a padding lane.  Wave-uniform control flow only.
__device__ __forceinline__ void fz_emit(const FzEmit &e, uint32_t row, uint64_t lim, uint32_t sc, uint32_t orig)
{
    const uint64_t key = ((uint64_t)sc << FZ_IDX_BITS) | (FZ_IDX_MASK - (e.base_index + (uint64_t)orig));
    const bool pass = orig != 0xFFFFFFFFu && sc <= 100u && key >= lim;
    const uint64_t m = __ballot(pass);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u, first = (uint32_t)__builtin_ctzll(m);
    uint32_t pos = 0u;
    if (lane == first) pos = atomicAdd(&e.cnt[row], (uint32_t)__popcll(m));
    pos = (uint32_t)__shfl((int)pos, (int)first, 64);
    const uint32_t mine = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pass && mine < e.cap) e.list[(uint64_t)row * e.cap + mine] = key;
}

// clear bits [s, e) (e > s) of a per-lane bit row stored word-major with stride 64 (row[w*64])
// One word of the W-word Hyyro update with the carry of the big addition kept where the hardware keeps it: a
// lane mask in a scalar register pair, produced by v_add_co / v_addc_co and consumed by the next word's
// v_addc_co.  Written as 64-bit arithmetic the compiler materialises every carry (v_cndmask 0/1, a zero
// high word, v_lshl_add_u64): about 14 instructions per word instead of 3.
//   u = v & row & keep;  t = v + u + carry;  v = t | (v ^ u)
template <bool FIRST>
__device__ __forceinline__ uint32_t fz_lcs_word(uint32_t v, uint32_t row, uint32_t keep, uint64_t &carry)
{
    const uint32_t u = __builtin_amdgcn_bitop3_b32(v, row, keep, 0x80);
    uint32_t t;
    uint64_t co;
    if constexpr (FIRST) {
        asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(t), "=s"(co) : "v"(v), "v"(u));   // EMUL: { const uint64_t t_ = (uint64_t)v + u; t = (uint32_t)t_; co = t_ >> 32; }
    } else {
        asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(co) : "v"(v), "v"(u), "s"(carry));   // EMUL: { const uint64_t t_ = (uint64_t)v + u + (carry & 1ull); t = (uint32_t)t_; co = t_ >> 32; }
    }
    carry = co;
    return __builtin_amdgcn_bitop3_b32(t, v, u, 0xF6);
}

// a + b + carry with the carry as a lane mask (the adder of the Myers / Hyyro Levenshtein recurrence)
template <bool FIRST>
__device__ __forceinline__ uint32_t fz_add_carry(uint32_t a, uint32_t b, uint64_t &carry)
{
    uint32_t t;
    uint64_t co;
    if constexpr (FIRST) {
        asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(t), "=s"(co) : "v"(a), "v"(b));   // EMUL: { const uint64_t t_ = (uint64_t)a + b; t = (uint32_t)t_; co = t_ >> 32; }
    } else {
        asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(co) : "v"(a), "v"(b), "s"(carry));   // EMUL: { const uint64_t t_ = (uint64_t)a + b + (carry & 1ull); t = (uint32_t)t_; co = t_ >> 32; }
    }
    carry = co;
    return t;
}

__device__ __forceinline__ void fz_clear_bits(uint32_t *row, uint32_t s, uint32_t e)
{
    uint32_t w0 = s >> 5, w1 = (e - 1u) >> 5;
    for (uint32_t w = w0; w <= w1; ++w) {
        uint32_t lo = (w == w0) ? (s & 31u) : 0u;
        uint32_t hi = (w == w1) ? ((e - 1u) & 31u) + 1u : 32u;
        uint32_t mask = (hi == 32u ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        row[w * 64u] &= ~mask;
    }
}

// ---------------------------------------------------------------------------------------------
// general scoring kernel: any pattern width W (32-bit words), any candidate length
// ---------------------------------------------------------------------------------------------
// MODE: FZ_MODE_TOKEN_SET  token_set_ratio score (u8)
//       FZ_MODE_RATIO      plain InDel ratio of pattern vs text, thefuzz ratio()/token_sort (u8)
//       FZ_MODE_INDEL      raw InDel distance (u16)
//       FZ_MODE_LEV        raw uniform-cost Levenshtein distance, Myers/Hyyro Pv/Mv recurrence (u16)
//       FZ_MODE_NHITS      number of candidate tokens that are query tokens (u16): set intersection size
//       FZ_MODE_FIND       1 if the pattern occurs in the text as a substring, else 0 (u16); Shift-And
// WIDE: the corpus has more than 254 distinct code points, so its symbols are 16 bits (two per
//       dword) and the Peq table is indirect: a u16 map [alphabet] -> row of a table that only holds
//       the symbols of THIS query (row 0 = all zero = every other symbol), both in LDS.
// LDS row stride (words) of a Peq table of W-word rows.  Every lane reads the row of ITS text symbol, so the 64 lanes
// of a word read land on rows chosen by the data: with an even stride rows share banks (W = 8: eight start banks for
// ~30 rows; 75 % of the kernel's cycles were LDS bank-conflict cycles, SQ_LDS_BANK_CONFLICT), with an odd stride up to
// 64 distinct rows fall on 64 distinct banks and a word read is conflict-free.
__host__ __device__ constexpr uint32_t fz_peq_stride(int W) { return (uint32_t)W | 1u; }

template <bool WIDE>
struct FzSym {
    static constexpr int SPD = WIDE ? 2 : 4;                  // symbols per dword
    static constexpr int BITS = WIDE ? 16 : 8;
    static constexpr uint32_t MASK = WIDE ? 0xFFFFu : 0xFFu;
    static constexpr uint32_t TKD = 32u / (uint32_t)SPD;      // symbol dwords covered by one keep-mask word
};

template <int W, int MODE, bool WIDE>
__global__ __launch_bounds__(256) void fz_score_general(FzScoreParams p)
{
    using SY = FzSym<WIDE>;
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr uint32_t RS = fz_peq_stride(W);
    uint32_t *s_peq = lds;                                    // [peq_rows][RS]: rows of W words at an odd stride
    uint32_t *s_qid = s_peq + p.peq_rows * RS;                // [pmax] query token ids, ascending
    uint32_t *s_qpos = s_qid + p.pmax;                        // first pattern bit of the token
    uint32_t *s_qlen = s_qpos + p.pmax;                       // token length + 1
    uint32_t *s_M = s_qlen + p.pmax;                          // [nwaves][W][64] pattern keep masks
    uint32_t *s_TK = s_M + p.nwaves * (uint32_t)W * 64u;      // [nwaves][ntkw][64] text keep masks
    const uint16_t *s_map = reinterpret_cast<const uint16_t *>(s_TK + p.nwaves * p.ntkw * 64u);   // WIDE: [alphabet]

    const FzQuery q = p.queries[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    for (uint32_t i = tid; i < q.peq_rows * (uint32_t)W; i += blockDim.x) s_peq[(i / (uint32_t)W) * RS + i % (uint32_t)W] = p.qblob[q.peq_off + i];
    if (WIDE) {
        uint32_t *m32 = s_TK + p.nwaves * p.ntkw * 64u;
        for (uint32_t i = tid; i < (p.alphabet + 1u) / 2u; i += blockDim.x) m32[i] = p.qblob[q.map_off + i];
    }
    for (uint32_t i = tid; i < q.n_ids; i += blockDim.x) {
        s_qid[i] = p.qtok_id[q.tok_off + i];
        s_qpos[i] = p.qtok_pos[q.tok_off + i];
        s_qlen[i] = p.qtok_len1[q.tok_off + i];
    }
    __syncthreads();

    uint32_t *Mrow = s_M + wave * (uint32_t)W * 64u + lane;
    uint32_t *TKrow = s_TK + wave * p.ntkw * 64u + lane;
    const uint32_t nsteps = q.n_ids ? 32u - (uint32_t)__builtin_clz(q.n_ids) : 0u;

    uint32_t t_end = (blockIdx.x + 1u) * p.tiles_per_block;
    if (t_end > p.ntiles) t_end = p.ntiles;
    for (uint32_t tix = blockIdx.x * p.tiles_per_block + wave; tix < t_end; tix += p.nwaves) {
        const uint32_t tile = p.tile_list ? p.tile_list[tix] : p.tile_base + tix;
        const FzTile td = p.tiles[tile];
        const uint32_t ntkw = ((uint32_t)td.ndw + SY::TKD - 1u) / SY::TKD;

#pragma unroll
        for (int w = 0; w < W; ++w) Mrow[w * 64] = 0xFFFFFFFFu;
        for (uint32_t t = 0; t < ntkw; ++t) TKrow[t * 64u] = 0xFFFFFFFFu;

        // ---- token intersection: which candidate tokens are query tokens ---------------------
        uint32_t sect1 = 0u, nhits = 0u, prev_end = 0u;
        if (MODE != FZ_MODE_TOKEN_SET && MODE != FZ_MODE_NHITS) {
            // plain views carry one slot per candidate: its length
            if (td.ntok) prev_end = p.tokend[td.toff + lane];
        } else
        for (uint32_t i = 0; i < td.ntok; ++i) {
            const uint32_t cid = p.tokid[td.toff + i * 64u + lane];
            const uint32_t cend = p.tokend[td.toff + i * 64u + lane];
            uint32_t lo = 0u, hi = q.n_ids;
            for (uint32_t s = 0; s < nsteps; ++s) {          // lower_bound, uniform trip count
                if (lo < hi) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (s_qid[mid] < cid) lo = mid + 1u; else hi = mid;
                }
            }
            if (lo < q.n_ids && s_qid[lo] == cid) {
                const uint32_t pos = s_qpos[lo], len1 = s_qlen[lo];
                fz_clear_bits(Mrow, pos, pos + len1);         // drop the token from the pattern
                fz_clear_bits(TKrow, prev_end, cend);         // and from the text
                sect1 += len1;
                nhits += 1u;
            }
            prev_end = cend;
        }
        const uint32_t clen = prev_end;                       // pad slots carry end = length

        if (MODE == FZ_MODE_NHITS) {
            reinterpret_cast<uint16_t *>(p.scores)[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint16_t)nhits;
            continue;
        }

        const uint32_t *symp = p.sym + td.coff + lane;
        if (MODE == FZ_MODE_FIND) {
            // ---- Shift-And: bit j of D = "pattern[0..j] ends here"; a set bit m-1 is an occurrence --
            uint32_t D[W];
#pragma unroll
            for (int w = 0; w < W; ++w) D[w] = 0u;
            uint32_t found = q.m == 0u ? 1u : 0u;                 // '' in t is True
            const uint32_t lastw = q.m ? (q.m - 1u) >> 5 : 0u, lastb = q.m ? (q.m - 1u) & 31u : 0u;
            for (uint32_t d = 0; d < td.ndw; ++d) {
                const uint32_t cw = symp[d * 64u];
#pragma unroll
                for (int k = 0; k < SY::SPD; ++k) {
                    const uint32_t s = (cw >> (SY::BITS * k)) & SY::MASK;   // padding symbols clear D: no false hit
                    const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                    uint32_t carry = 1u;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const uint32_t nxt = D[w] >> 31;
                        D[w] = ((D[w] << 1) | carry) & row[w];
                        carry = nxt;
                    }
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if ((uint32_t)w == lastw) found |= (D[w] >> lastb) & 1u;
                }
            }
            reinterpret_cast<uint16_t *>(p.scores)[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint16_t)found;
            continue;
        }
        if (MODE == FZ_MODE_LEV) {
            // ---- Myers/Hyyro bit-vector Levenshtein; the W words act as one big integer --------
            uint32_t Pv[W], Mv[W];
#pragma unroll
            for (int w = 0; w < W; ++w) { Pv[w] = 0xFFFFFFFFu; Mv[w] = 0u; }
            uint32_t dist = q.m;
            const uint32_t lastw = q.m ? (q.m - 1u) >> 5 : 0u, lastb = q.m ? (q.m - 1u) & 31u : 0u;
            for (uint32_t d = 0; d < td.ndw; ++d) {
                const uint32_t cw = symp[d * 64u];
#pragma unroll
                for (int k = 0; k < SY::SPD; ++k) {
                    if (d * (uint32_t)SY::SPD + (uint32_t)k < clen && q.m != 0u) {     // padding is not text here
                        const uint32_t s = (cw >> (SY::BITS * k)) & SY::MASK;
                        const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                        uint64_t carry = 0ull;
                        uint32_t phc = 1u, mhc = 0u;
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            const uint32_t Eq = row[w];
                            const uint32_t Xv = Eq | Mv[w];
                            const uint32_t t = w == 0 ? fz_add_carry<true>(Eq & Pv[w], Pv[w], carry) : fz_add_carry<false>(Eq & Pv[w], Pv[w], carry);
                            const uint32_t Xh = (t ^ Pv[w]) | Eq;
                            uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
                            uint32_t Mh = Pv[w] & Xh;
                            if ((uint32_t)w == lastw) dist += ((Ph >> lastb) & 1u) - ((Mh >> lastb) & 1u);
                            const uint32_t phn = Ph >> 31, mhn = Mh >> 31;
                            Ph = (Ph << 1) | phc; Mh = (Mh << 1) | mhc;
                            phc = phn; mhc = mhn;
                            Pv[w] = Mh | ~(Xv | Ph);
                            Mv[w] = Ph & Xv;
                        }
                    }
                }
            }
            if (q.m == 0u) dist = clen;
            reinterpret_cast<uint16_t *>(p.scores)[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint16_t)dist;
            continue;
        }

        uint32_t M[W], V[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { M[w] = Mrow[w * 64]; V[w] = 0xFFFFFFFFu; }

        // ---- bit-parallel LCS over the kept text --------------------------------------------
        // the next symbol dword is asked for before this one is scanned: its latency (the corpus of long strings comes
        // from HBM) passes under ~100 instructions of carry chain instead of being waited for at the top of every turn
        uint32_t tkw = 0u, cw_next = td.ndw ? symp[0] : 0u;
        for (uint32_t d = 0; d < td.ndw; ++d) {
            const uint32_t cw = cw_next;
            if (d + 1u < td.ndw) cw_next = symp[(d + 1u) * 64u];
            if ((d % SY::TKD) == 0u) tkw = TKrow[(d / SY::TKD) * 64u];
#pragma unroll
            for (int k = 0; k < SY::SPD; ++k) {
                uint32_t s = (cw >> (SY::BITS * k)) & SY::MASK;
                if (((tkw >> k) & 1u) == 0u) s = FZ_SYM_NULL;  // removed token: no match, no change
                const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                uint64_t carry = 0ull;
                V[0] = fz_lcs_word<true>(V[0], row[0], M[0], carry);
#pragma unroll
                for (int w = 1; w < W; ++w) V[w] = fz_lcs_word<false>(V[w], row[w], M[w], carry);
            }
            tkw >>= SY::SPD;
        }
        uint32_t lcs1 = 0u;
#pragma unroll
        for (int w = 0; w < W; ++w) lcs1 += (uint32_t)__builtin_popcount(~V[w]);

        if (MODE == FZ_MODE_RATIO || MODE == FZ_MODE_INDEL) {
            const uint32_t L = q.m + clen, dist = L - 2u * lcs1;
            if (MODE == FZ_MODE_INDEL) {
                reinterpret_cast<uint16_t *>(p.scores)[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint16_t)dist;
            } else {
                const uint32_t sc = L ? fz_round_ratio(dist, L, p.ties) : 100u;   // two empty strings: 100
                if (p.emit.lim) fz_emit(p.emit, q.out_row, p.emit.lim[q.out_row], sc, p.emit.orig_of_phys[tile * 64u + lane]);
                else p.scores[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint8_t)sc;
            }
            continue;
        }

        const uint32_t sc = fz_token_set_score(q.n_tok, q.m, clen, nhits, sect1, lcs1, p.ties);
        if (p.emit.lim) fz_emit(p.emit, q.out_row, p.emit.lim[q.out_row], sc, p.emit.orig_of_phys[tile * 64u + lane]);
        else p.scores[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint8_t)sc;
    }
}


// ---------------------------------------------------------------------------------------------
// Ratcliff/Obershelp matching characters: what difflib.SequenceMatcher(None, query, candidate) counts
// ---------------------------------------------------------------------------------------------
// reference app.py:329-337 (`SequenceMatcher(None, a, b).ratio()` = 2 * M / (|a| + |b|)); the kernel returns M.
// find_longest_match(alo, ahi, blo, bhi): the longest block a[i:i+k] == b[j:j+k] inside the rectangle; of the
// longest ones the one that starts earliest in a, of those the one that starts earliest in b; then the same to the
// left and to the right of it.  (difflib's "junk" heuristic starts at |b| >= 200; this kernel takes |a|, |b| <= 64
// and answers 0xFFFF otherwise: the host keeps difflib for those.)
// One candidate per lane, the query's Peq table in LDS as in fz_score_general.  Per lane, in LDS:
//   R[j]  match mask of b[j] over the positions of a (one word: |a| <= 32, two: <= 64)
//   S[j]  pass k: the positions i such that a block of length k ENDS at (i, j) inside the rectangle;
//         S_1[j] = R[j] & amask, S_{k+1}[j] = (S_k[j-1] << 1) & R[j] & amask, in place with j descending.
//         The last non-empty pass gives the longest length; its smallest set bit, then smallest j, is the block.
//   a stack of rectangles (their a-ranges are disjoint and non-empty: at most 64 entries)
template <int W> struct FzRoWord { typedef uint32_t T; };
template <> struct FzRoWord<2> { typedef uint64_t T; };
__device__ __forceinline__ uint32_t fz_ro_ctz(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
__device__ __forceinline__ uint32_t fz_ro_ctz(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }

template <int W, bool WIDE>
__global__ __launch_bounds__(256) void fz_ro_matches(FzScoreParams p)
{
    using SY = FzSym<WIDE>;
    typedef typename FzRoWord<W>::T T;
    constexpr uint32_t RS = fz_peq_stride(W);
    constexpr uint32_t NMAX = 64u;                                // longest pattern / candidate taken
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t *s_peq = lds;                                        // [peq_rows][RS]
    uint32_t *m32 = s_peq + p.peq_rows * RS;                      // WIDE: symbol -> row map
    const uint16_t *s_map = reinterpret_cast<const uint16_t *>(m32);
    uint32_t head = p.peq_rows * RS + (WIDE ? (p.alphabet + 1u) / 2u : 0u);
    head = (head + 1u) & ~1u;                                     // the per-wave arrays hold 64-bit words
    const FzQuery q = p.queries[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < q.peq_rows * (uint32_t)W; i += blockDim.x) s_peq[(i / (uint32_t)W) * RS + i % (uint32_t)W] = p.qblob[q.peq_off + i];
    if (WIDE)
        for (uint32_t i = tid; i < (p.alphabet + 1u) / 2u; i += blockDim.x) m32[i] = p.qblob[q.map_off + i];
    __syncthreads();
    constexpr uint32_t WAVE_WORDS = 64u * (2u * NMAX * (uint32_t)(sizeof(T) / 4u) + NMAX);
    T *s_R = reinterpret_cast<T *>(lds + head + wave * WAVE_WORDS) + lane;          // [j][lane]
    T *s_S = s_R + NMAX * 64u;
    uint32_t *s_st = reinterpret_cast<uint32_t *>(s_S + NMAX * 64u - lane) + lane;  // [depth][lane]
    const uint32_t m = q.m;

    uint32_t t_end = (blockIdx.x + 1u) * p.tiles_per_block;
    if (t_end > p.ntiles) t_end = p.ntiles;
    for (uint32_t tix = blockIdx.x * p.tiles_per_block + wave; tix < t_end; tix += p.nwaves) {
        const uint32_t tile = p.tile_list ? p.tile_list[tix] : p.tile_base + tix;
        const FzTile td = p.tiles[tile];
        const uint32_t clen = td.ntok ? p.tokend[td.toff + lane] : 0u;     // raw view: one slot = the length
        const uint32_t *symp = p.sym + td.coff + lane;
        const bool taken = m <= 32u * (uint32_t)W && m <= NMAX && clen <= NMAX;
        const uint32_t ndw = (uint32_t)td.ndw < NMAX / (uint32_t)SY::SPD ? (uint32_t)td.ndw : NMAX / (uint32_t)SY::SPD;
        for (uint32_t d = 0; d < ndw; ++d) {
            const uint32_t cw = symp[d * 64u];
#pragma unroll
            for (int k = 0; k < SY::SPD; ++k) {
                const uint32_t j = d * (uint32_t)SY::SPD + (uint32_t)k;
                const uint32_t s = (cw >> (SY::BITS * k)) & SY::MASK;
                const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                T r = (T)row[0];
                if (W == 2) r |= (T)((uint64_t)row[W - 1] << 32);
                s_R[j * 64u] = j < clen ? r : (T)0;                       // padding symbols match nothing
            }
        }
        uint32_t M = 0u, sp = 0u;
        if (taken && m != 0u && clen != 0u) { s_st[0] = m << 8 | clen << 24; sp = 1u; }     // {alo, ahi, blo, bhi} bytes
        while (sp != 0u) {
            --sp;
            const uint32_t rect = s_st[sp * 64u];
            const uint32_t alo = rect & 0xFFu, ahi = (rect >> 8) & 0xFFu, blo = (rect >> 16) & 0xFFu, bhi = rect >> 24;
            const T ones = ~(T)0;
            const T amask = (ahi >= 8u * sizeof(T) ? ones : (((T)1 << ahi) - (T)1)) & ~(((T)1 << alo) - (T)1);
            T any = 0;
            uint32_t bi = 0xFFFFFFFFu, bj = 0u;                            // end of the best block so far
            for (uint32_t j = bhi; j-- > blo;) {
                const T sj = s_R[j * 64u] & amask;
                s_S[j * 64u] = sj;
                any |= sj;
                if (sj != (T)0) { const uint32_t i = fz_ro_ctz(sj); if (i <= bi) { bi = i; bj = j; } }
            }
            if (any == (T)0) continue;
            uint32_t k = 1u;
            for (;;) {
                T any2 = 0;
                uint32_t ci = 0xFFFFFFFFu, cj = 0u;
                for (uint32_t j = bhi; j-- > blo + k;) {
                    const T t = (T)(s_S[(j - 1u) * 64u] << 1) & s_R[j * 64u] & amask;
                    s_S[j * 64u] = t;
                    any2 |= t;
                    if (t != (T)0) { const uint32_t i = fz_ro_ctz(t); if (i <= ci) { ci = i; cj = j; } }
                }
                if (any2 == (T)0) break;
                ++k; bi = ci; bj = cj;
            }
            M += k;
            const uint32_t ia = bi + 1u - k, jb = bj + 1u - k;             // where the block starts
            if (alo < ia && blo < jb) { s_st[sp * 64u] = alo | ia << 8 | blo << 16 | jb << 24; ++sp; }
            if (ia + k < ahi && jb + k < bhi) { s_st[sp * 64u] = (ia + k) | ahi << 8 | (jb + k) << 16 | bhi << 24; ++sp; }
        }
        reinterpret_cast<uint16_t *>(p.scores)[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = taken ? (uint16_t)M : (uint16_t)0xFFFFu;
    }
}

// ---------------------------------------------------------------------------------------------
// partial_ratio kernel (thefuzz.fuzz.partial_ratio, reference app.py:1023)
// ---------------------------------------------------------------------------------------------
// rapidfuzz partial_ratio: with `s` the shorter and `l` the longer string, the best ratio(s, w) over
// the windows w of l: every l[i:i+|s|], every proper prefix l[:i] and suffix l[i:] shorter than |s|
// (equal lengths: both directions).  One candidate per lane, the query's Peq in LDS as in the general
// kernel; the pattern is always the QUERY, the text always the CANDIDATE (LCS is symmetric):
//   A  query is the needle (m <= clen): pass i runs the text window [i, i+m) from a fresh state; the
//      padding symbols beyond a shorter candidate are no-ops, so the same uniform loop also yields
//      the suffix windows.  Pass 0 evaluates the prefix windows after every step.
//   B  candidate is the needle (clen <= m): pass i scans the whole candidate against the pattern
//      with the bits below i masked off; the zeros of the state in bits [i, e) then count
//      LCS(candidate, query[i:e]) (Hyyro: bit j is 0 iff row j added one to the LCS), which is the
//      window [i, min(i+clen, m)); pass 0 gives every prefix window the same way.
// The best window is the one with the largest LCS / (|s| + |w|), compared by cross-multiplication;
// one exact rounding at the end ((1 - d/L) * 100, the ratio() formula).
__device__ __forceinline__ void fz_partial_consider(uint32_t l, uint32_t L, uint32_t &bl, uint32_t &bL)
{
    if (l * bL > bl * L) { bl = l; bL = L; }
}

template <int W, bool WIDE>
__global__ __launch_bounds__(256) void fz_score_partial(FzScoreParams p)
{
    using SY = FzSym<WIDE>;
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr uint32_t RS = fz_peq_stride(W);
    uint32_t *s_peq = lds;                                    // [peq_rows][RS]
    // same LDS layout as fz_score_general (fz_score_lds_bytes): the map sits behind the mask arrays
    uint32_t *m32 = s_peq + p.peq_rows * RS + 3u * p.pmax + p.nwaves * 64u * ((uint32_t)W + p.ntkw);
    const uint16_t *s_map = reinterpret_cast<const uint16_t *>(m32);
    const FzQuery q = p.queries[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < q.peq_rows * (uint32_t)W; i += blockDim.x) s_peq[(i / (uint32_t)W) * RS + i % (uint32_t)W] = p.qblob[q.peq_off + i];
    if (WIDE)
        for (uint32_t i = tid; i < (p.alphabet + 1u) / 2u; i += blockDim.x) m32[i] = p.qblob[q.map_off + i];
    __syncthreads();
    const uint32_t m = q.m;

    uint32_t t_end = (blockIdx.x + 1u) * p.tiles_per_block;
    if (t_end > p.ntiles) t_end = p.ntiles;
    for (uint32_t tix = blockIdx.x * p.tiles_per_block + wave; tix < t_end; tix += p.nwaves) {
        const uint32_t tile = p.tile_list ? p.tile_list[tix] : p.tile_base + tix;
        const FzTile td = p.tiles[tile];
        const uint32_t clen = td.ntok ? p.tokend[td.toff + lane] : 0u;   // raw view: one slot = the length
        const uint32_t maxlen = (uint32_t)td.ndw * (uint32_t)SY::SPD;
        const uint32_t *symp = p.sym + td.coff + lane;
        const bool valid = m != 0u && clen != 0u;
        const bool doA = valid && m <= clen, doB = valid && clen <= m;
        uint32_t bl = 0u, bL = 1u;                             // best window so far: LCS bl, length sum bL

        if (__ballot(doA) != 0ull) {
            // ---- A: windows of the candidate, needle = query ---------------------------------
            for (uint32_t i = 0; i < maxlen; ++i) {
                if (__ballot(doA && i < clen) == 0ull) break;
                uint32_t V[W];
#pragma unroll
                for (int w = 0; w < W; ++w) V[w] = 0xFFFFFFFFu;
                uint32_t jend = i + m;
                if (jend > maxlen) jend = maxlen;
                uint32_t cw = 0u;
                for (uint32_t j = i; j < jend; ++j) {
                    if (j == i || (j % (uint32_t)SY::SPD) == 0u) cw = symp[(j / (uint32_t)SY::SPD) * 64u];
                    const uint32_t s = (cw >> ((uint32_t)SY::BITS * (j % (uint32_t)SY::SPD))) & SY::MASK;
                    const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                    uint64_t carry = 0ull;
                    V[0] = fz_lcs_word<true>(V[0], row[0], 0xFFFFFFFFu, carry);
#pragma unroll
                    for (int w = 1; w < W; ++w) V[w] = fz_lcs_word<false>(V[w], row[w], 0xFFFFFFFFu, carry);
                    if (i == 0u && j + 1u < m) {               // proper prefix window l[:j+1]
                        uint32_t l = 0u;
#pragma unroll
                        for (int w = 0; w < W; ++w) l += (uint32_t)__builtin_popcount(~V[w]);
                        if (doA) fz_partial_consider(l, m + j + 1u, bl, bL);
                    }
                }
                if (doA && i < clen) {
                    uint32_t l = 0u;
#pragma unroll
                    for (int w = 0; w < W; ++w) l += (uint32_t)__builtin_popcount(~V[w]);
                    const uint32_t wl = clen - i < m ? clen - i : m;
                    fz_partial_consider(l, m + wl, bl, bL);
                }
            }
        }

        if (__ballot(doB) != 0ull) {
            // ---- B: windows of the query, needle = candidate ---------------------------------
            for (uint32_t i = 0; i < m; ++i) {
                uint32_t V[W];
#pragma unroll
                for (int w = 0; w < W; ++w) V[w] = 0xFFFFFFFFu;
                for (uint32_t d = 0; d < td.ndw; ++d) {
                    const uint32_t cw = symp[d * 64u];
#pragma unroll
                    for (int k = 0; k < SY::SPD; ++k) {
                        const uint32_t s = (cw >> (SY::BITS * k)) & SY::MASK;
                        const uint32_t *row = s_peq + (WIDE ? (uint32_t)s_map[s] : s) * RS;
                        uint64_t carry = 0ull;
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            // pattern bits below i are masked off: they belong to no window of this pass
                            const uint32_t lo = (uint32_t)w * 32u;
                            const uint32_t keep = i <= lo ? 0xFFFFFFFFu : (i >= lo + 32u ? 0u : (0xFFFFFFFFu << (i - lo)));
                            V[w] = w == 0 ? fz_lcs_word<true>(V[w], row[w], keep, carry) : fz_lcs_word<false>(V[w], row[w], keep, carry);
                        }
                    }
                }
                if (doB) {
                    const uint32_t e = i + clen < m ? i + clen : m;       // window query[i:e]
                    uint32_t l = 0u;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const uint32_t lo = (uint32_t)w * 32u;
                        const uint32_t ge = i <= lo ? 0xFFFFFFFFu : (i >= lo + 32u ? 0u : (0xFFFFFFFFu << (i - lo)));
                        const uint32_t lt = e >= lo + 32u ? 0xFFFFFFFFu : (e <= lo ? 0u : ((1u << (e - lo)) - 1u));
                        l += (uint32_t)__builtin_popcount(~V[w] & ge & lt);
                    }
                    fz_partial_consider(l, clen + (e - i), bl, bL);
                }
                if (i == 0u) {
                    // proper prefix windows query[:j], j = 1 .. clen-1: running count of the zeros
                    uint32_t cnt = 0u;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        if ((uint32_t)w * 32u < m) {
                            const uint32_t z = ~V[w];
                            for (uint32_t b = 0; b < 32u; ++b) {
                                cnt += (z >> b) & 1u;
                                const uint32_t j = (uint32_t)w * 32u + b + 1u;
                                if (doB && j < clen) fz_partial_consider(cnt, clen + j, bl, bL);
                            }
                        }
                    }
                }
            }
        }

        uint32_t sc;
        if (!valid) sc = (m == 0u && clen == 0u) ? 100u : 0u;
        else sc = fz_round_ratio(bL - 2u * bl, bL, p.ties);
        if (p.emit.lim) fz_emit(p.emit, q.out_row, p.emit.lim[q.out_row], sc, p.emit.orig_of_phys[tile * 64u + lane]);
        else p.scores[(uint64_t)q.out_row * p.row_stride + (uint64_t)tile * 64u + lane] = (uint8_t)sc;
    }
}


// ---------------------------------------------------------------------------------------------
// stream scoring kernel: short candidates as per-lane PATTERNS, queries streamed as wave-uniform TEXT
// ---------------------------------------------------------------------------------------------
// LCS is symmetric, so the roles of the general kernel are swapped here: one wave owns one tile, every
// lane keeps ITS candidate's match masks (Peq, one row per text symbol that any query of the launch
// uses) in a private LDS column, and the queries of the block pass by as a stream of 16-byte records
// that is the same for all lanes (FzStreamParams, fz_common.h).  Per text symbol that costs one
// conflict-free ds_read_b32 at (lane column + row) and the three-instruction Hyyro update; the
// candidate side is built once per tile and reused for every query of the block.
//   token intersection: brute-force id compares, candidate ids in VGPRs against the record's ids;
//     a candidate token that the query also has is cleared from the pattern keep-mask M, a query
//     token that the candidate also has is scanned with an all-zero mask (it can match nothing).
//   separator: every token of either side is followed by one separator; its
//     match mask lives in a register, so the step after a token needs no lookup.
//   score: T[L][d >> 1] = the rounded ratio (a distance has the parity of its length sum, so half
//     the columns suffice), an 8 KiB table the host fills with the scorer's own rounding
//     (fz_host.cpp make_score_table); both ratios of token_set_ratio are lookups.
// The table is indexed [lane][row] with an odd stride: lanes hit distinct banks for any row.
// base + byte K of a wave-uniform word in ONE vector instruction (SDWA byte select on the scalar
// operand).  The record's symbol bytes would otherwise each cost a scalar s_bfe before the v_add, and
// the scalar unit is the busiest one in this kernel (profiles/r01_d_*: 0.78 SALU per CU-cycle).
template <int K>
__device__ __forceinline__ uint32_t fz_add_byte(uint32_t word, uint32_t base)
{
    uint32_t r;
    if constexpr (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "s"(word), "v"(base));
    if constexpr (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "s"(word), "v"(base));
    if constexpr (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "s"(word), "v"(base));
    if constexpr (K == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "s"(word), "v"(base));
    return r;
}

// LDS byte address of a pointer into LDS, and a dword load from such an address.  With the address as
// a plain integer the SDWA add above IS the address computation (no generic-pointer arithmetic left
// for the compiler to redo).  The sanitizer build swaps the bodies for host ones (tests/emul).
__device__ __forceinline__ uint32_t fz_lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;   // EMUL: return (uint32_t)((const unsigned char *)p - emul::g_lds);
}
__device__ __forceinline__ uint32_t fz_lds_load(uint32_t addr)
{
    return *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)addr;     // EMUL: return *(const uint32_t *)(emul::g_lds + addr);
}

// a + b + c in ONE lane operation with c in a scalar register (left to itself the compiler hoists a + c
// out of the dependency chain as a second instruction)
__device__ __forceinline__ uint32_t fz_add3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));   // EMUL: r = a + b + c;
    return r;
}

// One Hyyro step of the packed kernel as ONE asm block: u = v & row & mk; v = (v + u + negstop) | (v ^ u).
// Left as separate statements the compiler pads every consumer of an inline-asm result with an s_nop (it
// cannot see that the asm holds a plain v_add3), and a nop costs an issue slot like any other instruction.
__device__ __forceinline__ uint32_t fz_pk_step(uint32_t v, uint32_t row, uint32_t mk, uint32_t negstop)
{
    uint32_t u, t;
    asm("v_bitop3_b32 %1, %0, %3, %4 bitop3:0x80\n\tv_add3_u32 %2, %0, %1, %5\n\tv_bitop3_b32 %0, %2, %0, %1 bitop3:0xf6" : "+v"(v), "=&v"(u), "=&v"(t) : "v"(row), "v"(mk), "s"(negstop));   // EMUL: u = v & row & mk; t = v + u + negstop; v = t | (v ^ u);
    return v;
}

// a byte of the rounded-score table: index masked to the table and ORed onto its LDS address in one
// v_and_or (the table is the first thing in LDS, so its address is a multiple of its size: checked once
// per block by the packed kernel)
__device__ __forceinline__ uint32_t fz_tab_load(uint32_t tab_off, uint32_t idx)
{
    const uint32_t a = (idx & (FZ_STREAM_TAB_BYTES - 1u)) | tab_off;
    return *(__attribute__((address_space(3))) const uint8_t *)(uintptr_t)a;   // EMUL: return *(const uint8_t *)(emul::g_lds + a);
}

// ---- loads the compiler must keep scalar --------------------------------------------------------------
// The record stream and the row limits are written before the launch and only read by it.  A load through
// the CONSTANT address space is invariant to the compiler, so a wave-uniform one stays a scalar load
// (s_load_dwordx4) wherever it stands.  Through a plain pointer the EMIT kernels, whose list appends are
// atomics, turned every record load after the first query's epilogue into global_load_dwordx4 +
// s_waitcnt vmcnt(0) + 4 v_readfirstlane: a vector load the wave waits for on the spot, per record.
// The byte offset stays 32-bit so that the address is base + offset in the load itself.
__device__ __forceinline__ FzRec fz_rec_load(const FzRec *recs, uint32_t byte_off)
{
    typedef uint32_t fz_u32x4 __attribute__((ext_vector_type(4)));   // EMUL: ;
    const fz_u32x4 v = *(const __attribute__((address_space(4))) fz_u32x4 *)((uintptr_t)recs + byte_off);   // EMUL: const FzRec v = *(const FzRec *)((const char *)recs + byte_off);
    return FzRec{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ uint64_t fz_lim_load(const uint64_t *lim, uint32_t row)
{
    return *(const __attribute__((address_space(4))) uint64_t *)((uintptr_t)lim + 8ull * row);   // EMUL: return lim[row];
}

// ---- lane masks: a vector compare leaves one bit per lane in a scalar register pair -------------------
// Scalar ORs combine them (no lane operation per compare), and two single instructions bring a mask back
// into the lanes: a select, and "shift the lane's bit into a per-lane bit string" (v_addc: x + x + carry-in).
__device__ __forceinline__ uint64_t fz_eq_mask(uint32_t a, uint32_t b) { return __ballot(a == b); }

__device__ __forceinline__ uint32_t fz_shift_in(uint32_t bits, uint64_t mask)
{
    asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(mask) : "vcc");   // EMUL: bits = (bits << 1) | (uint32_t)((mask >> (threadIdx.x & 63u)) & 1ull);
    return bits;
}

// bit `pos` (a scalar) of x, spread over the whole word: one v_bfe_i32
__device__ __forceinline__ uint32_t fz_bit_spread(uint32_t x, uint32_t pos)
{
    return (uint32_t)__builtin_amdgcn_sbfe((int)x, pos, 1u);   // EMUL: return 0u - ((x >> pos) & 1u);
}

// this lane's bit of `mask` set ? a : all ones
__device__ __forceinline__ uint32_t fz_ones_unless(uint64_t mask, uint32_t a)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, -1, %1, %2" : "=v"(r) : "v"(a), "s"(mask));   // EMUL: r = ((mask >> (threadIdx.x & 63u)) & 1ull) ? a : 0xFFFFFFFFu;
    return r;
}

template <int W, int STRIDE>
__device__ __forceinline__ void fz_stream_step(uint32_t (&V)[W], const uint32_t (&row)[W], const uint32_t (&Mk)[W])
{
    if (W == 1) {
        const uint32_t v = V[0];
        const uint32_t u = __builtin_amdgcn_bitop3_b32(v, row[0], Mk[0], 0x80);        // v & row & Mk
        V[0] = __builtin_amdgcn_bitop3_b32(v + u, v, u, 0xF6);                         // (v + u) | (v ^ u)
    } else {
        const uint32_t v0 = V[0], v1 = V[W - 1];
        const uint32_t u0 = __builtin_amdgcn_bitop3_b32(v0, row[0], Mk[0], 0x80);
        const uint32_t u1 = __builtin_amdgcn_bitop3_b32(v1, row[W - 1], Mk[W - 1], 0x80);
        const uint64_t a = (((uint64_t)v1 << 32) | v0) + (((uint64_t)u1 << 32) | u0);
        V[0] = __builtin_amdgcn_bitop3_b32((uint32_t)a, v0, u0, 0xF6);
        V[W - 1] = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), v1, u1, 0xF6);
    }
}

// The query loop of one tile, specialised on the tile's token slots NT (padding slots hold FZ_TOK_NONE,
// which no query id equals).  Record stream of one query (fz_common.h, "stream records"):
//   header   {out_row, m | n_ids << 8 | n_token_records << 16, id0, id1}
//   ids      {id2..id5}, {id6..id9}, ... while n_ids says so: the query tokens the corpus vocabulary knows
//   tokens   {4 row bytes, 4 row bytes, 4 row bytes, meta}: up to 12 text symbols of one token, padded with
//            row 0 (all zero in every column: a step on it changes nothing) up to 4, 8 or 12 steps
// Phase 1 compares every query id with the NT candidate ids (lane masks, fz_eq_mask): a candidate token the
// query also has leaves the pattern (its range is cleared in M), and whether the CANDIDATE has query token j
// is shifted into a per-lane bit string, one bit per known query token, in token order.  Phase 2 scans the
// tokens: every token record names its token's bit of that string (a token without an id names a bit that is
// always zero) and is scanned with an all-zero mask when the bit is set (a token both sides have is in
// neither difference) -- one v_bfe_i32 per record, no branch on "known" or "first".  No lane operation depends
// on the number of symbols: the only scalar branches of a token are its step class and its last-record flag.
template <int W, int MODE, int STRIDE, int NT, bool EMIT>
__device__ __forceinline__ void fz_stream_queries(const FzStreamParams &p, const FzRec *__restrict__ recs,
                                                  uint8_t *__restrict__ scores, const uint8_t *s_tab,
                                                  uint32_t col_off, const uint32_t (&cid)[FZ_STREAM_MAX_TOK],
                                                  const uint32_t (&Rm)[FZ_STREAM_MAX_TOK][W],
                                                  const uint32_t (&psep)[W], uint32_t clen, uint32_t out_off,
                                                  uint32_t orig, uint32_t q0, uint32_t q1)
{
    constexpr uint32_t PLANE = 64u * (uint32_t)STRIDE * 4u;
    if (q0 >= q1) return;
    // ONE record is in flight: a consumer copies what it needs of `cur` and then asks for the next record
    // (no cur = nxt rotation of two records: fewer scalar moves, and measured 1.5 % faster; the scalar
    // cache answers quickly enough that a deeper prefetch buys nothing).  The stream ends with padding records.
    uint32_t ro = p.qrec[q0] * (uint32_t)sizeof(FzRec);               // byte offset of the current record
    FzRec cur = fz_rec_load(recs, ro);
#define FZ_ADVANCE() do { ro += 16u; cur = fz_rec_load(recs, ro); } while (0)

    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t out_row = cur.x, m = cur.y & 0xFFu;
        const uint32_t n_ids = (cur.y >> 8) & 0xFFu, nrec = cur.y >> 16;
        const uint32_t id0 = cur.z, id1 = cur.w;
        FZ_ADVANCE();
        const uint64_t lim = EMIT ? fz_lim_load(p.emit.lim, out_row) : 0ull;   // scalar load, used at the end of the query

        uint32_t M[W], has = 0u;                         // has: bit string "the candidate has query token j"
#pragma unroll
        for (int w = 0; w < W; ++w) M[w] = 0xFFFFFFFFu;
        if (MODE == FZ_MODE_TOKEN_SET && n_ids != 0u) {
            uint64_t hit[NT];
            {                                                         // id 0: the masks start here
                uint64_t any = 0ull;
#pragma unroll
                for (int i = 0; i < NT; ++i) { hit[i] = fz_eq_mask(cid[i], id0); any |= hit[i]; }
                has = fz_shift_in(has, any);
            }
#define FZ_NEXT_ID(sj) do { uint64_t any_ = 0ull; _Pragma("unroll") for (int i = 0; i < NT; ++i) { const uint64_t c_ = fz_eq_mask(cid[i], (sj)); hit[i] |= c_; any_ |= c_; } \
                            has = fz_shift_in(has, any_); } while (0)
            if (n_ids > 1u) {
                FZ_NEXT_ID(id1);
                for (uint32_t j = 2u; j < n_ids; j += 4u) {
                    const uint32_t ia = cur.x, ib = cur.y, ic = cur.z, id = cur.w;
                    FZ_ADVANCE();
                    FZ_NEXT_ID(ia);
                    if (j + 1u < n_ids) { FZ_NEXT_ID(ib);
                    if (j + 2u < n_ids) { FZ_NEXT_ID(ic);
                    if (j + 3u < n_ids) { FZ_NEXT_ID(id); } } }
                }
            }
#undef FZ_NEXT_ID
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int w = 0; w < W; ++w) M[w] &= fz_ones_unless(hit[i], ~Rm[i][w]);
        }

        uint32_t V[W], Mk[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { V[w] = 0xFFFFFFFFu; Mk[w] = M[w]; }
        for (uint32_t r = 0; r < nrec; ++r) {
            // token record {symbols 0-3, symbols 4-7, symbols 8-11, class | last << 3 | first << 4 | 2 * tpos << 8}
            const uint32_t s0 = cur.x, s1 = cur.y, s2 = cur.z, meta = cur.w;
            if (MODE == FZ_MODE_TOKEN_SET) {
                // the token's bit of the lane string, spread over the word: all ones when both sides have the
                // token, which is then scanned with an all-zero mask.  No branch: a token without an id reads
                // a bit that is always zero, a continuation record repeats its token's position.
                const uint32_t t = fz_bit_spread(has, (meta >> 9) & 15u);
#pragma unroll
                for (int w = 0; w < W; ++w) Mk[w] = M[w] & ~t;
            }
            // LDS rows of this lane's column: column address + row byte (one SDWA add each).  All rows of the
            // record are in flight before its first step: the empty asm pins them there, and its memory
            // clobber keeps the next record's scalar load BELOW the wait, so that the load overlaps this
            // record's steps instead of being waited for at once.
            uint32_t row[8][W];
#define FZ_ROWS4(g, word) do { const uint32_t o_[4] = {fz_add_byte<0>((word), col_off), fz_add_byte<1>((word), col_off), fz_add_byte<2>((word), col_off), fz_add_byte<3>((word), col_off)}; \
                               _Pragma("unroll") for (int k = 0; k < 4; ++k) _Pragma("unroll") for (int w = 0; w < W; ++w) row[4 * (g) + k][w] = fz_lds_load(o_[k] + (uint32_t)w * PLANE); } while (0)
#define FZ_PIN4(g) do { _Pragma("unroll") for (int k = 0; k < 4; ++k) _Pragma("unroll") for (int w = 0; w < W; ++w) asm volatile("" : "+v"(row[4 * (g) + k][w]) : : "memory"); } while (0)
#define FZ_STEPS4(g) do { _Pragma("unroll") for (int k = 0; k < 4; ++k) fz_stream_step<W, STRIDE>(V, row[4 * (g) + k], Mk); } while (0)
            const uint32_t cls = meta & 3u;
            if (cls == 0u) {
                FZ_ROWS4(0, s0); FZ_PIN4(0);
                FZ_ADVANCE();
                FZ_STEPS4(0);
            } else if (cls == 1u) {
                FZ_ROWS4(0, s0); FZ_ROWS4(1, s1); FZ_PIN4(0); FZ_PIN4(1);
                FZ_ADVANCE();
                FZ_STEPS4(0); FZ_STEPS4(1);
            } else {
                // 9 to 12 symbols (rare): eight rows in flight, then the last four into the first slots again
                // (twelve at once cost the registers that keep 8 waves per SIMD from fitting)
                FZ_ROWS4(0, s0); FZ_ROWS4(1, s1); FZ_PIN4(0); FZ_PIN4(1);
                FZ_ADVANCE();
                FZ_STEPS4(0); FZ_STEPS4(1);
                FZ_ROWS4(0, s2); FZ_PIN4(0);
                FZ_STEPS4(0);
            }
#undef FZ_ROWS4
#undef FZ_PIN4
#undef FZ_STEPS4
            if (MODE == FZ_MODE_TOKEN_SET && (meta & 8u)) fz_stream_step<W, STRIDE>(V, psep, Mk);
        }

        uint32_t lcs = 0u;
#pragma unroll
        for (int w = 0; w < W; ++w) lcs += (uint32_t)__builtin_popcount(~V[w]);
        uint32_t sc;
        if (MODE == FZ_MODE_TOKEN_SET) {
            // with sect1 = sum(len+1) over the common tokens, PA = m - sect1, TB = clen - sect1:
            //   r0 = T[m + clen - 2][PA + TB - 2*lcs]              (the joined differences)
            //   r1 = T[2*sect1 - 2 + min(PA, TB)][min(PA, TB)]     (intersection vs intersection + shorter
            //        difference; 100 when one side is contained in the other), used when sect1 > 0
            uint32_t sect1 = 0u;
#pragma unroll
            for (int w = 0; w < W; ++w) sect1 += (uint32_t)__builtin_popcount(~M[w]);
            const uint32_t mc = m + clen;
            const uint32_t d0 = mc - 2u * (sect1 + lcs);
            const uint32_t r0 = s_tab[(((mc - 2u) << 6) + (d0 >> 1)) & (FZ_STREAM_TAB_BYTES - 1u)];
            const uint32_t d1 = (m < clen ? m : clen) - sect1;
            const uint32_t r1 = s_tab[(((2u * sect1 - 2u + d1) << 6) + (d1 >> 1)) & (FZ_STREAM_TAB_BYTES - 1u)];
            sc = (sect1 != 0u && r1 > r0) ? r1 : r0;
            if (clen == 0u) sc = 0u;                                  // a candidate without tokens scores 0
        } else {
            const uint32_t L = m + clen;
            sc = s_tab[((L << 6) + ((L - 2u * lcs) >> 1)) & (FZ_STREAM_TAB_BYTES - 1u)];   // T[0][0] = 100: two empty strings
        }
        if (EMIT) {
            // fused threshold + top-k: no score matrix; a lane below the limit's score is done here
            if (__ballot(sc >= (uint32_t)(lim >> FZ_IDX_BITS)) != 0ull) fz_emit(p.emit, out_row, lim, sc, orig);
        } else {
            uint8_t *rowp = scores + (uint64_t)out_row * (uint64_t)p.row_stride;     // wave-uniform
            rowp[out_off] = (uint8_t)sc;
        }
    }
#undef FZ_ADVANCE
}

// item -> (query group, unit): items are numbered slab-major.  Slab s holds the units [s*S, min((s+1)*S, units))
// and its items come group after group: item = s*S*groups + group*size(s) + (unit - s*S).  All scalars.
__device__ __forceinline__ void fz_item_decode(uint32_t item, uint32_t units, uint32_t S, uint32_t groups,
                                               uint32_t &qgroup, uint32_t &unit)
{
    const uint32_t per_slab = S * groups;
    const uint32_t slab = item / per_slab, r = item - slab * per_slab;
    const uint32_t first = slab * S;
    const uint32_t size = units - first < S ? units - first : S;       // the last slab may be shorter
    qgroup = r / size;
    unit = first + (r - qgroup * size);
}

// ---- work queues ----------------------------------------------------------------------------------------
// One BLOCK of G waves owns one unit (a tile, or a pair of tiles) at a time: the unit's match masks live in ONE
// LDS column set that all G waves read, and the queries of the item are cut into G runs, one per wave.  Sharing
// the column is what lets 8 waves per SIMD run: a column is 9.5 KB (19 KB with two pattern words), so a wave
// with a column of its own leaves room for 4 (2) waves per SIMD, where one wave issues an instruction every ~7
// cycles and the SIMD idles in between (tools/valu_rate.hip: 1.9 cycles per plain VALU instruction at 4 waves
// per SIMD, 1.4 at 8; measured on this kernel with a timing-only build: score phase 3.27 -> 2.65 ms).
// Items are (query group, unit) pairs numbered slab-major and dealt round-robin to FZ_STREAM_QUEUES queues, each
// behind its own counter (one counter for the whole chip serialises at ~12 ns per draw); wave 0 of a block draws
// `chunk` positions at a time from its home queue and moves on to the next queue when that one is empty.  Exit:
// counters only grow, and a block leaves when a scan finds every queue drained (every wave of the block sees
// the same end marker behind the same barrier).
struct FzDraw { uint32_t queue, pos, pos_end; };

// next item of this block, or 0xFFFFFFFF when every queue is drained; called by ONE whole wave (all 64 lanes)
__device__ __forceinline__ uint32_t fz_draw_item(const FzStreamParams &p, FzDraw &d, uint32_t lane)
{
    for (;;) {
        if (d.pos < d.pos_end) {
            const uint32_t item = d.queue + d.pos * FZ_STREAM_QUEUES;
            ++d.pos;
            return item;
        }
        // queue q holds the items q, q + QUEUES, q + 2*QUEUES, ...: all queues walk the query groups
        // in the same order, so the chip works on the same records at the same time (scalar cache)
        const uint32_t q_len = d.queue < p.nitems ? (p.nitems - d.queue + FZ_STREAM_QUEUES - 1u) / FZ_STREAM_QUEUES : 0u;
        // counters sit FZ_STREAM_QPAD words apart (one cache line each); a plain load first, so
        // that a drained queue costs no atomic
        uint32_t *cnt = p.work_counter + d.queue * FZ_STREAM_QPAD;
        uint32_t got = 0xFFFFFFFFu;
        if (lane == 0u && q_len != 0u && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < q_len)
            got = atomicAdd(cnt, p.chunk);
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (got >= q_len) {
            // this queue is drained: lane l looks at queue l, the wave moves to the next queue that
            // still has positions, or reports the end when there is none (one load round trip, not QUEUES)
            bool open = false;
            if (lane < FZ_STREAM_QUEUES && lane < p.nitems) {
                const uint32_t len_l = (p.nitems - lane + FZ_STREAM_QUEUES - 1u) / FZ_STREAM_QUEUES;
                open = __hip_atomic_load(p.work_counter + lane * FZ_STREAM_QPAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len_l;
            }
            const uint64_t mask = __ballot(open);
            if (mask == 0ull) return 0xFFFFFFFFu;
            const uint64_t ahead = mask >> d.queue;                    // queues at or after this one
            d.queue = ahead != 0ull ? d.queue + (uint32_t)__builtin_ctzll(ahead) : (uint32_t)__builtin_ctzll(mask);
            continue;
        }
        d.pos = got;
        d.pos_end = got + p.chunk < q_len ? got + p.chunk : q_len;
    }
}

// the run of queries wave `wave` of `nw` takes out of group `qgroup`
__device__ __forceinline__ void fz_wave_queries(const FzStreamParams &p, uint32_t qgroup, uint32_t wave, uint32_t nw,
                                                uint32_t &q0, uint32_t &q1)
{
    uint32_t g0 = qgroup * p.q_per_block, g1 = g0 + p.q_per_block;
    if (g1 > p.nq) g1 = p.nq;
    const uint32_t per = (g1 - g0 + nw - 1u) / nw;
    q0 = g0 + wave * per;
    q1 = q0 + per < g1 ? q0 + per : g1;
}

template <int W, int MODE, int STRIDE, bool EMIT>
__global__ __launch_bounds__(512)
__attribute__((amdgpu_waves_per_eu(W == 1 ? 8 : 5, 8)))      // two pattern words: 96 registers, 5 waves per SIMD   // EMUL: /* occupancy hint: no meaning on the host */
void fz_score_stream(FzStreamParams p, const FzRec *__restrict__ recs, uint8_t *__restrict__ scores)
{
    extern __shared__ __align__(16) uint32_t lds[];
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(lds);                   // [128][64] rounded ratios T[L][d >> 1]
    uint8_t *s_row = s_tab + FZ_STREAM_TAB_BYTES;                        // [256] symbol -> 4 * row
    uint8_t *s_peq = s_row + 256u;                                       // [W][64][STRIDE] u32: the block's ONE column set
    constexpr uint32_t PLANE = 64u * (uint32_t)STRIDE * 4u;              // bytes of one word plane
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_peq + (uint32_t)W * PLANE);   // [0]: the block's current item

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t i = tid; i < FZ_STREAM_TAB_BYTES / 16u; i += blockDim.x)
        reinterpret_cast<uint4 *>(s_tab)[i] = reinterpret_cast<const uint4 *>(p.score_tab)[i];
    if (tid < 64u) reinterpret_cast<uint32_t *>(s_row)[tid] = reinterpret_cast<const uint32_t *>(p.row_of_sym)[tid];

    uint8_t *col = s_peq + lane * ((uint32_t)STRIDE * 4u);               // this lane's column (shared by the block's waves)
    const uint32_t col_off = fz_lds_addr(col);
    FzDraw draw{blockIdx.x % FZ_STREAM_QUEUES, 0u, 0u};
    for (;;) {
        __syncthreads();                                 // every wave is done with the column and with s_ctl
        if (wave == 0u) {
            const uint32_t it = fz_draw_item(p, draw, lane);
            if (lane == 0u) s_ctl[0] = it;
        }
        // the column starts from zero: wave w clears the rows w, w + nw, ...
#pragma unroll
        for (int w = 0; w < W; ++w)
            for (uint32_t r = wave; r < (uint32_t)STRIDE; r += nw) *reinterpret_cast<uint32_t *>(col + (uint32_t)w * PLANE + r * 4u) = 0u;
        __syncthreads();                                 // the item is known, the column is clear
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[0]);   // the same for every lane: keep what follows scalar
        if (item == 0xFFFFFFFFu) break;
        uint32_t qgroup, tix;
        fz_item_decode(item, p.ntiles, p.slab_units, (p.nq + p.q_per_block - 1u) / p.q_per_block, qgroup, tix);
        const uint32_t tile = p.tile_list[tix];
        const FzTile td = p.tiles[tile];
        const uint32_t ndw = td.ndw, ntok = td.ntok;

        // ---- candidate side, once per item (lane constants in every wave; the LDS writes are shared out) ----
        uint32_t cid[FZ_STREAM_MAX_TOK], Rm[FZ_STREAM_MAX_TOK][W], prev_end = 0u;
#pragma unroll
        for (int i = 0; i < (int)FZ_STREAM_MAX_TOK; ++i) {
            uint32_t end = prev_end;
            cid[i] = FZ_TOK_NONE;
            if ((uint32_t)i < ntok) {
                if (MODE == FZ_MODE_TOKEN_SET) cid[i] = p.tokid[td.toff + (uint32_t)i * 64u + lane];
                end = p.tokend[td.toff + (uint32_t)i * 64u + lane];
            }
            const uint64_t hi = end >= 64u ? ~0ull : ((1ull << end) - 1ull);
            const uint64_t lo = prev_end >= 64u ? ~0ull : ((1ull << prev_end) - 1ull);
            const uint64_t r = hi & ~lo;                                     // pattern positions of token i
            Rm[i][0] = (uint32_t)r;
            if (W == 2) Rm[i][W - 1] = (uint32_t)(r >> 32);
            prev_end = end;
        }
        const uint32_t clen = prev_end;

        uint32_t psep[W];
#pragma unroll
        for (int w = 0; w < W; ++w) psep[w] = 0u;
#pragma unroll
        for (int d = 0; d < 8 * W; ++d) {
            if ((uint32_t)d < ndw) {
                const uint32_t cw = p.sym[td.coff + (uint32_t)d * 64u + lane];
                const bool mine = ((uint32_t)d % nw) == wave;                // this wave writes dword d's match bits
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t s = (cw >> (8 * k)) & 0xFFu;
                    const uint32_t bit = 1u << ((d * 4 + k) & 31);
                    const int w = (d * 4 + k) >> 5;
                    if (MODE == FZ_MODE_TOKEN_SET && s == FZ_SYM_SPACE) {
                        psep[w] |= bit;
                    } else if (mine) {
                        const uint32_t r4 = s_row[s];                        // 0xFF: no query uses this symbol
                        if (r4 != 0xFFu)
                            __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(col + (uint32_t)w * PLANE + r4), bit,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __syncthreads();                                 // the column is complete

        // ---- this wave's run of the group's queries: a stream of wave-uniform records (scalar loads) ----
        uint32_t q0, q1;
        fz_wave_queries(p, qgroup, wave, nw, q0, q1);
        const uint32_t out_off = tile * 64u + lane;
        const uint32_t orig = EMIT ? p.emit.orig_of_phys[out_off] : 0u;
        if (MODE != FZ_MODE_TOKEN_SET || ntok <= 1u) fz_stream_queries<W, MODE, STRIDE, 1, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
        else if (ntok == 2u) fz_stream_queries<W, MODE, STRIDE, 2, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
        else if (ntok == 3u) fz_stream_queries<W, MODE, STRIDE, 3, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
        else if (ntok == 4u) fz_stream_queries<W, MODE, STRIDE, 4, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
        else if (ntok == 5u) fz_stream_queries<W, MODE, STRIDE, 5, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
        else fz_stream_queries<W, MODE, STRIDE, (int)FZ_STREAM_MAX_TOK, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, orig, q0, q1);
    }
}

// ---------------------------------------------------------------------------------------------
// packed stream kernel: TWO short candidates per lane in one 32-bit pattern word
// ---------------------------------------------------------------------------------------------
// Two tiles whose longest candidates have S and at most 31 - S symbols are taken together: tile A lives
// in bits 0..S-1 of every pattern word of the lane, tile B in bits S+1..31 (the host pairs short tiles
// with long ones and passes S with the pair).  One LDS row read and one Hyyro update then serve two
// candidates.  Bit S is the carry stopper: it is 1 in V and 0 in every match mask, and it is cleared in
// the addend, so the carry chain of A ends there (bit S of the sum is dropped by the OR with V ^ u, whose
// bit S is always 1) and never reaches B; the carry out of bit 31 is discarded by the 32-bit add.  Bits
// of a field beyond its candidate stay 1 in V and in M, so they count neither as matches nor as common
// tokens.  Everything else is the stream kernel above, with the per-candidate pieces doubled.

template <int MODE, int STRIDE, int NT, bool EMIT>
__device__ __forceinline__ void fz_stream_queries_pk(const FzStreamParams &p, const FzRec *__restrict__ recs,
                                                     uint8_t *__restrict__ scores, const uint8_t *s_tab,
                                                     uint32_t col_off, const uint32_t (&cid)[2][FZ_PK_MAX_TOK],
                                                     const uint32_t (&Rm)[2][FZ_PK_MAX_TOK], uint32_t psep,
                                                     const uint32_t (&clen)[2], const uint32_t (&out_off)[2],
                                                     const uint32_t (&fmask)[2], const uint32_t (&fw)[2], uint32_t negstop,
                                                     const uint32_t (&orig)[2], uint32_t q0, uint32_t q1)
{
    if (q0 >= q1) return;
    // Lane constants of the score look-ups (end of the query loop).  With bm = popcount(M & field) and
    // bv = popcount(V & field): sect1 = fw - bm, lcs = fw - bv, and (m + clen) >> 1 = (m >> 1) + (clen >> 1)
    // + (m & clen & 1), so every table index is a per-tile lane constant + a per-query scalar + popcounts.
    // The empty asm keeps the constants as they are (the compiler would re-derive them per query).
    const uint32_t tab_off = fz_lds_addr(s_tab);
    uint32_t c_tab[2], c_odd[2], c_len1[2], nRm[2][NT], fm[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        c_tab[h] = MODE == FZ_MODE_TOKEN_SET ? ((clen[h] - 2u) << 6) + (clen[h] >> 1) - 2u * fw[h]
                                             : (clen[h] << 6) + (clen[h] >> 1) - fw[h];
        c_odd[h] = clen[h] & 1u;
        c_len1[h] = clen[h] + fw[h] - 2u;
        fm[h] = fmask[h];                                 // in a lane register: a scalar operand costs a vector
        asm volatile("" : "+v"(c_tab[h]), "+v"(c_odd[h]), "+v"(c_len1[h]), "+v"(fm[h]));   // instruction an issue cycle
#pragma unroll
        for (int i = 0; i < NT; ++i) nRm[h][i] = ~Rm[h][i];
    }
    // ONE record is in flight: a consumer copies what it needs of `cur` and then asks for the next record
    // (no cur = nxt rotation of two records: fewer scalar moves, and measured 1.5 % faster; the scalar
    // cache answers quickly enough that a deeper prefetch buys nothing).  The stream ends with padding records.
    uint32_t ro = p.qrec[q0] * (uint32_t)sizeof(FzRec);               // byte offset of the current record
    FzRec cur = fz_rec_load(recs, ro);
#define FZ_ADVANCE() do { ro += 16u; cur = fz_rec_load(recs, ro); } while (0)

    for (uint32_t q = q0; q < q1; ++q) {
        // record formats and the two phases: fz_stream_queries above; here every query token leaves TWO
        // bits in the lane's bit string, "tile A's candidate has it" and "tile B's candidate has it"
        const uint32_t out_row = cur.x, m = cur.y & 0xFFu;
        const uint32_t n_ids = (cur.y >> 8) & 0xFFu, nrec = cur.y >> 16;
        const uint32_t id0 = cur.z, id1 = cur.w;
        FZ_ADVANCE();
        const uint64_t lim = EMIT ? fz_lim_load(p.emit.lim, out_row) : 0ull;   // scalar load, used at the end of the query

        uint32_t M = 0xFFFFFFFFu, has = 0u;
        if (MODE == FZ_MODE_TOKEN_SET && n_ids != 0u) {
            uint64_t hit[2][NT];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                             // id 0: the masks start here
                uint64_t any = 0ull;
#pragma unroll
                for (int i = 0; i < NT; ++i) { hit[h][i] = fz_eq_mask(cid[h][i], id0); any |= hit[h][i]; }
                has = fz_shift_in(has, any);
            }
#define FZ_NEXT_ID(sj) do { _Pragma("unroll") for (int h = 0; h < 2; ++h) { uint64_t any_ = 0ull; \
                                _Pragma("unroll") for (int i = 0; i < NT; ++i) { const uint64_t c_ = fz_eq_mask(cid[h][i], (sj)); hit[h][i] |= c_; any_ |= c_; } \
                                has = fz_shift_in(has, any_); } } while (0)
            if (n_ids > 1u) {
                FZ_NEXT_ID(id1);
                for (uint32_t j = 2u; j < n_ids; j += 4u) {
                    const uint32_t ia = cur.x, ib = cur.y, ic = cur.z, id = cur.w;
                    FZ_ADVANCE();
                    FZ_NEXT_ID(ia);
                    if (j + 1u < n_ids) { FZ_NEXT_ID(ib);
                    if (j + 2u < n_ids) { FZ_NEXT_ID(ic);
                    if (j + 3u < n_ids) { FZ_NEXT_ID(id); } } }
                }
            }
#undef FZ_NEXT_ID
            // a candidate token the query also has leaves the pattern: its range is cleared in M
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < NT; ++i) M &= fz_ones_unless(hit[h][i], nRm[h][i]);
        }

        uint32_t V = 0xFFFFFFFFu, Mk = M;
        for (uint32_t r = 0; r < nrec; ++r) {
            const uint32_t s0 = cur.x, s1 = cur.y, s2 = cur.z, meta = cur.w;
            // the mask of the token (issued between the row reads and the wait for them): a token both sides
            // have is scanned with a zero mask, per half: Mk = M & ~(tA & fieldA) & ~(tB & fieldB)
#define FZ_MK() do { if (MODE == FZ_MODE_TOKEN_SET) { const uint32_t pos_ = (meta >> 8) & 31u; \
                         const uint32_t tA = fz_bit_spread(has >> 1, pos_), tB = fz_bit_spread(has, pos_); \
                         Mk = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(M, tA, fm[0], 0x70), tB, fm[1], 0x70); } } while (0)
            uint32_t row[8];
#define FZ_ROWS4(g, word) do { const uint32_t o_[4] = {fz_add_byte<0>((word), col_off), fz_add_byte<1>((word), col_off), fz_add_byte<2>((word), col_off), fz_add_byte<3>((word), col_off)}; \
                               _Pragma("unroll") for (int k = 0; k < 4; ++k) row[4 * (g) + k] = fz_lds_load(o_[k]); } while (0)
#define FZ_PIN4(g) do { _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(row[4 * (g) + k]) : : "memory"); } while (0)
#define FZ_PK_STEP(rw) do { V = fz_pk_step(V, (rw), Mk, negstop); } while (0)
#define FZ_STEPS4(g) do { FZ_PK_STEP(row[4 * (g)]); FZ_PK_STEP(row[4 * (g) + 1]); FZ_PK_STEP(row[4 * (g) + 2]); FZ_PK_STEP(row[4 * (g) + 3]); } while (0)
            const uint32_t cls = meta & 3u;
            if (cls == 0u) {
                FZ_ROWS4(0, s0); FZ_MK(); FZ_PIN4(0);
                FZ_ADVANCE();
                FZ_STEPS4(0);
            } else if (cls == 1u) {
                FZ_ROWS4(0, s0); FZ_ROWS4(1, s1); FZ_MK(); FZ_PIN4(0); FZ_PIN4(1);
                FZ_ADVANCE();
                FZ_STEPS4(0); FZ_STEPS4(1);
            } else {
                // 9 to 12 symbols (rare): eight rows in flight, then the last four into the first slots again
                FZ_ROWS4(0, s0); FZ_ROWS4(1, s1); FZ_MK(); FZ_PIN4(0); FZ_PIN4(1);
                FZ_ADVANCE();
                FZ_STEPS4(0); FZ_STEPS4(1);
                FZ_ROWS4(0, s2); FZ_PIN4(0);
                FZ_STEPS4(0);
            }
            if (MODE == FZ_MODE_TOKEN_SET && (meta & 8u)) FZ_PK_STEP(psep);
#undef FZ_MK
#undef FZ_ROWS4
#undef FZ_PIN4
#undef FZ_STEPS4
#undef FZ_PK_STEP
        }

        const uint32_t qs = (m << 6) + (m >> 1), m_odd = 0u - (m & 1u);          // scalars
        uint32_t sc2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t bv = (uint32_t)__builtin_popcount(V & fm[h]);
            const uint32_t both_odd = c_odd[h] & m_odd;
            uint32_t sc;
            if (MODE == FZ_MODE_TOKEN_SET) {
                //   r0 = T[m + clen - 2][(m + clen - 2*(sect1 + lcs)) >> 1]      (the joined differences)
                //   r1 = T[sect1 + mn - 2][(mn - sect1) >> 1], mn = min(m, clen)  (intersection vs intersection +
                //        shorter difference; 100 when one side is contained in the other), used when sect1 > 0
                const uint32_t bm = (uint32_t)__builtin_popcount(M & fm[h]);
                uint32_t r0 = fz_tab_load(tab_off, c_tab[h] + both_odd + qs + bm + bv);
                const uint32_t m1 = m + fw[h] - 2u;                                      // scalar
                const uint32_t mn1 = m1 < c_len1[h] ? m1 : c_len1[h];                    // mn + fw - 2
                const uint32_t d1 = mn1 + bm - (2u * fw[h] - 2u);                        // mn - sect1
                uint32_t r1 = fz_tab_load(tab_off, ((mn1 - bm) << 6) + (d1 >> 1));
                asm volatile("" : "+v"(r0), "+v"(r1));               // plain 32-bit values (no 16-bit max + re-mask)
                sc = (bm != fw[h] && r1 > r0) ? r1 : r0;
                if (clen[h] == 0u) sc = 0u;                           // a candidate without tokens scores 0
            } else {
                // T[L][(L - 2*lcs) >> 1] with L = m + clen
                sc = fz_tab_load(tab_off, c_tab[h] + both_odd + qs + bv);
            }
            sc2[h] = sc;
        }
        if (EMIT) {
            // fused threshold + top-k: no score matrix; lanes below the limit's score are done here
            const uint32_t ls = (uint32_t)(lim >> FZ_IDX_BITS);
            if (__ballot((sc2[0] >= ls) | (sc2[1] >= ls)) != 0ull) {
                fz_emit(p.emit, out_row, lim, sc2[0], orig[0]);
                fz_emit(p.emit, out_row, lim, sc2[1], orig[1]);
            }
        } else {
            uint8_t *rowp = scores + (uint64_t)out_row * (uint64_t)p.row_stride;     // wave-uniform
            rowp[out_off[0]] = (uint8_t)sc2[0];
            rowp[out_off[1]] = (uint8_t)sc2[1];
        }
    }
#undef FZ_ADVANCE
}

template <int MODE, int STRIDE, bool EMIT>
__global__ __launch_bounds__(512)
__attribute__((amdgpu_waves_per_eu(8, 8)))                   // 64 registers: 8 waves per SIMD   // EMUL: /* occupancy hint: no meaning on the host */
void fz_score_stream_pk(FzStreamParams p, const FzRec *__restrict__ recs, uint8_t *__restrict__ scores)
{
    extern __shared__ __align__(16) uint32_t lds[];
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(lds);                   // [128][64] rounded ratios T[L][d >> 1]
    uint8_t *s_row = s_tab + FZ_STREAM_TAB_BYTES;                        // [256] symbol -> 4 * row
    uint8_t *s_peq = s_row + 256u;                                       // [64][STRIDE] u32: the block's ONE column set
    constexpr uint32_t PLANE = 64u * (uint32_t)STRIDE * 4u;
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_peq + PLANE);       // [0]: the block's current item

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t i = tid; i < FZ_STREAM_TAB_BYTES / 16u; i += blockDim.x)
        reinterpret_cast<uint4 *>(s_tab)[i] = reinterpret_cast<const uint4 *>(p.score_tab)[i];
    if (tid < 64u) reinterpret_cast<uint32_t *>(s_row)[tid] = reinterpret_cast<const uint32_t *>(p.row_of_sym)[tid];
    if (fz_lds_addr(s_tab) & (FZ_STREAM_TAB_BYTES - 1u)) __builtin_trap();    // fz_tab_load ORs indices onto this address

    uint8_t *col = s_peq + lane * ((uint32_t)STRIDE * 4u);               // this lane's column (shared by the block's waves)
    const uint32_t col_off = fz_lds_addr(col);
    const uint32_t npairs = (p.ntiles + 1u) / 2u;                        // an item = (query group, pair of tiles)
    FzDraw draw{blockIdx.x % FZ_STREAM_QUEUES, 0u, 0u};
    for (;;) {                                                           // same block protocol as fz_score_stream
        __syncthreads();
        if (wave == 0u) {
            const uint32_t it = fz_draw_item(p, draw, lane);
            if (lane == 0u) s_ctl[0] = it;
        }
        for (uint32_t r = wave; r < (uint32_t)STRIDE; r += nw) *reinterpret_cast<uint32_t *>(col + r * 4u) = 0u;
        __syncthreads();
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[0]);   // the same for every lane: keep what follows scalar
        if (item == 0xFFFFFFFFu) break;
        uint32_t qgroup, pair;
        fz_item_decode(item, npairs, p.slab_units, (p.nq + p.q_per_block - 1u) / p.q_per_block, qgroup, pair);

        // ---- candidate side: two tiles, once per item --------------------------------------------
        uint32_t cid[2][FZ_PK_MAX_TOK], Rm[2][FZ_PK_MAX_TOK], clen[2], out_off[2], psep = 0u, ntok = 0u;
        // entry 2*pair carries the split S above the tile index: field A = bits [0, S), B = bits [S + 1, 32)
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tile_list[2u * pair]);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tile_list[2u * pair + 1u]);
        const uint32_t S = e0 >> FZ_PK_SPLIT_SHIFT;
        const uint32_t sh[2] = {0u, S + 1u}, fw[2] = {S, 31u - S};
        const uint32_t fmask[2] = {(1u << S) - 1u, 0xFFFFFFFFu << (S + 1u)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t tile = (h == 0 ? e0 : e1) & ((1u << FZ_PK_SPLIT_SHIFT) - 1u);
            const FzTile td = p.tiles[tile];
            out_off[h] = tile * 64u + lane;
            if (td.ntok > ntok) ntok = td.ntok;
            uint32_t prev_end = 0u;
#pragma unroll
            for (int i = 0; i < (int)FZ_PK_MAX_TOK; ++i) {
                uint32_t end = prev_end;
                cid[h][i] = FZ_TOK_NONE;
                if ((uint32_t)i < td.ntok) {
                    if (MODE == FZ_MODE_TOKEN_SET) cid[h][i] = p.tokid[td.toff + (uint32_t)i * 64u + lane];
                    end = p.tokend[td.toff + (uint32_t)i * 64u + lane];
                }
                if (end > fw[h]) end = fw[h];                            // host guarantee; keeps the shifts defined
                Rm[h][i] = (((1u << end) - 1u) & ~((1u << prev_end) - 1u)) << sh[h];       // pattern positions of token i
                prev_end = end;
            }
            clen[h] = prev_end;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                if ((uint32_t)d < td.ndw) {
                    const uint32_t cw = p.sym[td.coff + (uint32_t)d * 64u + lane];
                    const bool mine = ((uint32_t)(d + h) % nw) == wave;      // this wave writes the match bits of dword d
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t s = (cw >> (8 * k)) & 0xFFu;
                        if ((uint32_t)(d * 4 + k) < fw[h]) {
                            const uint32_t bit = 1u << ((uint32_t)(d * 4 + k) + sh[h]);
                            if (MODE == FZ_MODE_TOKEN_SET && s == FZ_SYM_SPACE) {
                                psep |= bit;
                            } else if (mine) {
                                const uint32_t r4 = s_row[s];            // 0xFF: no query uses this symbol
                                if (r4 != 0xFFu)
                                    __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(col + r4), bit,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
            }
        }
        ntok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ntok);
        __syncthreads();                                 // the column is complete
        uint32_t q0, q1;
        fz_wave_queries(p, qgroup, wave, nw, q0, q1);
        uint32_t orig[2] = {0u, 0u};
        if (EMIT) { orig[0] = p.emit.orig_of_phys[out_off[0]]; orig[1] = p.emit.orig_of_phys[out_off[1]]; }
        if (MODE != FZ_MODE_TOKEN_SET || ntok <= 1u) fz_stream_queries_pk<MODE, STRIDE, 1, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, fmask, fw, 0u - (1u << S), orig, q0, q1);
        else if (ntok == 2u) fz_stream_queries_pk<MODE, STRIDE, 2, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, fmask, fw, 0u - (1u << S), orig, q0, q1);
        else if (ntok == 3u) fz_stream_queries_pk<MODE, STRIDE, 3, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, fmask, fw, 0u - (1u << S), orig, q0, q1);
        else fz_stream_queries_pk<MODE, STRIDE, (int)FZ_PK_MAX_TOK, EMIT>(p, recs, scores, s_tab, col_off, cid, Rm, psep, clen, out_off, fmask, fw, 0u - (1u << S), orig, q0, q1);
    }
}

// LDS of one block: score table, symbol -> row map, ONE column set (shared by the block's waves), control words
size_t fz_stream_lds_bytes(int W, int stride)
{
    return (size_t)FZ_STREAM_TAB_BYTES + 256u + (size_t)W * 64u * stride * 4u + 16u;
}

template <int W, int MODE, int STRIDE, bool EMIT>
static hipError_t launch_stream4(const FzStreamParams &p, hipStream_t st)
{
    const size_t lds = fz_stream_lds_bytes(W, STRIDE);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_score_stream<W, MODE, STRIDE, EMIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // persistent blocks (as many as a CU holds at this LDS size and 8 waves per SIMD) drain the (query group, tile) work queues
    const uint64_t blocks_needed = p.nitems;
    dim3 grid((uint32_t)(blocks_needed < p.grid_blocks ? blocks_needed : p.grid_blocks));
    hipLaunchKernelGGL((fz_score_stream<W, MODE, STRIDE, EMIT>), grid, dim3(p.nwaves * 64u), lds, st, p, p.recs, p.scores);
    return hipGetLastError();
}

// EMIT (fused threshold + top-k, FzEmit) is a template flag: the score-matrix build carries no trace of it
template <int W, int MODE, int STRIDE>
static hipError_t launch_stream3(const FzStreamParams &p, hipStream_t st)
{
    return p.emit.lim ? launch_stream4<W, MODE, STRIDE, true>(p, st) : launch_stream4<W, MODE, STRIDE, false>(p, st);
}

template <int W, int MODE>
static hipError_t launch_stream2(const FzStreamParams &p, hipStream_t st)
{
    if (p.stride == 37u) return launch_stream3<W, MODE, 37>(p, st);
    if (p.stride == 41u) return launch_stream3<W, MODE, 41>(p, st);
    if (p.stride == 65u) return launch_stream3<W, MODE, 65>(p, st);
    return hipErrorInvalidValue;
}

template <int MODE, int STRIDE, bool EMIT>
static hipError_t launch_stream_pk3(const FzStreamParams &p, hipStream_t st)
{
    const size_t lds = fz_stream_lds_bytes(1, STRIDE);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_score_stream_pk<MODE, STRIDE, EMIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t blocks_needed = p.nitems;
    dim3 grid((uint32_t)(blocks_needed < p.grid_blocks ? blocks_needed : p.grid_blocks));
    hipLaunchKernelGGL((fz_score_stream_pk<MODE, STRIDE, EMIT>), grid, dim3(p.nwaves * 64u), lds, st, p, p.recs, p.scores);
    return hipGetLastError();
}

template <int MODE, int STRIDE>
static hipError_t launch_stream_pk2(const FzStreamParams &p, hipStream_t st)
{
    return p.emit.lim ? launch_stream_pk3<MODE, STRIDE, true>(p, st) : launch_stream_pk3<MODE, STRIDE, false>(p, st);
}

template <int MODE>
static hipError_t launch_stream_pk(const FzStreamParams &p, hipStream_t st)
{
    if (p.stride == 37u) return launch_stream_pk2<MODE, 37>(p, st);
    if (p.stride == 41u) return launch_stream_pk2<MODE, 41>(p, st);
    if (p.stride == 65u) return launch_stream_pk2<MODE, 65>(p, st);
    return hipErrorInvalidValue;
}

// two tiles per wave: `p.ntiles` tiles in pairs (entry 2i carries the split of pair i), items = (query group, pair)
hipError_t fz_launch_stream_pk(int mode, const FzStreamParams &p, hipStream_t st)
{
    if (p.nq == 0u || p.ntiles == 0u) return hipSuccess;
    if (p.nwaves == 0u || p.nwaves > 8u || p.q_per_block == 0u || p.grid_blocks == 0u || !p.work_counter || p.chunk == 0u || p.chunk > 64u || p.slab_units == 0u) return hipErrorInvalidValue;
    if ((p.ntiles & 1u) || (uint64_t)(p.ntiles / 2u) * ((p.nq + p.q_per_block - 1u) / p.q_per_block) != p.nitems) return hipErrorInvalidValue;
    if (mode == FZ_MODE_TOKEN_SET) return launch_stream_pk<FZ_MODE_TOKEN_SET>(p, st);
    if (mode == FZ_MODE_RATIO) return launch_stream_pk<FZ_MODE_RATIO>(p, st);
    return hipErrorInvalidValue;
}

hipError_t fz_launch_stream(int W, int mode, const FzStreamParams &p, hipStream_t st)
{
    if (p.nq == 0u || p.ntiles == 0u) return hipSuccess;
    if (p.nwaves == 0u || p.nwaves > 8u || p.q_per_block == 0u || p.grid_blocks == 0u || !p.work_counter || p.chunk == 0u || p.chunk > 64u || p.slab_units == 0u) return hipErrorInvalidValue;
    if ((uint64_t)p.ntiles * ((p.nq + p.q_per_block - 1u) / p.q_per_block) != p.nitems) return hipErrorInvalidValue;
    if (W == 1 && mode == FZ_MODE_TOKEN_SET) return launch_stream2<1, FZ_MODE_TOKEN_SET>(p, st);
    if (W == 2 && mode == FZ_MODE_TOKEN_SET) return launch_stream2<2, FZ_MODE_TOKEN_SET>(p, st);
    if (W == 1 && mode == FZ_MODE_RATIO) return launch_stream2<1, FZ_MODE_RATIO>(p, st);
    if (W == 2 && mode == FZ_MODE_RATIO) return launch_stream2<2, FZ_MODE_RATIO>(p, st);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// top-k selection
// ---------------------------------------------------------------------------------------------
#define SEL16_THREADS 256u     // fz_select16_topk: one thread per histogram bin
#define SEL_TIECAP 2048u
#define SEL_WALK_STEPS 32u    // steps of the walk in original order before the radix select takes over
#define SEL_PICK_MAX 16u      // up to this many ties are picked by repeated minima instead of a sort
#define SEL_VEC 4u            // 16-byte vectors per thread per iteration of the row passes

__device__ __forceinline__ uint64_t fz_local_key(uint32_t score, uint32_t orig)
{
    return ((uint64_t)score << 32) | (uint64_t)(0xFFFFFFFFu - orig);
}

// block-wide suffix sums of the 128-bin histogram; afterwards s_cum[t] = #scores >= t (t < 128),
// s_cum[128] = 0, and s_tlow is raised to the largest t that still has >= k scores at or above it.
__device__ __forceinline__ void fz_sel_update(uint32_t *s_hist, uint32_t *s_cum, uint32_t *s_misc,
                                              uint32_t k)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    __syncthreads();
    uint32_t v = 0u;
    if (tid < 128u) {
        v = s_hist[tid];
        for (uint32_t off = 1u; off < 64u; off <<= 1) {
            uint32_t t = __shfl_down(v, off, 64);
            if (lane + off < 64u) v += t;
        }
        if (tid == 64u) s_misc[1] = v;                        // total of bins 64..127
    }
    __syncthreads();
    if (tid < 128u) {
        if (tid < 64u) v += s_misc[1];
        s_cum[tid] = v;
    }
    if (tid == 128u) s_cum[128] = 0u;
    __syncthreads();
    if (tid < 128u) {
        if (s_cum[tid] >= k && s_cum[tid + 1u] < k && tid > s_misc[0]) s_misc[0] = tid;
    }
    __syncthreads();
    // Once the cut value itself holds more than SEL_TIECAP + k candidates, further ones with that value
    // change nothing (enough ties for any `need`, and too many to sort): count only better scores from
    // here on.  Without this a row with 10^5 ties at the cut spends its whole pass in the histogram.
    if (tid == 0u) s_misc[8] = s_misc[0] + (s_hist[s_misc[0]] > SEL_TIECAP + k ? 1u : 0u);
    __syncthreads();
}

// Wave-aggregated histogram add: the lanes of a wave that hold the same bin share ONE LDS atomic.
// A row where tens of thousands of candidates tie on one score (a one-token query that is a subset
// of 10 % of the corpus) otherwise serialises on a single LDS address.  Every lane of the wave must
// call this (convergent); `valid` says whether the lane has something to add.
__device__ __forceinline__ void fz_hist_add(uint32_t *hist, uint32_t bin, bool valid)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t todo = __ballot(valid);
    while (todo != 0ull) {
        const uint32_t first = (uint32_t)__builtin_ctzll(todo);
        const uint32_t b0 = (uint32_t)__shfl((int)bin, (int)first, 64);
        const uint64_t same = __ballot(valid && bin == b0);
        if (lane == first) atomicAdd(&hist[b0], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

template <typename T, bool DESC>
__device__ __forceinline__ void fz_bitonic(T *a, uint32_t P)
{
    for (uint32_t k2 = 2u; k2 <= P; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0u; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    bool up = ((i & k2) == 0u);
                    T x = a[i], y = a[ixj];
                    bool sw = DESC ? (up ? (x < y) : (x > y)) : (up ? (x > y) : (x < y));
                    if (sw) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// THREADS: 1024 for the row of a whole shard; 256 for short rows (the sample of a fused call), where the
// block-wide steps (bound updates, tie handling, sorts) dominate and eight rows share a CU instead of two
// (8 waves per SIMD asked for: at more than 80 SGPRs a CU admits one 1024-thread block instead of two)
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS, 8) void fz_select_topk(FzSelectParams p)
{
    __shared__ uint32_t s_hist[128];
    __shared__ uint32_t s_cum[129];
    __shared__ uint64_t s_pick[(THREADS / 64u)];
    __shared__ uint32_t s_misc[16 + (THREADS / 64u)];   // 0: tlow  1: scratch  2: n_gt  3: n_tie  8: count filter  16..: wave counts
    __shared__ uint64_t s_keys[FZ_MAX_K];
    __shared__ uint32_t s_ties[SEL_TIECAP];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t row = blockIdx.x, k = p.k;
    // A sliced row (a few rows over a large shard) is dealt out tile by tile: block `slice` of the row's S blocks takes
    // the 64-position tiles t with t % S == slice, so every block sees every length class and original indices from
    // the whole range (runs of consecutive positions would not: inside a length class the physical order is the
    // original order, and the walk below would have to pass most of the list to reach a late run).  S is a power of two.
    const uint32_t S = p.n_slices > 1u ? p.n_slices : 1u, slice = p.n_slices > 1u ? blockIdx.y : 0u;
    const uint32_t tiles_all = (p.n + 63u) >> 6;
    // local position l (what the passes below count in) -> physical position: tile (l >> 6) * S + slice, byte l & 63
    const uint32_t n = S > 1u ? (tiles_all > slice ? ((tiles_all - slice + S - 1u) / S) << 6 : 0u) : p.n;
    const uint8_t *sc = p.scores + (uint64_t)row * p.row_stride;
    const uint32_t *orig_of_phys = p.orig_of_phys;
#define FZ_SEL_PHYS(l) (S > 1u ? ((((l) >> 6) * S + slice) << 6) + ((l) & 63u) : (l))
    const uint32_t min_score = p.min_score < 0 ? 0u : (uint32_t)p.min_score;

    if (tid < 128u) s_hist[tid] = 0u;
    if (tid == 0u) { s_misc[0] = min_score; s_misc[2] = 0u; s_misc[3] = 0u; s_misc[8] = min_score; }
    __syncthreads();
#define FZ_DBG(i) do { if (p.dbg && tid == 0u) p.dbg[(uint64_t)row * 8u + (i)] = (uint64_t)wall_clock64(); } while (0)
    FZ_DBG(0);

    // ---- pass A: histogram of the scores that can still matter ------------------------------
    // Each thread takes SEL_VEC 16-byte vectors per iteration (all loads issued before the first use)
    // and tests a dword with one SWAR expression before it looks at its bytes: with t in 1..128,
    // ((x + 0x01010101 * (128 - t)) | x) & 0x80808080 is non-zero iff some byte of x is >= t (a carry
    // out of a byte >= 128 can only add false positives; the byte loop below is exact).
    // Pass B reads again only what can matter: `seen` has one bit per vector of this thread, set when the
    // vector held a byte >= the bound of its time.  That bound only rises and never exceeds pass B's own
    // filter (tlow <= sstar, or sstar + 1 once the ties at sstar are past sorting, and then pass B filters
    // at sstar + 1 as well), so an unmarked vector holds nothing pass B would take.  After the first
    // update of the bound (two iterations) almost no vector is marked.  Rows of more than 64 vectors per
    // thread (> 2^20 candidates) are read in full.
    // Every thread takes nv 16-byte vectors, vector i of thread t at local position (i * THREADS + t) * 16.  They go in
    // groups: the FIRST group is one vector per thread, so that the bound rises after THREADS * 16 positions already
    // (a k-th best of a prefix is a valid lower bound of the row's; until the first update everything >= min_score takes
    // the slow path below), then SEL_VEC vectors at a time.
    const uint32_t nv = (n + THREADS * 16u - 1u) / (THREADS * 16u);
    const bool track = nv <= 64u;
    uint64_t seen = 0ull;
    uint32_t upd = 0u;
    for (uint32_t v0 = 0u; v0 < nv; ) {
        const uint32_t gcnt = v0 == 0u ? 1u : SEL_VEC;
        const uint32_t tlow = s_misc[8];                     // count scores >= this (<= 101)
        const uint32_t add = 0x01010101u * (128u - (tlow ? tlow : 1u));
        uint4 v[SEL_VEC];
#pragma unroll
        for (uint32_t j = 0; j < SEL_VEC; ++j) {
            const uint32_t base = ((v0 + j) * THREADS + tid) * 16u;
            v[j] = (j < gcnt && base < n) ? *reinterpret_cast<const uint4 *>(sc + FZ_SEL_PHYS(base)) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (uint32_t j = 0; j < SEL_VEC; ++j) {
            if (j >= gcnt) break;                                            // (uniform)
            const uint32_t base = ((v0 + j) * THREADS + tid) * 16u;
            const uint32_t pbase = FZ_SEL_PHYS(base);
            const uint32_t wv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            uint32_t any = 0u;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // one flag bit (7, 15, 23, 31) per candidate byte; the lane then visits only those bytes
                uint32_t m = tlow != 0u ? (((wv[a] + add) | wv[a]) & 0x80808080u) : 0x80808080u;
                any |= m;
                while (__ballot(m != 0u) != 0ull) {                      // wave-uniform trip count
                    const bool on = m != 0u;
                    const uint32_t sh = on ? (uint32_t)__builtin_ctz(m) - 7u : 0u;
                    m &= m - 1u;
                    uint32_t s = (wv[a] >> sh) & 0xFFu;                  // a byte > 100 is not a score: skipped
                    if (p.flat && s >= min_score) s = s <= 100u ? min_score : s;   // index order only
                    fz_hist_add(s_hist, s, on && s >= tlow && s <= 100u && base < n && pbase + (uint32_t)(4 * a) + (sh >> 3) < p.n);
                }
            }
            if (track && any != 0u) seen |= 1ull << (v0 + j);
        }
        v0 += gcnt;
        if (upd < 2u || (upd & 1u) == 0u) fz_sel_update(s_hist, s_cum, s_misc, k);   // after the first two groups, then every other one
        ++upd;
    }
    fz_sel_update(s_hist, s_cum, s_misc, k);

    FZ_DBG(1);
    const uint32_t sstar = s_misc[0];                    // k-th best score (or min_score)
    const uint32_t n_gt = s_cum[sstar + 1u];             // strictly better: all of them are results
    const uint32_t c_eq = s_hist[sstar];                 // ties at the cut
    uint32_t need = k - n_gt;                            // n_gt < k by construction
    if (need > c_eq) need = c_eq;
    const bool few_ties = c_eq <= SEL_TIECAP;

    // ---- pass B: collect ----------------------------------------------------------------------
    {
        // ties are collected here only when they are few enough to sort; otherwise look at better scores only
        const uint32_t bfil = (few_ties && need > 0u) ? sstar : sstar + 1u;
        const uint32_t add = 0x01010101u * (128u - (bfil ? bfil : 1u));
        for (uint32_t v0 = 0u; v0 < nv; v0 += SEL_VEC) {
            uint4 v[SEL_VEC];
#pragma unroll
            for (uint32_t j = 0; j < SEL_VEC; ++j) {
                const uint32_t base = ((v0 + j) * THREADS + tid) * 16u;
                const bool want = !track || ((seen >> (v0 + j)) & 1ull) != 0ull;
                v[j] = (base < n && want) ? *reinterpret_cast<const uint4 *>(sc + FZ_SEL_PHYS(base)) : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (uint32_t j = 0; j < SEL_VEC; ++j) {
                const uint32_t base = FZ_SEL_PHYS(((v0 + j) * THREADS + tid) * 16u);     // physical from here on
                const uint32_t wv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    uint32_t m = bfil != 0u ? (((wv[a] + add) | wv[a]) & 0x80808080u) : 0x80808080u;
                    while (m != 0u) {
                        const uint32_t sh = (uint32_t)__builtin_ctz(m) - 7u;
                        m &= m - 1u;
                        uint32_t s = (wv[a] >> sh) & 0xFFu;
                        if (p.flat && s >= min_score) s = s <= 100u ? min_score : s;
                        const uint32_t ph = base + (uint32_t)(4 * a) + (sh >> 3);
                        if (ph < p.n && s >= sstar && s <= 100u) {
                            if (s > sstar) {
                                uint32_t pos = atomicAdd(&s_misc[2], 1u);
                                s_keys[pos] = fz_local_key(s, orig_of_phys[ph]);
                            } else if (few_ties && need > 0u) {
                                uint32_t pos = atomicAdd(&s_misc[3], 1u);
                                s_ties[pos] = orig_of_phys[ph];
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    FZ_DBG(2);
    // ---- ties at the cut: the `need` smallest original indices ------------------------------
    if (need > 0u) {
        if (few_ties) {
            if (need < c_eq && need <= SEL_PICK_MAX) {
                // a few out of many (top-10 under a thousand ties): `need` block-wide minima instead of a
                // sort of all ties.  Round r moves the smallest of s_ties[r, c_eq) to position r; original
                // indices are distinct, so (index << 32 | position) has one minimum.
                for (uint32_t r = 0; r < need; ++r) {
                    uint64_t best = ~0ull;
                    for (uint32_t i = r + tid; i < c_eq; i += blockDim.x) {
                        const uint64_t key = ((uint64_t)s_ties[i] << 32) | i;
                        if (key < best) best = key;
                    }
                    for (uint32_t off = 32u; off > 0u; off >>= 1) {
                        const uint32_t hi = __shfl_down((uint32_t)(best >> 32), off, 64), lo = __shfl_down((uint32_t)best, off, 64);
                        const uint64_t other = ((uint64_t)hi << 32) | lo;
                        if (lane + off < 64u && other < best) best = other;
                    }
                    if (lane == 0u) s_pick[wave] = best;
                    __syncthreads();
                    if (tid == 0u) {
                        uint64_t m = s_pick[0];
                        for (uint32_t w = 1; w < (THREADS / 64u); ++w) if (s_pick[w] < m) m = s_pick[w];
                        const uint32_t pos = (uint32_t)m;
                        s_ties[pos] = s_ties[r];
                        s_ties[r] = (uint32_t)(m >> 32);
                    }
                    __syncthreads();
                }
            } else if (need < c_eq) {
                uint32_t P = 2u;
                while (P < c_eq) P <<= 1;
                for (uint32_t i = c_eq + tid; i < P; i += blockDim.x) s_ties[i] = 0xFFFFFFFFu;
                __syncthreads();
                fz_bitonic<uint32_t, false>(s_ties, P);
            }
        } else {
            // many ties: walk candidates in ORIGINAL order and keep the first `need` tied ones.  Where ties are dense from
            // the start of the list (a one-token query that a third of the names contain) that takes one or two steps.
            uint32_t found = 0u, o0 = 0u;
            // (a sample row walks the sample's own list of original indices, ascending, not all of the shard's)
            const uint32_t n_walk = p.walk ? n : p.n_orig;
            const uint32_t budget = SEL_WALK_STEPS * THREADS;
            for (; o0 < n_walk && found < need && o0 < budget; o0 += THREADS) {
                const uint32_t wi = o0 + tid;
                const uint32_t o = wi < n_walk ? (p.walk ? p.walk[wi] : wi) : 0xFFFFFFFFu;
                bool pred = false;
                if (o < p.n_orig) {
                    const uint32_t ph = p.phys_of_orig[o];
                    if (ph < p.n && ((ph >> 6) & (S - 1u)) == slice) {           // a row may hold only the front of the physical order; a block only its tiles
                        const uint32_t s = sc[ph];
                        pred = p.flat ? (s >= sstar && s <= 100u) : (s == sstar);    // flat: sstar == min_score
                    }
                }
                const uint64_t bal = __ballot(pred);
                if (lane == 0u) s_misc[16 + wave] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t before = found;
                for (uint32_t w = 0; w < wave; ++w) before += s_misc[16 + w];
                const uint32_t mine = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (pred && mine < need) s_ties[mine] = o;
                for (uint32_t w = 0; w < (THREADS / 64u); ++w) found += s_misc[16 + w];
                __syncthreads();
            }
            if (found < need && o0 < n_walk) {
                // The ties lie far into the list (the holders of a token that only one region of the list uses: the walk
                // would pass millions of candidates first).  Radix select instead: the need-th smallest original index X
                // among the ties, 11 bits per pass over this block's scores with a histogram in s_ties, then the ties with
                // index <= X -- exactly `need` of them, indices being distinct -- are collected.  Independent of where the
                // ties sit; a pass costs one read of the block's scores plus one index load and one LDS add per tie.
                auto scan_ties = [&](auto &&fn) {
                    const uint32_t addq = 0x01010101u * (128u - (sstar ? sstar : 1u));
                    for (uint32_t v0 = 0u; v0 < nv; ++v0) {
                        const uint32_t base = (v0 * THREADS + tid) * 16u;
                        if (base >= n) break;
                        const uint32_t pb = FZ_SEL_PHYS(base);
                        const uint4 vq = *reinterpret_cast<const uint4 *>(sc + pb);
                        const uint32_t wv[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            uint32_t m = sstar != 0u ? (((wv[a] + addq) | wv[a]) & 0x80808080u) : 0x80808080u;
                            while (m != 0u) {
                                const uint32_t sh = (uint32_t)__builtin_ctz(m) - 7u;
                                m &= m - 1u;
                                const uint32_t sv = (wv[a] >> sh) & 0xFFu;
                                const uint32_t ph = pb + (uint32_t)(4 * a) + (sh >> 3);
                                if (ph < p.n && (p.flat ? (sv >= sstar && sv <= 100u) : (sv == sstar))) fn(orig_of_phys[ph]);
                            }
                        }
                    }
                };
                const uint32_t total_bits = p.n_orig > 1u ? 32u - (uint32_t)__builtin_clz(p.n_orig - 1u) : 1u;
                uint32_t hi = total_bits, prefix = 0u, remaining = need;
                while (hi > 0u) {
                    const uint32_t lo_b = hi > 11u ? hi - 11u : 0u, nb = 1u << (hi - lo_b);
                    for (uint32_t i = tid; i < nb; i += THREADS) s_ties[i] = 0u;
                    __syncthreads();
                    const bool first = hi == total_bits;
                    scan_ties([&](uint32_t o) {
                        if (first || (o >> hi) == prefix) atomicAdd(&s_ties[(o >> lo_b) & (nb - 1u)], 1u);
                    });
                    __syncthreads();
                    if (wave == 0u) {
                        // the bin where the running count reaches `remaining`: each lane sums a run of bins, a wave
                        // scan finds the run, its lane walks it
                        const uint32_t per = nb >= 64u ? nb / 64u : 1u;
                        uint32_t sum = 0u;
                        for (uint32_t j = 0u; j < per; ++j) { const uint32_t b_ = lane * per + j; if (b_ < nb) sum += s_ties[b_]; }
                        uint32_t cum = sum;
                        for (uint32_t off = 1u; off < 64u; off <<= 1) {
                            const uint32_t t_ = (uint32_t)__shfl((int)cum, (int)(lane >= off ? lane - off : lane), 64);
                            if (lane >= off) cum += t_;
                        }
                        const uint64_t reach = __ballot(cum >= remaining);      // non-empty: the ties number at least `need`
                        const uint32_t who = reach != 0ull ? (uint32_t)__builtin_ctzll(reach) : 63u;
                        if (lane == who) {
                            uint32_t below = cum - sum, b_ = lane * per;
                            for (uint32_t j = 0u; j + 1u < per && below + s_ties[b_] < remaining; ++j) { below += s_ties[b_]; ++b_; }
                            s_misc[4] = b_; s_misc[5] = below;
                        }
                    }
                    __syncthreads();
                    prefix = (prefix << (hi - lo_b)) | s_misc[4];
                    remaining -= s_misc[5];
                    hi = lo_b;
                    __syncthreads();
                }
                if (tid == 0u) s_misc[3] = 0u;
                __syncthreads();
                const uint32_t X = prefix;
                scan_ties([&](uint32_t o) {
                    if (o <= X) { const uint32_t pos = atomicAdd(&s_misc[3], 1u); if (pos < need) s_ties[pos] = o; }
                });
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < need; i += blockDim.x) s_keys[n_gt + i] = fz_local_key(sstar, s_ties[i]);
    }
    FZ_DBG(3);
    const uint32_t total = n_gt + need;
    uint32_t P = 2u;
    while (P < total) P <<= 1;
    __syncthreads();
    for (uint32_t i = total + tid; i < P; i += blockDim.x) s_keys[i] = 0ull;
    __syncthreads();
    fz_bitonic<uint64_t, true>(s_keys, P);
    FZ_DBG(4);
    if (p.dbg && tid == 0u) { p.dbg[(uint64_t)row * 8u + 5u] = c_eq; p.dbg[(uint64_t)row * 8u + 6u] = need; p.dbg[(uint64_t)row * 8u + 7u] = sstar; }

    if (p.n_slices > 1u) {
        // ---- sliced row: leave this run's keys, and merge the row if this block is the last one to finish -------
        const uint32_t S = p.n_slices;
        uint64_t *mine = p.slice_keys + ((uint64_t)row * S + blockIdx.y) * k;
        for (uint32_t i = tid; i < k; i += blockDim.x) {
            uint64_t out = 0ull;
            if (i < total) {
                const uint64_t key = s_keys[i];
                const uint64_t gidx = p.base_index + (uint64_t)(0xFFFFFFFFu - (uint32_t)key);
                out = ((key >> 32) << FZ_IDX_BITS) | (FZ_IDX_MASK - gidx);
            }
            __hip_atomic_store(&mine[i], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        __syncthreads();
        if (tid == 0u) s_misc[1] = atomicAdd(&p.slice_done[row], 1u);
        __syncthreads();
        if (s_misc[1] != S - 1u) return;                      // (uniform for the block)
        __threadfence();
        const uint32_t all = S * k;                           // <= FZ_MAX_K (host)
        uint32_t P2 = 2u;
        while (P2 < all) P2 <<= 1;
        __syncthreads();
        for (uint32_t i = tid; i < P2; i += blockDim.x)       // the other blocks' keys: loads that bypass this CU's L1
            s_keys[i] = i < all ? __hip_atomic_load(&p.slice_keys[(uint64_t)row * all + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        __syncthreads();
        fz_bitonic<uint64_t, true>(s_keys, P2);
        uint32_t cnt = 0u;
        for (uint32_t i = tid; i < k; i += blockDim.x) {
            const uint64_t v = s_keys[i];
            p.out_keys[(uint64_t)row * k + i] = v;
            cnt += v != 0ull ? 1u : 0u;
        }
        if (tid == 0u) s_misc[2] = 0u;
        __syncthreads();
        if (cnt) atomicAdd(&s_misc[2], cnt);
        __syncthreads();
        if (tid == 0u) {
            if (p.out_cnt != nullptr) p.out_cnt[row] = s_misc[2];
            p.slice_done[row] = 0u;                           // ready for the next launch
        }
        return;
    }
    for (uint32_t i = tid; i < k; i += blockDim.x) {
        uint64_t out = 0ull;
        if (i < total) {
            const uint64_t key = s_keys[i];
            const uint64_t score = key >> 32;
            const uint64_t gidx = p.base_index + (uint64_t)(0xFFFFFFFFu - (uint32_t)key);
            out = (score << FZ_IDX_BITS) | (FZ_IDX_MASK - gidx);
        }
        p.out_keys[(uint64_t)row * k + i] = out;
    }
    if (p.out_cnt != nullptr && tid == 0u) p.out_cnt[row] = total;
    if (p.out_lim != nullptr && tid == 0u) {
        // what a candidate outside this row must reach to enter the result: beat the k-th key, or, while the
        // row holds fewer than k, pass the threshold
        uint64_t lim = (uint64_t)min_score << FZ_IDX_BITS;
        if (total == k && k != 0u) {
            const uint64_t key = s_keys[k - 1u];
            const uint64_t gidx = p.base_index + (uint64_t)(0xFFFFFFFFu - (uint32_t)key);
            lim = (((key >> 32) << FZ_IDX_BITS) | (FZ_IDX_MASK - gidx)) + 1ull;
        }
        p.out_lim[row] = lim;
    }
}

// ---------------------------------------------------------------------------------------------
// fused top-k, last step: merge the sample's top-k with the keys the other tiles appended
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fz_merge_lists(FzMergeParams p)
{
    extern __shared__ __align__(16) uint64_t s_mk[];
    const uint32_t tid = threadIdx.x, row = blockIdx.x, k = p.k;
    const uint32_t cnt = p.cnt[row];
    const uint32_t nl = cnt < p.cap ? cnt : p.cap;
    if (cnt > p.cap && tid == 0u) {                                   // the host redoes the row (or the chunk) through the score matrix
        atomicAdd(p.overflow, 1u);
        if (p.ovf_rows != nullptr) {
            const uint32_t at = atomicAdd(&p.ovf_rows[0], 1u);
            if (at < p.ovf_rows_cap) p.ovf_rows[1u + at] = p.row_base + row;
        }
    }
    const uint32_t total = k + nl;
    uint32_t P = 2u;
    while (P < total) P <<= 1;
    for (uint32_t i = tid; i < P; i += blockDim.x) {
        uint64_t v = 0ull;
        if (i < k) v = p.sample_keys[(uint64_t)row * k + i];
        else if (i < total) v = p.list[(uint64_t)row * p.cap + (i - k)];
        s_mk[i] = v;
    }
    __syncthreads();
    fz_bitonic<uint64_t, true>(s_mk, P);
    uint32_t mine = 0u;
    for (uint32_t i = tid; i < k; i += blockDim.x) {
        const uint64_t v = s_mk[i];
        p.out_keys[(uint64_t)row * k + i] = v;
        mine += v != 0ull ? 1u : 0u;
    }
    if (p.out_cnt != nullptr) {
        __shared__ uint32_t s_n;
        if (tid == 0u) s_n = 0u;
        __syncthreads();
        if (mine) atomicAdd(&s_n, mine);
        __syncthreads();
        if (tid == 0u) p.out_cnt[row] = s_n;
    }
}

// ---------------------------------------------------------------------------------------------
// sharded path: the gathered key lists [n_lists][nq][k] of a query -> its global top-k (descending key)
// ---------------------------------------------------------------------------------------------
// list l of rank-major gathered data starts at all + l * list_stride (nq*k keys, then words the merge ignores).
// status_out (optional): the largest of the words all[l * list_stride + nq*k], l = 0 .. n_lists-1 -- the status
// each rank sends behind its keys (fz_query_topk_sharded)
__global__ __launch_bounds__(256) void fz_merge_gathered(const uint64_t *__restrict__ all, uint64_t list_stride, uint32_t n_lists, uint32_t nq,
                                                         uint32_t k, uint64_t *__restrict__ out, uint64_t *__restrict__ status_out)
{
    extern __shared__ __align__(16) uint64_t s_mg[];
    const uint32_t tid = threadIdx.x, row = blockIdx.x;
    const uint32_t total = n_lists * k;
    uint32_t P = 2u;
    while (P < total) P <<= 1;
    for (uint32_t i = tid; i < P; i += blockDim.x) {
        uint64_t v = 0ull;                                  // padding sorts last: a real key is never 0
        if (i < total) v = all[(uint64_t)(i / k) * list_stride + (uint64_t)row * k + (i % k)];
        s_mg[i] = v;
    }
    __syncthreads();
    fz_bitonic<uint64_t, true>(s_mg, P);
    for (uint32_t i = tid; i < k; i += blockDim.x) out[(uint64_t)row * k + i] = s_mg[i];
    if (status_out != nullptr && row == 0u && tid == 0u) {
        uint64_t worst = 0ull;
        for (uint32_t l = 0; l < n_lists; ++l) {
            const uint64_t v = all[(uint64_t)l * list_stride + (uint64_t)nq * k];
            if (v > worst) worst = v;
        }
        *status_out = worst;
    }
}

hipError_t fz_launch_merge_gathered_strided(const uint64_t *all, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                                            uint64_t *out, uint64_t *status_out, hipStream_t st)
{
    if (nq == 0u || k == 0u || n_lists == 0u) return hipSuccess;
    if ((uint64_t)n_lists * k > FZ_COMM_MAX_MERGE) return hipErrorInvalidValue;
    uint32_t P = 2u;
    while (P < n_lists * k) P <<= 1;
    const size_t lds = (size_t)P * 8u;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_merge_gathered), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fz_merge_gathered, dim3(nq), dim3(256), lds, st, all, list_stride, n_lists, nq, k, out, status_out);
    return hipGetLastError();
}

hipError_t fz_launch_merge_gathered(const uint64_t *all, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t *out, hipStream_t st)
{
    return fz_launch_merge_gathered_strided(all, (uint64_t)nq * k, n_lists, nq, k, out, nullptr, st);
}

// status word of an asynchronous fused call: 2 when any chunk's overflow counter is non-zero (the word was
// cleared on the stream before the call's kernels)
__global__ __launch_bounds__(64) void fz_fold_status(const uint32_t *__restrict__ ovf, uint32_t n, uint64_t *__restrict__ status)
{
    bool any = false;
    for (uint32_t i = threadIdx.x; i < n; i += 64u) any |= ovf[i] != 0u;
    if (__ballot(any) != 0ull && threadIdx.x == 0u) *status = 2ull;
}

hipError_t fz_launch_fold_status(const uint32_t *ovf, uint32_t n, uint64_t *status, hipStream_t st)
{
    if (status == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fz_fold_status, dim3(1), dim3(64), 0, st, ovf, n, status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// score / distance rows from physical to ORIGINAL candidate order (fz_score_all, fz_distance_all)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fz_unpermute_rows(const T *__restrict__ in, uint64_t in_stride, T *__restrict__ out,
                                                         uint64_t out_stride, const uint32_t *__restrict__ phys_of_orig, uint32_t n)
{
    const T *src = in + (uint64_t)blockIdx.y * in_stride;
    T *dst = out + (uint64_t)blockIdx.y * out_stride;
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) dst[o] = src[phys_of_orig[o]];
}

hipError_t fz_launch_unpermute(const void *in, uint64_t in_stride, void *out, uint64_t out_stride, const uint32_t *phys_of_orig,
                               uint32_t n, uint32_t rows, uint32_t elem_bytes, hipStream_t st)
{
    if (rows == 0u || n == 0u) return hipSuccess;
    const uint32_t bx = (n + 255u) / 256u;
    for (uint32_t r0 = 0u; r0 < rows; r0 += FZ_GRID_Y_MAX) {
        const uint32_t cnt = rows - r0 < FZ_GRID_Y_MAX ? rows - r0 : FZ_GRID_Y_MAX;
        const dim3 grid(bx < 4096u ? bx : 4096u, cnt);
        if (elem_bytes == 1u)
            hipLaunchKernelGGL(fz_unpermute_rows<uint8_t>, grid, dim3(256), 0, st, static_cast<const uint8_t *>(in) + (uint64_t)r0 * in_stride, in_stride,
                               static_cast<uint8_t *>(out) + (uint64_t)r0 * out_stride, out_stride, phys_of_orig, n);
        else
            hipLaunchKernelGGL(fz_unpermute_rows<uint16_t>, grid, dim3(256), 0, st, static_cast<const uint16_t *>(in) + (uint64_t)r0 * in_stride, in_stride,
                               static_cast<uint16_t *>(out) + (uint64_t)r0 * out_stride, out_stride, phys_of_orig, n);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t fz_launch_merge(const FzMergeParams &p, uint32_t rows, hipStream_t st)
{
    if (rows == 0u || p.k == 0u) return hipSuccess;
    uint32_t P = 2u;
    while (P < p.k + p.cap) P <<= 1;
    const size_t lds = (size_t)P * 8u;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_merge_lists), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fz_merge_lists, dim3(rows), dim3(256), lds, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// two scorers -> one rank per candidate -> top-k  (weighted FuzzySearch of reference app.py:995-1032)
// ---------------------------------------------------------------------------------------------
// The reference's weighted score is an fp64 expression of two integer scores; the host evaluates it
// for all 101 x 101 pairs with the reference's own arithmetic and hands over the RANK of each value
// among the distinct results, so ordering and threshold on the device are integer-exact.
__global__ __launch_bounds__(256) void fz_combine_rank(FzCombineParams p)
{
    __shared__ uint16_t s_tab[FZ_RANK_DIM * FZ_RANK_DIM];
    for (uint32_t i = threadIdx.x; i < FZ_RANK_DIM * FZ_RANK_DIM; i += blockDim.x) s_tab[i] = p.rank_tab[i];
    __syncthreads();
    const uint32_t row = blockIdx.y;
    const uint8_t *sa = p.scores_a + (uint64_t)row * p.stride_a;
    const uint8_t *sb = p.scores_b + (uint64_t)row * p.stride_b;
    uint16_t *out = p.ranks + (uint64_t)row * p.stride_r;
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < p.n; o += gridDim.x * blockDim.x) {
        uint32_t a = sa[p.phys_a[o]], b = sb[p.phys_b[o]];
        if (a > 100u) a = 100u;
        if (b > 100u) b = 100u;
        out[o] = s_tab[a * FZ_RANK_DIM + b];
    }
}

// suffix sums of a 256-bin histogram: s_cum[t] = #values in bins >= t, s_cum[256] = 0
__device__ __forceinline__ void fz_sel16_suffix(const uint32_t *s_hist, uint32_t *s_cum, uint32_t *s_part)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __syncthreads();
    uint32_t v = s_hist[tid];
    for (uint32_t off = 1u; off < 64u; off <<= 1) {
        const uint32_t t = __shfl_down(v, off, 64);
        if (lane + off < 64u) v += t;
    }
    if (lane == 0u) s_part[wave] = v;
    __syncthreads();
    for (uint32_t w = wave + 1u; w < 4u; ++w) v += s_part[w];
    s_cum[tid] = v;
    if (tid == 0u) s_cum[256] = 0u;
    __syncthreads();
}

__global__ __launch_bounds__(256) void fz_select16_topk(FzSelect16Params p)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_cum[257];
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_misc[8];          // 0: hi bin  1: r*  2: n_gt  3: n_tie  4..7: wave counts
    __shared__ uint64_t s_keys[FZ_MAX_K];
    __shared__ uint32_t s_ties[SEL_TIECAP];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = p.n, k = p.k, minr = p.min_rank;
    const uint16_t *rk = p.ranks + (uint64_t)blockIdx.x * p.stride_r;

    // ---- pass 1: high bytes ------------------------------------------------------------------
    s_hist[tid] = 0u;
    if (tid == 0u) { s_misc[0] = minr >> 8; s_misc[1] = minr; s_misc[2] = 0u; s_misc[3] = 0u; }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {        // uniform trip count: fz_hist_add is convergent
        const uint32_t i = i0 + tid;
        const uint32_t r = i < n ? rk[i] : 0u;
        fz_hist_add(s_hist, r >> 8, i < n && r >= minr);
    }
    fz_sel16_suffix(s_hist, s_cum, s_part);
    if (s_cum[tid] >= k && s_cum[tid + 1u] < k && tid > s_misc[0]) s_misc[0] = tid;   // at most one bin
    __syncthreads();
    const uint32_t B = s_misc[0];
    const uint32_t above = s_cum[B + 1u];                    // values in higher bins: all are results
    __syncthreads();

    // ---- pass 2: low bytes inside bin B ------------------------------------------------------
    s_hist[tid] = 0u;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t i = i0 + tid;
        const uint32_t r = i < n ? rk[i] : 0u;
        fz_hist_add(s_hist, r & 255u, i < n && r >= minr && (r >> 8) == B);
    }
    fz_sel16_suffix(s_hist, s_cum, s_part);
    {
        const uint32_t kk = k - above;                       // above < k by construction (or B is the floor)
        const uint32_t r = (B << 8) | tid;
        if (above < k && s_cum[tid] >= kk && s_cum[tid + 1u] < kk && r > s_misc[1]) s_misc[1] = r;
    }
    __syncthreads();
    const uint32_t rstar = s_misc[1];                        // k-th best rank (or min_rank)
    const uint32_t lo = rstar & 255u;
    const uint32_t n_gt = above + ((rstar >> 8) == B ? s_cum[lo + 1u] : 0u);
    const uint32_t c_eq = (rstar >> 8) == B ? s_hist[lo] : 0u;
    uint32_t need = n_gt < k ? k - n_gt : 0u;
    if (need > c_eq) need = c_eq;
    const bool few_ties = c_eq <= SEL_TIECAP;
    __syncthreads();

    // ---- pass 3: collect ---------------------------------------------------------------------
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint32_t r = rk[i];
        if (r > rstar) {
            const uint32_t pos = atomicAdd(&s_misc[2], 1u);
            if (pos < FZ_MAX_K) s_keys[pos] = ((uint64_t)r << 32) | (uint64_t)(0xFFFFFFFFu - i);
        } else if (r == rstar && r >= minr && few_ties && need > 0u) {
            const uint32_t pos = atomicAdd(&s_misc[3], 1u);
            if (pos < SEL_TIECAP) s_ties[pos] = i;
        }
    }
    __syncthreads();
    if (need > 0u) {
        if (few_ties) {
            if (need < c_eq) {
                uint32_t P = 2u;
                while (P < c_eq) P <<= 1;
                for (uint32_t i = c_eq + tid; i < P; i += blockDim.x) s_ties[i] = 0xFFFFFFFFu;
                __syncthreads();
                fz_bitonic<uint32_t, false>(s_ties, P);
            }
        } else {
            // many ties: candidates are in ORIGINAL order here, so the first `need` tied ones win
            uint32_t found = 0u;
            for (uint32_t o0 = 0u; o0 < n && found < need; o0 += SEL16_THREADS) {
                const uint32_t o = o0 + tid;
                const bool pred = o < n && rk[o] == (uint16_t)rstar;
                const uint64_t bal = __ballot(pred);
                if (lane == 0u) s_misc[4 + wave] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t before = found;
                for (uint32_t w = 0; w < wave; ++w) before += s_misc[4 + w];
                const uint32_t mine = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (pred && mine < need) s_ties[mine] = o;
                found += s_misc[4] + s_misc[5] + s_misc[6] + s_misc[7];
                __syncthreads();
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < need; i += blockDim.x)
            s_keys[n_gt + i] = ((uint64_t)rstar << 32) | (uint64_t)(0xFFFFFFFFu - s_ties[i]);
    }
    const uint32_t total = n_gt + need;
    uint32_t P = 2u;
    while (P < total) P <<= 1;
    __syncthreads();
    for (uint32_t i = total + tid; i < P; i += blockDim.x) s_keys[i] = 0ull;
    __syncthreads();
    fz_bitonic<uint64_t, true>(s_keys, P);
    for (uint32_t i = tid; i < k; i += blockDim.x) {
        uint64_t out = 0ull;
        if (i < total) {
            const uint64_t key = s_keys[i];
            const uint64_t gidx = p.base_index + (uint64_t)(0xFFFFFFFFu - (uint32_t)key);
            out = ((key >> 32) << FZ_IDX_BITS) | (FZ_IDX_MASK - gidx);
        }
        p.out_keys[(uint64_t)blockIdx.x * k + i] = out;
    }
    if (p.out_cnt != nullptr && tid == 0u) p.out_cnt[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// static scratch of fz_query_topk_noalloc
// ---------------------------------------------------------------------------------------------
// One array per GPU, part of the code object: no call ever allocates it.  In host code the NAME of a
// __device__ array is not a device address: the address on a given GPU comes from hipGetSymbolAddress
// with that GPU current, once per device, and only that pointer is handed to kernels, memsets and copies.
static __device__ __align__(256) unsigned char fz_scratch[FZ_NOALLOC_SCRATCH_BYTES];

#define FZ_MAX_DEVICES 64
hipError_t fz_scratch_acquire(int device, void **ptr, size_t *bytes)
{
    static std::mutex mu;
    static void *ptrs[FZ_MAX_DEVICES] = {};
    if (device < 0 || device >= FZ_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(mu);
    if (ptrs[device] == nullptr) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur != device && (e = hipSetDevice(device)) != hipSuccess) return e;
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(fz_scratch));
        if (cur != device) (void)hipSetDevice(cur);
        if (e != hipSuccess) return e;
        if (p == nullptr) return hipErrorInvalidValue;
        ptrs[device] = p;
    }
    *ptr = ptrs[device];
    *bytes = FZ_NOALLOC_SCRATCH_BYTES;
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from fz_host.cpp)
// ---------------------------------------------------------------------------------------------
static const int kGeneralWidths[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};

int fz_pick_width(uint32_t m)
{
    uint32_t need = (m + 31u) / 32u;
    if (need == 0u) need = 1u;
    for (int w : kGeneralWidths)
        if ((uint32_t)w >= need) return w;
    return -1;
}

size_t fz_score_lds_bytes(int W, const FzScoreParams &p)
{
    if (p.ro) {                                                   // fz_ro_matches: Peq, map, then per wave R | S | stack
        size_t head = (size_t)p.peq_rows * fz_peq_stride(W) + (p.wide ? ((size_t)p.alphabet + 1ull) / 2ull : 0ull);
        head = (head + 1ull) & ~1ull;
        return 4ull * (head + (size_t)p.nwaves * 64ull * (2ull * 64ull * (W == 1 ? 1ull : 2ull) + 64ull));
    }
    return 4ull * ((size_t)p.peq_rows * fz_peq_stride(W) + 3ull * p.pmax + (size_t)p.nwaves * 64ull * ((size_t)W + p.ntkw) +
                   (p.wide ? ((size_t)p.alphabet + 1ull) / 2ull : 0ull));
}

template <int W, int MODE, bool WIDE>
static hipError_t launch_general3(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_score_general<W, MODE, WIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((fz_score_general<W, MODE, WIDE>), grid, dim3(p.nwaves * 64u), lds, st, p);
    return hipGetLastError();
}

template <int W, int MODE>
static hipError_t launch_general2(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    return p.wide ? launch_general3<W, MODE, true>(p, grid, lds, st) : launch_general3<W, MODE, false>(p, grid, lds, st);
}

template <int W, bool WIDE>
static hipError_t launch_partial2(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_score_partial<W, WIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((fz_score_partial<W, WIDE>), grid, dim3(p.nwaves * 64u), lds, st, p);
    return hipGetLastError();
}

template <int W>
static hipError_t launch_partial(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    return p.wide ? launch_partial2<W, true>(p, grid, lds, st) : launch_partial2<W, false>(p, grid, lds, st);
}

template <int W, bool WIDE>
static hipError_t launch_ro2(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fz_ro_matches<W, WIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((fz_ro_matches<W, WIDE>), grid, dim3(p.nwaves * 64u), lds, st, p);
    return hipGetLastError();
}

template <int W>
static hipError_t launch_ro(const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    if constexpr (W <= 2) return p.wide ? launch_ro2<W, true>(p, grid, lds, st) : launch_ro2<W, false>(p, grid, lds, st);
    else return hipErrorInvalidValue;                             // patterns beyond 64 symbols: refused by the host
}

template <int W>
static hipError_t launch_general(int mode, const FzScoreParams &p, dim3 grid, size_t lds, hipStream_t st)
{
    if ((mode == FZ_MODE_RO) != (p.ro != 0u)) return hipErrorInvalidValue;
    switch (mode) {
    case FZ_MODE_RO: return launch_ro<W>(p, grid, lds, st);
    case FZ_MODE_PARTIAL: return launch_partial<W>(p, grid, lds, st);
    case FZ_MODE_TOKEN_SET: return launch_general2<W, FZ_MODE_TOKEN_SET>(p, grid, lds, st);
    case FZ_MODE_RATIO: return launch_general2<W, FZ_MODE_RATIO>(p, grid, lds, st);
    case FZ_MODE_INDEL: return launch_general2<W, FZ_MODE_INDEL>(p, grid, lds, st);
    case FZ_MODE_LEV: return launch_general2<W, FZ_MODE_LEV>(p, grid, lds, st);
    case FZ_MODE_NHITS: return launch_general2<W, FZ_MODE_NHITS>(p, grid, lds, st);
    case FZ_MODE_FIND: return launch_general2<W, FZ_MODE_FIND>(p, grid, lds, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t fz_launch_score(int W, int mode, const FzScoreParams &p, uint32_t nq, hipStream_t st)
{
    if (nq == 0u || p.ntiles == 0u) return hipSuccess;
    const size_t lds = fz_score_lds_bytes(W, p);
    // blockIdx.y indexes the queries of the launch and a grid holds at most 65535 rows in y: a larger
    // batch goes in slices, each with its own window of the query list
    for (uint32_t q0 = 0u; q0 < nq; q0 += FZ_GRID_Y_MAX) {
        const uint32_t cnt = nq - q0 < FZ_GRID_Y_MAX ? nq - q0 : FZ_GRID_Y_MAX;
        FzScoreParams ps = p;
        ps.queries = p.queries + q0;
        const dim3 grid((p.ntiles + p.tiles_per_block - 1u) / p.tiles_per_block, cnt);
        hipError_t e;
        switch (W) {
        case 1: e = launch_general<1>(mode, ps, grid, lds, st); break;
        case 2: e = launch_general<2>(mode, ps, grid, lds, st); break;
        case 3: e = launch_general<3>(mode, ps, grid, lds, st); break;
        case 4: e = launch_general<4>(mode, ps, grid, lds, st); break;
        case 6: e = launch_general<6>(mode, ps, grid, lds, st); break;
        case 8: e = launch_general<8>(mode, ps, grid, lds, st); break;
        case 12: e = launch_general<12>(mode, ps, grid, lds, st); break;
        case 16: e = launch_general<16>(mode, ps, grid, lds, st); break;
        case 24: e = launch_general<24>(mode, ps, grid, lds, st); break;
        case 32: e = launch_general<32>(mode, ps, grid, lds, st); break;
        default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t fz_launch_select(const FzSelectParams &p, uint32_t rows, hipStream_t st)
{
    if (rows == 0u || p.k == 0u) return hipSuccess;
    if (p.n_slices > 1u) {
        if ((uint64_t)p.n_slices * p.k > FZ_MAX_K || (p.n_slices & (p.n_slices - 1u)) || !p.slice_keys || !p.slice_done || p.out_lim || p.walk)
            return hipErrorInvalidValue;
        if (p.n / p.n_slices <= 262144u) hipLaunchKernelGGL(fz_select_topk<256u>, dim3(rows, p.n_slices), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(fz_select_topk<1024u>, dim3(rows, p.n_slices), dim3(1024), 0, st, p);
        return hipGetLastError();
    }
    if (p.n <= 262144u) hipLaunchKernelGGL(fz_select_topk<256u>, dim3(rows), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(fz_select_topk<1024u>, dim3(rows), dim3(1024), 0, st, p);
    return hipGetLastError();
}

hipError_t fz_launch_combine(const FzCombineParams &p, uint32_t rows, hipStream_t st)
{
    if (rows == 0u || p.n == 0u) return hipSuccess;
    const uint32_t bx = (uint32_t)((p.n + 255u) / 256u);
    for (uint32_t r0 = 0u; r0 < rows; r0 += FZ_GRID_Y_MAX) {          // blockIdx.y = row: at most 65535 per launch
        const uint32_t cnt = rows - r0 < FZ_GRID_Y_MAX ? rows - r0 : FZ_GRID_Y_MAX;
        FzCombineParams ps = p;
        ps.scores_a = p.scores_a + (uint64_t)r0 * p.stride_a;
        ps.scores_b = p.scores_b + (uint64_t)r0 * p.stride_b;
        ps.ranks = p.ranks + (uint64_t)r0 * p.stride_r;
        hipLaunchKernelGGL(fz_combine_rank, dim3(bx < 2048u ? bx : 2048u, cnt), dim3(256), 0, st, ps);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t fz_launch_select16(const FzSelect16Params &p, uint32_t rows, hipStream_t st)
{
    if (rows == 0u || p.k == 0u) return hipSuccess;
    hipLaunchKernelGGL(fz_select16_topk, dim3(rows), dim3(SEL16_THREADS), 0, st, p);
    return hipGetLastError();
}
