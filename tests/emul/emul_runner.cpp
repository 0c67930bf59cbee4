// emul_runner.cpp -- TEST INFRASTRUCTURE ONLY.  Drives the C ABI of include/fuzzyhip.h in the
// sanitizer (CPU-emulated) build of tts_amd/csrc on one case file and writes the raw results.
//   case file: 9 x u64 {magic, op, scorer_or_kind, min_score, k, n, ncps, nq, nqcps},
//              u32 cps[ncps], u64 offs[n+1], u32 qcps[nqcps], u64 qoffs[nq+1]
//   op 0 fz_score_all -> u8[nq*n];  op 1 fz_distance_all -> u16[nq*n];
//   op 2 fz_query_topk -> u64 idx[nq*k], i32 score[nq*k], u32 cnt[nq]
//   op 4 fz_corpus_save + fz_corpus_load, then fz_score_all on the loaded corpus -> u8[nq*n]
//   op 5 fz_query_topk_submit x 2 (the batch, then its first half) + a refused third, waits in reverse order -> as op 2
//   op 3 fz_query_topk_pair (scorer a | scorer b << 8; second corpus, queries and rank table appended) -> as op 2
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "fuzzyhip.h"

template <typename T>
static bool rd(FILE *f, std::vector<T> &v, size_t n)
{
    v.assign(n ? n : 1, T());
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: emul_runner case.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("case"); return 2; }
    uint64_t h[9];
    if (fread(h, 8, 9, f) != 9 || h[0] != 0x465A454D554C0001ull) { fprintf(stderr, "bad case header\n"); return 2; }
    const uint64_t op = h[1], sk = h[2], k = h[4], n = h[5], ncps = h[6], nq = h[7], nqcps = h[8];
    const int32_t min_score = (int32_t)(int64_t)h[3];
    std::vector<uint32_t> cps, qcps; std::vector<uint64_t> offs, qoffs;
    if (!rd(f, cps, ncps) || !rd(f, offs, n + 1) || !rd(f, qcps, nqcps) || !rd(f, qoffs, nq + 1)) { fprintf(stderr, "short case file\n"); return 2; }
    std::vector<uint32_t> cps_b, qcps_b; std::vector<uint64_t> offs_b, qoffs_b; std::vector<uint16_t> rank_tab;
    if (op == 3) {          // fz_query_topk_pair: second corpus, second query list, rank table follow
        uint64_t hb[2];
        if (fread(hb, 8, 2, f) != 2 || !rd(f, cps_b, hb[0]) || !rd(f, offs_b, n + 1) || !rd(f, qcps_b, hb[1]) ||
            !rd(f, qoffs_b, nq + 1) || !rd(f, rank_tab, 101 * 101)) { fprintf(stderr, "short pair case\n"); return 2; }
    }
    fclose(f);

    fz_corpus *c = nullptr;
    int rc = fz_corpus_create(cps.data(), offs.data(), n, 0, 0, &c);
    if (rc != FZ_OK) { fprintf(stderr, "fz_corpus_create: %d %s\n", rc, fz_last_error()); return 3; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    if (op == 0) {
        std::vector<uint8_t> out(nq * n ? nq * n : 1);
        rc = fz_score_all(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, out.data());
        if (rc == FZ_OK) fwrite(out.data(), 1, nq * n, o);
    } else if (op == 1) {
        std::vector<uint16_t> out(nq * n ? nq * n : 1);
        rc = fz_distance_all(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, out.data());
        if (rc == FZ_OK) fwrite(out.data(), 2, nq * n, o);
    } else if (op == 4) {     // save -> destroy -> load -> fz_score_all on the loaded corpus
        std::string path = std::string(argv[2]) + ".corpus";
        std::vector<uint8_t> out(nq * n ? nq * n : 1), warm(nq * n ? nq * n : 1);
        if (sk != 0) rc = fz_score_all(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, warm.data());   // builds that view too
        if (rc == FZ_OK) rc = fz_corpus_save(c, path.c_str());
        fz_corpus_destroy(c);
        c = nullptr;
        if (rc == FZ_OK) rc = fz_corpus_load(path.c_str(), 0, ~0ull, &c);
        if (rc == FZ_OK) rc = fz_score_all(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, out.data());
        if (rc == FZ_OK) fwrite(out.data(), 1, nq * n, o);
        if (rc == FZ_OK && h[3] == 1) {   // truncated copy must be refused, not crash
            FILE *src = fopen(path.c_str(), "rb"); std::string tp = path + ".trunc"; FILE *dst = fopen(tp.c_str(), "wb");
            fseek(src, 0, SEEK_END); long sz = ftell(src); fseek(src, 0, SEEK_SET);
            std::vector<char> buf(sz * 2 / 3); if (fread(buf.data(), 1, buf.size(), src) != buf.size()) return 2;
            fwrite(buf.data(), 1, buf.size(), dst); fclose(src); fclose(dst);
            fz_corpus *bad = nullptr;
            if (fz_corpus_load(tp.c_str(), 0, ~0ull, &bad) == FZ_OK) { fprintf(stderr, "truncated file was accepted\n"); return 4; }
        }
    } else if (op == 3) {
        fz_corpus *cb = nullptr;
        rc = fz_corpus_create(cps_b.data(), offs_b.data(), n, 0, 0, &cb);
        if (rc != FZ_OK) { fprintf(stderr, "fz_corpus_create(b): %d %s\n", rc, fz_last_error()); return 3; }
        std::vector<uint64_t> idx(nq * k ? nq * k : 1);
        std::vector<int32_t> rk(nq * k ? nq * k : 1);
        std::vector<uint32_t> cnt(nq ? nq : 1);
        rc = fz_query_topk_pair(c, (int)(sk & 0xFF), qcps.data(), qoffs.data(), cb, (int)(sk >> 8), qcps_b.data(), qoffs_b.data(),
                                (uint32_t)nq, rank_tab.data(), (uint32_t)h[3], (uint32_t)k, idx.data(), rk.data(), cnt.data());
        if (rc == FZ_OK) { fwrite(idx.data(), 8, nq * k, o); fwrite(rk.data(), 4, nq * k, o); fwrite(cnt.data(), 4, nq, o); }
        fz_corpus_destroy(cb);
    } else if (op == 5) {
        // two calls in flight: the whole batch (ticket a) and its first half (ticket b); a third submit is refused; b is
        // fetched first; b's rows must be a's
        std::vector<uint64_t> idx(nq * k ? nq * k : 1), idx_b(idx.size(), ~0ull);
        std::vector<int32_t> sc(nq * k ? nq * k : 1), sc_b(sc.size(), -1);
        std::vector<uint32_t> cnt(nq ? nq : 1), cnt_b(cnt.size(), ~0u);
        const uint32_t nb = (uint32_t)(nq / 2);
        int ta = -1, tb = -1, tc = -1;
        rc = fz_query_topk_submit(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, min_score, (uint32_t)k, &ta);
        if (rc == FZ_OK) rc = fz_query_topk_submit(c, qcps.data(), qoffs.data(), nb, (int)sk, min_score, (uint32_t)k, &tb);
        if (rc == FZ_OK && fz_query_topk_submit(c, qcps.data(), qoffs.data(), nb, (int)sk, min_score, (uint32_t)k, &tc) != FZ_ERR_LIMIT) {
            fprintf(stderr, "a third call in flight was accepted\n"); return 4;
        }
        if (rc == FZ_OK) rc = fz_query_topk_wait(c, tb, idx_b.data(), sc_b.data(), cnt_b.data());
        if (rc == FZ_OK) rc = fz_query_topk_wait(c, ta, idx.data(), sc.data(), cnt.data());
        if (rc == FZ_OK && fz_query_topk_wait(c, ta, idx.data(), sc.data(), cnt.data()) == FZ_OK) { fprintf(stderr, "a ticket was fetched twice\n"); return 4; }
        if (rc == FZ_OK) {
            for (uint32_t q = 0; q < nb; ++q) {
                bool same = cnt_b[q] == cnt[q];
                for (uint64_t j = 0; same && j < k; ++j) same = idx_b[q * k + j] == idx[q * k + j] && sc_b[q * k + j] == sc[q * k + j];
                if (!same) { fprintf(stderr, "row %u of the second call in flight differs from the first call's\n", q); return 4; }
            }
            // the slots are free again; this time in submission order: the second call has only its tables built when the
            // first one's wait sees the GPU free and sends it out
            std::vector<uint64_t> idx_c(idx.size(), ~0ull); std::vector<int32_t> sc_c(sc.size(), -1); std::vector<uint32_t> cnt_c(cnt.size(), ~0u);
            int td = -1;
            rc = fz_query_topk_submit(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, min_score, (uint32_t)k, &tc);
            if (rc == FZ_OK) rc = fz_query_topk_submit(c, qcps.data(), qoffs.data(), nb, (int)sk, min_score, (uint32_t)k, &td);
            if (rc == FZ_OK) rc = fz_query_topk_wait(c, tc, idx_c.data(), sc_c.data(), cnt_c.data());
            std::fill(idx_b.begin(), idx_b.end(), ~0ull);
            if (rc == FZ_OK) rc = fz_query_topk_wait(c, td, idx_b.data(), sc_b.data(), cnt_b.data());
            if (rc == FZ_OK) {
                bool same = idx_c == idx && sc_c == sc && cnt_c == cnt;
                for (uint32_t q = 0; same && q < nb; ++q) {
                    same = cnt_b[q] == cnt[q];
                    for (uint64_t j = 0; same && j < k; ++j) same = idx_b[q * k + j] == idx[q * k + j] && sc_b[q * k + j] == sc[q * k + j];
                }
                if (!same) { fprintf(stderr, "the second round of calls in flight gave other results\n"); return 4; }
            }
        }
        if (rc == FZ_OK) { fwrite(idx.data(), 8, nq * k, o); fwrite(sc.data(), 4, nq * k, o); fwrite(cnt.data(), 4, nq, o); }
    } else {
        std::vector<uint64_t> idx(nq * k ? nq * k : 1);
        std::vector<int32_t> sc(nq * k ? nq * k : 1);
        std::vector<uint32_t> cnt(nq ? nq : 1);
        rc = fz_query_topk(c, qcps.data(), qoffs.data(), (uint32_t)nq, (int)sk, min_score, (uint32_t)k, idx.data(), sc.data(), cnt.data());
        if (rc == FZ_OK) { fwrite(idx.data(), 8, nq * k, o); fwrite(sc.data(), 4, nq * k, o); fwrite(cnt.data(), 4, nq, o); }
    }
    fclose(o);
    if (rc != FZ_OK) { fprintf(stderr, "scoring call: %d %s\n", rc, fz_last_error()); if (c) fz_corpus_destroy(c); return 3; }
    fz_corpus_destroy(c);
    return 0;
}
