// dme_ingest.hip — host-side text ingestion for the CLI (no device code).
//
// Replaces the per-row `csv.reader` + `list(map(int, row))` loop of
// DirichletEstimation/findDirichletPrior.py:67-79 for the common case: every line holds exactly K
// base-10 integers separated by one delimiter byte.  Anything the strict grammar below does not
// cover (quotes, blank lines, ragged rows, signs other than a leading +/-, underscores, values
// beyond int32, ...) makes the parser give up with DME_PARSE_FALLBACK and the caller re-reads the
// text with the reference's own row loop, so error behaviour stays the reference's.
//
// Grammar of a field: [ ]* [+-]? [0-9]+ [ ]*      line end: \n or \r\n; a final line may lack it.
// Threads split the buffer at newline boundaries: pass 1 counts lines, pass 2 parses in place.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "dme_common.h"

namespace {

struct Chunk {
    const char* b;
    const char* e;
    long long lines = 0;
    long long first_line = 0;
    long long first_row = 0;
    int status = 0;  // 0 ok, 1 fallback
};

inline long long count_lines(const char* b, const char* e) {
    long long n = 0;
    for (const char* p = b; p < e;) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
        if (!q) { n += 1; break; }  // unterminated last line
        n += 1;
        p = q + 1;
    }
    return n;
}

// parses [b, e) (whole lines) into out[first_row...]; returns 0 or 1 (fallback).
// keep != NULL: keep[i] says whether line first_line + i is wanted; the others are skipped UNPARSED (the
// reference's loop converts the fields of a row only after its Bernoulli draw kept it, DE/fdp.py:78-79,
// so text it never looks at cannot fail there either) and the kept ones are written back to back.
int parse_chunk(const char* b, const char* e, char delim, long long K, int32_t* out,
                long long first_row, const unsigned char* keep = nullptr, long long first_line = 0) {
    int32_t* o = out + first_row * K;
    const char* p = b;
    long long line = first_line;
    while (p < e) {
        if (keep && !keep[line++]) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            p = nl ? nl + 1 : e;
            continue;
        }
        for (long long k = 0; k < K; ++k) {
            // count matrices are mostly one-digit cells: "d<delim>" takes this short cut
            if (k + 1 < K && e - p >= 2 && (unsigned)(p[0] - '0') <= 9u && p[1] == delim) {
                *o++ = p[0] - '0';
                p += 2;
                continue;
            }
            while (p < e && *p == ' ') ++p;
            bool neg = false;
            if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
            if (p >= e || *p < '0' || *p > '9') return 1;
            long long v = 0;
            int digits = 0;
            while (p < e && *p >= '0' && *p <= '9') {
                v = v * 10 + (*p - '0');
                if (++digits > 10) return 1;
                ++p;
            }
            while (p < e && *p == ' ') ++p;
            if (neg) v = -v;
            if (v > 2147483647LL || v < -2147483648LL) return 1;
            *o++ = (int32_t)v;
            if (k + 1 < K) {
                if (p >= e || *p != delim) return 1;
                ++p;
            }
        }
        if (p < e) {
            if (*p == '\r') ++p;
            if (p < e) {
                if (*p != '\n') return 1;
                ++p;
            }
        }
    }
    return 0;
}

}  // namespace

extern "C" int64_t dme_parse_count_rows(const char* text, int64_t nbytes) {
    if (!text || nbytes <= 0) return 0;
    return count_lines(text, text + nbytes);
}

namespace {
int parse_impl(const char* text, int64_t nbytes, int delimiter, int64_t K, const unsigned char* keep,
               int64_t n_keep, int32_t* out, int64_t out_rows, int64_t* n_rows, int nthreads);
}

extern "C" int dme_parse_counts(const char* text, int64_t nbytes, int delimiter, int64_t K,
                                int32_t* out, int64_t out_rows, int64_t* n_rows, int nthreads) {
    return parse_impl(text, nbytes, delimiter, K, nullptr, 0, out, out_rows, n_rows, nthreads);
}

extern "C" int dme_parse_counts_selected(const char* text, int64_t nbytes, int delimiter, int64_t K,
                                         const uint8_t* keep, int64_t n_keep, int32_t* out,
                                         int64_t out_rows, int64_t* n_rows, int nthreads) {
    DME_REQUIRE(keep != nullptr && n_keep >= 0, "keep is null");
    return parse_impl(text, nbytes, delimiter, K, keep, n_keep, out, out_rows, n_rows, nthreads);
}

namespace {

int parse_impl(const char* text, int64_t nbytes, int delimiter, int64_t K, const unsigned char* keep,
               int64_t n_keep, int32_t* out, int64_t out_rows, int64_t* n_rows, int nthreads) {
    DME_REQUIRE(K >= 1 && n_rows, "bad arguments");
    *n_rows = 0;
    if (!text || nbytes <= 0) return DME_OK;
    DME_REQUIRE(out != nullptr, "out is null");
    DME_REQUIRE(delimiter > 0 && delimiter < 128 && delimiter != '\n' && delimiter != '\r' &&
                    delimiter != ' ' && !(delimiter >= '0' && delimiter <= '9') &&
                    delimiter != '+' && delimiter != '-' && delimiter != '"',
                "delimiter not supported by the fast parser");
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nbytes < (1 << 20)) nthreads = 1;
    const char* b = text;
    const char* e = text + nbytes;
    std::vector<Chunk> ch((size_t)nthreads);
    const char* p = b;
    for (int t = 0; t < nthreads; ++t) {
        const char* q = (t + 1 == nthreads) ? e : b + (nbytes / nthreads) * (t + 1);
        if (q < p) q = p;
        if (q < e) {  // advance to just after the next newline
            const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
            q = nl ? nl + 1 : e;
        }
        ch[(size_t)t].b = p;
        ch[(size_t)t].e = q;
        p = q;
    }
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t)
            th.emplace_back([&ch, t] { ch[(size_t)t].lines = count_lines(ch[(size_t)t].b, ch[(size_t)t].e); });
        ch[0].lines = count_lines(ch[0].b, ch[0].e);
        for (auto& x : th) x.join();
    }
    long long total = 0, lines = 0;
    for (auto& c : ch) {
        c.first_line = lines;
        lines += c.lines;
    }
    if (keep && lines != n_keep)
        return dme_set_error(DME_ERR_INVALID_ARG, "parse: %lld lines, keep holds %lld", lines,
                             (long long)n_keep);
    for (auto& c : ch) {  // rows written: every line, or the kept ones
        c.first_row = total;
        if (keep) {
            long long n = 0;
            for (long long i = 0; i < c.lines; ++i) n += keep[c.first_line + i] != 0;
            total += n;
        } else {
            total += c.lines;
        }
    }
    if (total > out_rows)
        return dme_set_error(DME_ERR_WORKSPACE, "parse: %lld rows, room for %lld", total,
                             (long long)out_rows);
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t)
            th.emplace_back([&ch, t, delimiter, K, out, keep] {
                Chunk& c = ch[(size_t)t];
                c.status = parse_chunk(c.b, c.e, (char)delimiter, K, out, c.first_row, keep, c.first_line);
            });
        ch[0].status = parse_chunk(ch[0].b, ch[0].e, (char)delimiter, K, out, ch[0].first_row, keep,
                                   ch[0].first_line);
        for (auto& x : th) x.join();
    }
    for (auto& c : ch)
        if (c.status) return DME_PARSE_FALLBACK;
    *n_rows = total;
    return DME_OK;
}

}  // namespace

// -s and -M on parsed rows with the reference's draws in the reference's order (DE/fdp.py:78, :85;
// drawCategory: DE/samplingTools.py:49-58).  u[] is the caller's supply of uniform draws in stream
// order; row i takes one (kept iff u < sample_rate) and, if kept and above the cap, one per decrement:
// threshold = sum(counts) * u, the first index whose running sum exceeds it (else the last index) loses
// one count.  Kept rows are moved to the front of X in place.
extern "C" int64_t dme_cap_draws_needed(const int32_t* X, int64_t n, int64_t K, int64_t M) {
    if (!X || n < 0 || K < 1) return -1;
    long long need = n;
    for (long long i = 0; i < n; ++i) {
        long long tot = 0;
        for (long long k = 0; k < K; ++k) tot += X[i * K + k];
        if (tot > M) need += tot - M;
    }
    return need;
}

extern "C" int dme_sample_and_cap_i32(int32_t* X, int64_t n, int64_t K, double sample_rate, int64_t M,
                                      const double* u, int64_t n_u, int64_t* n_kept, int64_t* n_used) {
    DME_REQUIRE(X && u && n_kept && n_used && n >= 0 && K >= 1, "bad arguments");
    long long p = 0, w = 0;
    for (long long i = 0; i < n; ++i) {
        if (p >= n_u) return dme_set_error(DME_ERR_WORKSPACE, "sample_and_cap: out of uniform draws");
        const bool keep = u[p++] < sample_rate;
        if (!keep) continue;
        int32_t* row = X + i * K;
        long long tot = 0;
        for (long long k = 0; k < K; ++k) tot += row[k];
        while (tot > M) {
            if (p >= n_u) return dme_set_error(DME_ERR_WORKSPACE, "sample_and_cap: out of uniform draws");
            const double threshold = (double)tot * u[p++];
            long long run = 0, idx = K - 1;
            for (long long k = 0; k < K; ++k) {
                run += row[k];
                if (threshold < (double)run) { idx = k; break; }
            }
            row[idx] -= 1;
            tot -= 1;
        }
        if (w != i) memmove(X + w * K, row, (size_t)K * sizeof(int32_t));
        ++w;
    }
    *n_kept = w;
    *n_used = p;
    return DME_OK;
}
