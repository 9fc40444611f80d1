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

// parses [b, e) (whole lines) into out[first_row...]; returns 0 or 1 (fallback)
int parse_chunk(const char* b, const char* e, char delim, long long K, int32_t* out,
                long long first_row) {
    int32_t* o = out + first_row * K;
    const char* p = b;
    while (p < e) {
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

extern "C" int dme_parse_counts(const char* text, int64_t nbytes, int delimiter, int64_t K,
                                int32_t* out, int64_t out_rows, int64_t* n_rows, int nthreads) {
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
    long long total = 0;
    for (auto& c : ch) { c.first_row = total; total += c.lines; }
    if (total > out_rows)
        return dme_set_error(DME_ERR_WORKSPACE, "parse: %lld rows, room for %lld", total,
                             (long long)out_rows);
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t)
            th.emplace_back([&ch, t, delimiter, K, out] {
                Chunk& c = ch[(size_t)t];
                c.status = parse_chunk(c.b, c.e, (char)delimiter, K, out, c.first_row);
            });
        ch[0].status = parse_chunk(ch[0].b, ch[0].e, (char)delimiter, K, out, ch[0].first_row);
        for (auto& x : th) x.join();
    }
    for (auto& c : ch)
        if (c.status) return DME_PARSE_FALLBACK;
    *n_rows = total;
    return DME_OK;
}
